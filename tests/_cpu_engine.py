"""TEST DOUBLE (tests only): an engine with the HipEngine interface whose local stages run on the CPU
through the oracle, and whose particle exchange goes through torch.distributed (gloo).

Purpose: exercise the PRODUCT's sharded driver logic (smc_lt_amd.driver.run_smc with a multi-rank
communicator: residual prefix, output-slot bases, reductions, loop control) with world_size 2 on a
machine without GPUs.  The product never imports this file."""
import numpy as np

PRED, FILT = 0, 1


class OracleEngine:
    def __init__(self, O, data, priors, n_local, n_global, rank, world, dist=None):
        self.O, self.data, self.priors = O, data, priors
        self.n_local, self.n_global, self.dim = n_local, n_global, 3
        self.rank, self.world, self.dist = rank, world, dist
        self.theta = [np.zeros((n_local, 3)), np.zeros((n_local, 3))]
        self.lk = [np.zeros(n_local), np.zeros(n_local)]
        self.r_ac = np.zeros(n_local, dtype=np.uint8)
        self._o = None

    # movement
    def upload_particles(self, which, aos):
        self.theta[which] = np.array(aos, dtype=np.float64)

    def download_particles(self, which, n=None):
        return self.theta[which].copy()

    def upload_lk(self, which, lk):
        self.lk[which] = np.array(lk, dtype=np.float64)

    def download_lk(self, which, n=None):
        return self.lk[which].copy()

    def commit_filt_to_pred(self):
        self.theta[PRED] = self.theta[FILT].copy()
        self.lk[PRED] = self.lk[FILT].copy()

    def set_prior_mode(self, mode):           # the test double implements the reference's live branch only
        assert mode in ("mask", 0)

    def set_resampling(self, scheme):
        assert scheme in ("residual_systematic", 0)

    def synchronize(self):
        pass

    def reset_accept_flags(self):
        self.r_ac[:] = 0

    # likelihood
    def loglik(self, which=PRED):
        lk, _, info = self.O.mm_loglik_batch(self.theta[which], self.data)
        self.lk[which] = lk
        return {"n_failed": info["n_failed"], "rk_attempts": info["n_attempts"]}

    # weights
    def max_lk_local(self):
        return float(np.max(self.lk[PRED]))

    def ess_partials(self, max_lk, gms):
        d = self.lk[PRED] - max_lk
        sw = np.array([np.sum(np.exp(d * g)) for g in gms])
        sw2 = np.array([np.sum(np.exp(d * g) ** 2) for g in gms])
        return sw, sw2

    # resampling: the same three-phase protocol as include/smc_hip.h
    def _wcr(self, max_lk, gm, sum_w):
        N = self.n_global
        w = np.exp((self.lk[PRED] - max_lk) * gm) / sum_w
        c = np.trunc(w * N).astype(np.int64)
        r = w - c * (1 / N)
        return w, c, r

    def resample_phase1(self, max_lk, gm, sum_w):
        _, c, r = self._wcr(max_lk, gm, sum_w)
        s = 0.0
        for x in r:
            s += x
        return s, int(c.sum())

    def resample_phase2(self, max_lk, gm, sum_w, residual_prefix, wrand):
        N = self.n_global
        _, c, r = self._wcr(max_lk, gm, sum_w)

        def m(S):
            return int(np.floor((S - wrand) * N)) + 1 if S >= wrand else 0
        S = residual_prefix
        m_prev = 0 if self.rank == 0 else m(S)
        o = c.copy()
        for j in range(self.n_local):
            S += r[j]
            mj = m(S)
            o[j] += max(mj - m_prev, 0)
            m_prev = max(mj, m_prev)
        self._o = o
        return int(o.sum())

    def download_offspring(self):
        return self._o.copy()

    def resample_phase3(self, bases, o_all, first_step):
        nl, R = self.n_local, self.rank
        rows = np.repeat(np.arange(nl), self._o)
        mine = (int(bases[R]), self.theta[PRED][rows], self.lk[PRED][rows])
        if self.world > 1:
            box = [None] * self.world
            self.dist.all_gather_object(box, mine)
        else:
            box = [mine]
        total = int(np.sum(o_all))
        new_t = np.zeros((nl, 3)) if first_step else self.theta[PRED].copy()
        new_l = np.zeros(nl) if first_step else self.lk[PRED].copy()
        for base, th, lk in box:
            for k in range(len(lk)):
                slot = base + k
                if R * nl <= slot < (R + 1) * nl:
                    new_t[slot - R * nl] = th[k]
                    new_l[slot - R * nl] = lk[k]
        assert total <= self.n_global
        self.theta[FILT], self.lk[FILT] = new_t, new_l

    # moments
    def moment_sums_local(self):
        return self.theta[FILT].sum(axis=0)

    def moment_centered_local(self, mean):
        x = self.theta[FILT] - np.asarray(mean)
        return x.T @ x

    # MH
    def mh_step_host_rng(self, gamma, ratio, noise, rr):
        O = self.O
        p_filt, lk1 = self.theta[FILT], self.lk[FILT]
        p_pred = p_filt + noise * ratio
        p0 = np.int32(O.cal_prior(p_pred, self.priors) > 0)
        p_pred = p_pred * p0[:, None] + p_filt * (1.0 - p0[:, None])
        lk2, _, info = O.mm_loglik_batch(p_pred, self.data)
        with np.errstate(over="ignore"):
            r = np.int32(np.exp((lk2 - lk1) * gamma) * p0 >= rr)
        self.theta[FILT] = p_pred * r[:, None] + p_filt * (1.0 - r[:, None])
        self.lk[FILT] = lk2 * r + lk1 * (1.0 - r)
        self.theta[PRED] = p_pred
        self.r_ac = np.maximum(self.r_ac, r.astype(np.uint8))
        return {"accepted_now": int(r.sum()), "accepted_ever": int(self.r_ac.sum()), "n_failed": info["n_failed"],
                "rk_attempts": info["n_attempts"]}

    # the *_global entry points (include/smc_hip.h): the engine reduces over the ranks itself - here through
    # torch.distributed, in the product through RCCL on the device buffers
    def _comm(self):
        from _torch_comm import TorchDistComm
        if self.world == 1:
            return None
        if getattr(self, "_tc", None) is None:
            self._tc = TorchDistComm()
        return self._tc

    def max_lk_global(self):
        c = self._comm()
        v = self.max_lk_local()
        return float(c.allreduce_max([v])[0]) if c else v

    def ess_partials_global(self, max_lk, gms):
        sw, sw2 = self.ess_partials(max_lk, gms)
        c = self._comm()
        if c:
            tot = c.allreduce_sum(np.concatenate([sw, sw2]))
            sw, sw2 = tot[:len(gms)], tot[len(gms):]
        return sw, sw2

    def ess_search_global(self, gms, with_max=True):
        """include/smc_hip.h: smc_ess_search_global - maximum + up to 32 candidates, one synchronisation."""
        assert 1 <= len(gms) <= 32
        if with_max:
            self._search_max = self.max_lk_global()
        sw, sw2 = self.ess_partials_global(self._search_max, gms)
        return self._search_max, sw, sw2

    def resample_global(self, max_lk, gm, sum_w, wrand, first_step):
        c = self._comm()
        r_loc, c_loc = self.resample_phase1(max_lk, gm, sum_w)
        r_all = c.allgather([r_loc])[:, 0] if c else np.array([r_loc])
        c_all = c.allgather_i64([c_loc])[:, 0] if c else np.array([c_loc])
        prefix = 0.0
        for q in range(self.rank):
            prefix = prefix + float(r_all[q])
        o_loc = self.resample_phase2(max_lk, gm, sum_w, prefix, wrand)
        o_all = c.allgather_i64([o_loc])[:, 0] if c else np.array([o_loc])
        bases = np.concatenate([[0], np.cumsum(o_all)[:-1]]).astype(np.int64)
        self.resample_phase3(bases, o_all, first_step)
        return {"n_offspring": int(o_all.sum()), "count_sum": int(c_all.sum())}


    # round 4: the deferred form of the same call (HipEngine.resample_enqueue / resample_result): run_smc enqueues the
    # resampling and reads its two logged numbers after the Metropolis loop
    def resample_enqueue(self, max_lk, gm, sum_w, wrand, first_step):
        self._rs = self.resample_global(max_lk, gm, sum_w, wrand, first_step)
        self.n_deferred = getattr(self, "n_deferred", 0) + 1

    def resample_result(self):
        rs, self._rs = self._rs, None
        assert rs is not None, "resample_result without resample_enqueue"
        return rs


class EngineSideComm:
    """What comm.RcclComm is to the HipEngine: `on_device` tells the driver to use the engine's *_global entry points;
    the remaining host-side collectives (parity mode draws the noise on the host, so its moments still go through the
    communicator) are delegated."""
    on_device = True

    def __init__(self, inner):
        self._c = inner
        self.rank, self.size = inner.rank, inner.size

    def __getattr__(self, name):
        return getattr(self._c, name)
