"""HipEngine - Python handle on one libsmc_hip.so context (one per process and GPU).

Every method is a thin call through the C ABI (include/smc_hip.h); all arithmetic happens in the
HIP kernels.  NumPy is used only to own the host buffers the reference would own
("Ownership").
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import binding as B
from .binding import SMC_SET_FILT, SMC_SET_PRED, SmcError, check, lib


def _dp(a):
    return a.ctypes.data_as(B.c_dp)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class _PinnedPool:
    """Page-locked host buffers for downloaded results (include/smc_hip.h: smc_pinned_alloc).  A buffer belongs to the NumPy array
    that views it and returns to the pool when that array (and every view of it) is garbage-collected; the memory is
    process-wide, so such an array stays valid after its engine is closed.  Buffers are reused by size: a run's final 24 + 8 MB
    download costs one DMA transfer each instead of the runtime's staged copy into pageable memory.
    Thread-safe (several engines may run in threads of one process); at most `cap_bytes` of returned buffers are kept - what
    exceeds the cap is unpinned and freed at once (smc_pinned_free), and release() / interpreter exit free the rest."""

    def __init__(self, cap_bytes=1 << 30):
        import threading
        self.free = {}          # nbytes -> [ptr, ...]
        self.pooled = 0         # bytes held in `free`
        self.cap_bytes = int(cap_bytes)
        self.lock = threading.Lock()

    def take(self, nbytes):
        with self.lock:
            lst = self.free.get(nbytes)
            if lst:
                self.pooled -= nbytes
                return lst.pop()
        p = ctypes.c_void_p(0)
        st = lib().smc_pinned_alloc(ctypes.c_size_t(nbytes), ctypes.byref(p))
        if st != 0:
            msg = lib().smc_last_error(None)
            raise SmcError(f"smc_pinned_alloc: {msg.decode() if msg else 'unknown error'}")
        return p.value

    def give(self, ptr, nbytes):
        with self.lock:
            if self.pooled + nbytes <= self.cap_bytes:
                self.free.setdefault(nbytes, []).append(ptr)
                self.pooled += nbytes
                return
        lib().smc_pinned_free(ctypes.c_void_p(ptr))

    def release(self):
        """Unpin and free every buffer the pool holds (buffers still owned by live arrays are not touched)."""
        with self.lock:
            ptrs = [p for lst in self.free.values() for p in lst]
            self.free.clear()
            self.pooled = 0
        for p in ptrs:
            lib().smc_pinned_free(ctypes.c_void_p(p))


_PINNED = _PinnedPool()


def release_pinned_pool():
    """Free the page-locked result buffers that are waiting for reuse (a size sweep would otherwise keep them for the life of the process)."""
    _PINNED.release()


import atexit  # noqa: E402

atexit.register(lambda: _PINNED.release() if B._LIB is not None else None)


class _PinnedOwner:
    """What a pinned result array keeps alive (ndarray.base): gives the buffer back to the pool when the last view is gone."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        self.nbytes = max(8, int(np.prod(self.shape)) * self.dtype.itemsize)
        self.ptr = _PINNED.take(self.nbytes)
        self.__array_interface__ = {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            _PINNED.give(self.ptr, self.nbytes)
        except Exception:       # interpreter shutdown
            pass


def pinned_empty(shape, dtype=np.float64):
    """np.empty in page-locked host memory (see _PinnedPool)."""
    return np.asarray(_PinnedOwner(shape, dtype))


class HipEngine:
    """Device-resident particle sets p_pred/lk and p_filt/lk1 plus the stages that act on them."""
    pinned_downloads = True     # download_particles / download_lk accept pinned=True (page-locked result arrays)

    def __init__(self, n_local: int, dim: int = 3, device: int = 0, n_global: int | None = None):
        self.L = lib()
        self.n_local = int(n_local)
        self.n_global = int(n_global if n_global is not None else n_local)
        self.dim = int(dim)
        self.device = int(device)
        self.rank, self.world = 0, 1
        ctx = B.c_ctx()
        st = self.L.smc_create(ctypes.byref(ctx), self.device, self.n_local, self.n_global, self.dim)
        if st != 0:
            msg = self.L.smc_last_error(None)
            raise SmcError(f"smc_create: {msg.decode() if msg else 'unknown error'}")
        self.ctx = ctx
        self.model = None
        if "smc_ess_search_global" in B.MISSING:   # A/B build of an older revision (SMC_HIP_LIB): the driver falls back
            self.ess_search_global = None
        self._peer_barrier = None

    # ---- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.L.smc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st, what):
        check(self.ctx, st, what)

    def synchronize(self):
        self._ck(self.L.smc_synchronize(self.ctx), "smc_synchronize")

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        arch = ctypes.create_string_buffer(256)
        cu = ctypes.c_int(0)
        self._ck(self.L.smc_device_info(self.ctx, name, 256, arch, 256, ctypes.byref(cu)), "smc_device_info")
        return {"name": name.value.decode(), "arch": arch.value.decode(), "cu_count": cu.value}

    # ---- model / prior -------------------------------------------------------------------------
    def set_model_mm(self, t, P_obs, S0, est_sigma=True, sigma_fixed=5.0, rtol=1e-3, atol=1e-6):
        t = _f64(t)
        P_obs = _f64(P_obs, t.shape)
        S0 = _f64(S0, (t.shape[0],))
        self._ck(self.L.smc_set_model_mm(self.ctx, _dp(t), _dp(P_obs), _dp(S0), t.shape[0], t.shape[1],
                                         int(bool(est_sigma)), float(sigma_fixed), float(rtol), float(atol)),
                 "smc_set_model_mm")
        self.model = ("mm", t.shape[0], t.shape[1])

    def set_model_user(self, source: str, n_states: int, t, obs, cond=None, est_sigma=True, sigma_fixed=5.0, rtol=1e-3,
                       atol=1e-6):
        """A user-written model in place of Micmem_likelihood.py (include/smc_hip.h, smc_set_model_user): `source`
        defines smc_user_y0 / smc_user_rhs / smc_user_obs as HIP device functions; t, obs: (n_ex, n_t); cond: (n_ex, n_cond)
        per-experiment numbers (e.g. the initial concentration).  Raises SmcError with the compiler log if the source
        does not compile."""
        t = _f64(t)
        obs = _f64(obs, t.shape)
        cond = np.zeros((t.shape[0], 0)) if cond is None else _f64(np.asarray(cond).reshape(t.shape[0], -1))
        cbuf = np.ascontiguousarray(cond if cond.size else np.zeros((t.shape[0], 1)))
        self._ck(self.L.smc_set_model_user(self.ctx, source.encode(), int(n_states), _dp(t), _dp(obs), _dp(cbuf), t.shape[0],
                                           t.shape[1], cond.shape[1], int(bool(est_sigma)), float(sigma_fixed), float(rtol),
                                           float(atol)), "smc_set_model_user")
        self.model = ("user", t.shape[0], t.shape[1])

    def set_model_methanation(self, cond, guess, obs, base_params, est_position, est_sigma=True, sigma_fixed=5.0,
                              tf=75.0, rtol=1e-6, atol=1e-6):
        """cond: dict with Ca_in..Ce_in, T_in, T_jacket, u_in, void, reactorlength (the reference's settings arrays,
        methanation_set_conditon.py:141-214) or an (n_data, 10) array; guess (n_data, 357); obs (5, n_data)."""
        if isinstance(cond, dict):
            n_data = int(cond.get("n_data", len(cond["Ca_in"])))
            cols = [np.asarray(cond[k], dtype=np.float64)[:n_data] for k in
                    ("Ca_in", "Cb_in", "Cc_in", "Cd_in", "Ce_in", "T_in", "T_jacket", "u_in", "void")]
            cols.append(np.asarray(cond["reactorlength"], dtype=np.float64)[:n_data] / (51 - 1))
            cond = np.column_stack(cols)
        cond = _f64(cond)
        n_data = cond.shape[0]
        guess = _f64(np.asarray(guess)[:n_data], (n_data, 357))
        obs = _f64(obs, (5, n_data))
        base = _f64(base_params, (9,))
        pos = np.ascontiguousarray(est_position, dtype=np.int32)
        assert pos.shape == (self.dim,)
        self._ck(self.L.smc_set_model_methanation(self.ctx, _dp(cond), _dp(guess), _dp(obs), n_data, _dp(base),
                                                  pos.ctypes.data_as(B.c_ip), int(bool(est_sigma)), float(sigma_fixed),
                                                  float(tf), float(rtol), float(atol)), "smc_set_model_methanation")
        self.model = ("methanation", n_data, 357)

    def set_prior(self, priors: dict):
        """priors: the reference's dict (Micmem_settings.py:55-67), one entry per parameter, in order."""
        kinds, a, b = [], [], []
        for name, p in priors.items():
            if p["dist"] == "uniform":
                kinds.append(B.SMC_PRIOR_UNIFORM)
                a.append(p["low"])
                b.append(p["high"])
            elif p["dist"] == "normal":
                kinds.append(B.SMC_PRIOR_NORMAL)
                a.append(p["mu"])
                b.append(p["sigma"])
            elif p["dist"] == "flat":       # drawn like a normal, no factor in the prior density
                kinds.append(B.SMC_PRIOR_FLAT)
                a.append(p["mu"])
                b.append(p["sigma"])
            else:
                raise ValueError(f"Unknown prior: {p['dist']}")
        k = np.array(kinds, dtype=np.int32)
        a = np.array(a, dtype=np.float64)
        b = np.array(b, dtype=np.float64)
        self._ck(self.L.smc_set_prior(self.ctx, k.ctypes.data_as(B.c_ip), _dp(a), _dp(b), len(kinds)), "smc_set_prior")

    def meth_sweep_counters(self):
        """Device-counted work of the last methanation sweep: BDF steps, Newton iterations, factorisations, failed solves."""
        out = (ctypes.c_int64 * 4)()
        self._ck(self.L.smc_meth_sweep_counters(self.ctx, out), "smc_meth_sweep_counters")
        return {"bdf_steps": out[0], "newton_iters": out[1], "factorisations": out[2], "failed_solves": out[3]}

    def meth_sweep_check(self):
        """Completeness of the last methanation sweep (the library already fails the sweep when these disagree)."""
        out = (ctypes.c_int64 * 5)()
        self._ck(self.L.smc_meth_sweep_check(self.ctx, out), "smc_meth_sweep_check")
        return {"expected_solves": out[0], "completed_solves": out[1], "unsolved_items": out[2], "wave_split": out[3],
                "cancelled_solves": out[4]}

    def meth_download_solves(self, n=None):
        """Outlet flows (n, n_data, 5) and solver status (n, n_data) of the last methanation sweep."""
        n = self.n_local if n is None else n
        nd = self.model[1]
        flows = np.empty((n, nd, 5))
        status = np.empty((n, nd), dtype=np.int32)
        self._ck(self.L.smc_meth_download_solves(self.ctx, _dp(flows), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n),
                 "smc_meth_download_solves")
        return flows, status

    def set_prior_mode(self, mode):
        """"mask" (default; the live branch of both reference drivers), "ratio_mask" (normal_pred and taylor,
        SMC_methanation_main.py:320-349) or "ratio" (normal_pred, :358-374)."""
        m = B.PRIOR_MODES[mode] if isinstance(mode, str) else int(mode)
        self._ck(self.L.smc_set_prior_mode(self.ctx, m), "smc_set_prior_mode")

    def set_early_reject(self, enable=True):
        """Stop (Michaelis-Menten) or do not start (methanation) the solves of a proposal that is already certain to be rejected
        (include/smc_hip.h: smc_set_early_reject, smc_meth_sweep_check)."""
        self._ck(self.L.smc_set_early_reject(self.ctx, int(bool(enable))), "smc_set_early_reject")

    def set_exact_pow(self, enable=True):
        """Parity mode of the step controller: correctly rounded pow(x, -0.2) (include/smc_hip.h: smc_set_exact_pow)."""
        if "smc_set_exact_pow" in B.MISSING:       # A/B build of an older revision (SMC_HIP_LIB)
            return
        self._ck(self.L.smc_set_exact_pow(self.ctx, int(bool(enable))), "smc_set_exact_pow")

    def set_in_phase(self, enable=True):
        """Let homogeneous Michaelis-Menten Metropolis sweeps run their waves in phase (include/smc_hip.h: smc_set_in_phase)."""
        if "smc_set_in_phase" in B.MISSING:        # A/B build of an older revision (SMC_HIP_LIB)
            return
        self._ck(self.L.smc_set_in_phase(self.ctx, int(bool(enable))), "smc_set_in_phase")

    def set_cost_order(self, enable=True):
        """Cost-ordered, in-phase hand-out of heterogeneous Metropolis sweeps (include/smc_hip.h: smc_set_cost_order)."""
        if "smc_set_cost_order" in B.MISSING:      # A/B build of an earlier revision (SMC_HIP_LIB)
            return
        self._ck(self.L.smc_set_cost_order(self.ctx, int(bool(enable))), "smc_set_cost_order")

    def set_fast_tail(self, enable=True):
        """Hand-written lone-chain attempt loop of the Michaelis-Menten kernel (include/smc_hip.h: smc_set_fast_tail)."""
        if "smc_set_fast_tail" in B.MISSING:       # A/B build of an earlier revision (SMC_HIP_LIB)
            return
        self._ck(self.L.smc_set_fast_tail(self.ctx, int(bool(enable))), "smc_set_fast_tail")

    def set_stiff_first(self, enable=True):
        """Hand the predictably long Michaelis-Menten solves out first (include/smc_hip.h: smc_set_stiff_first)."""
        if "smc_set_stiff_first" in B.MISSING:     # A/B build of a revision before the stiff list (SMC_HIP_LIB)
            return
        self._ck(self.L.smc_set_stiff_first(self.ctx, int(bool(enable))), "smc_set_stiff_first")

    def set_resampling(self, scheme):
        """"residual_systematic" (default, Micmem_SMC_main.py:147-184) or "systematic"."""
        k = B.RESAMPLING[scheme] if isinstance(scheme, str) else int(scheme)
        self._ck(self.L.smc_set_resampling(self.ctx, k), "smc_set_resampling")

    # ---- movement ------------------------------------------------------------------------------
    def upload_particles(self, which, aos):
        aos = _f64(aos)
        assert aos.ndim == 2 and aos.shape[1] == self.dim
        self._ck(self.L.smc_upload_particles(self.ctx, which, _dp(aos), aos.shape[0]), "smc_upload_particles")

    def download_particles(self, which, n=None, pinned=False):
        """(n, d) particles of set `which`.  pinned: the array lives in page-locked host memory (one DMA transfer; see _PinnedPool)."""
        n = self.n_local if n is None else n
        out = pinned_empty((n, self.dim)) if pinned else np.empty((n, self.dim))
        self._ck(self.L.smc_download_particles(self.ctx, which, _dp(out), n), "smc_download_particles")
        return out

    def upload_lk(self, which, lk):
        lk = _f64(lk)
        self._ck(self.L.smc_upload_lk(self.ctx, which, _dp(lk), lk.shape[0]), "smc_upload_lk")

    def download_lk(self, which, n=None, pinned=False):
        n = self.n_local if n is None else n
        out = pinned_empty((n,)) if pinned else np.empty(n)
        self._ck(self.L.smc_download_lk(self.ctx, which, _dp(out), n), "smc_download_lk")
        return out

    def download_accept_flags(self):
        out = np.empty(self.n_local, dtype=np.uint8)
        self._ck(self.L.smc_download_accept_flags(self.ctx, out.ctypes.data_as(B.c_u8p), self.n_local),
                 "smc_download_accept_flags")
        return out

    def debug_set_order(self, order, patience=0):
        """Probes: hand the index-ordered items of every MM likelihood sweep out in this order (None: off); smc_debug_set_order."""
        if order is None:
            self._ck(self.L.smc_debug_set_order(self.ctx, None, 0, 0), "smc_debug_set_order")
            return
        o = np.ascontiguousarray(order, dtype=np.int32)
        self._ck(self.L.smc_debug_set_order(self.ctx, o.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), o.size, int(patience)), "smc_debug_set_order")

    def download_item_info(self, n=None):
        """(n_ex, n) records of the last Michaelis-Menten sweep: attempts | cancelled << 29 | failed << 30 (diagnostics)."""
        n = self.n_local if n is None else int(n)
        out = np.empty((self.model[1], n), dtype=np.int32)
        self._ck(self.L.smc_download_item_info(self.ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n), "smc_download_item_info")
        return out

    def commit_filt_to_pred(self):
        self._ck(self.L.smc_commit_filt_to_pred(self.ctx), "smc_commit_filt_to_pred")

    def sample_prior_device(self, seed, global_offset=0):
        self._ck(self.L.smc_sample_prior_device(self.ctx, int(seed), int(global_offset)), "smc_sample_prior_device")

    # ---- likelihood ----------------------------------------------------------------------------
    def loglik(self, which=SMC_SET_PRED):
        nf, att = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self.L.smc_loglik(self.ctx, which, ctypes.byref(nf), ctypes.byref(att)), "smc_loglik")
        return {"n_failed": nf.value, "rk_attempts": att.value}

    def loglik_host(self, particle, want_pred=False):
        particle = _f64(particle)
        assert particle.ndim == 2 and particle.shape[1] == 3
        n = particle.shape[0]
        lk = np.empty(n)
        pred = np.empty((n, self.model[1], self.model[2])) if want_pred else None
        nf, att = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self.L.smc_mm_loglik_host(self.ctx, _dp(particle), n, _dp(lk), _dp(pred) if want_pred else None,
                                           ctypes.byref(nf), ctypes.byref(att)), "smc_mm_loglik_host")
        return lk, pred, {"n_failed": nf.value, "rk_attempts": att.value}

    # ---- weights / ESS -------------------------------------------------------------------------
    def max_lk_local(self):
        v = ctypes.c_double(0)
        self._ck(self.L.smc_max_lk_local(self.ctx, ctypes.byref(v)), "smc_max_lk_local")
        return v.value

    def ess_partials(self, max_lk, gms):
        gms = _f64(gms)
        k = gms.shape[0]
        sw, sw2 = np.empty(k), np.empty(k)
        self._ck(self.L.smc_ess_partials(self.ctx, float(max_lk), _dp(gms), k, _dp(sw), _dp(sw2)), "smc_ess_partials")
        return sw, sw2

    # ---- the same stages reduced over all ranks on the device (RCCL in place, one read-back) ------
    def max_lk_global(self):
        v = ctypes.c_double(0)
        self._ck(self.L.smc_max_lk_global(self.ctx, ctypes.byref(v)), "smc_max_lk_global")
        return v.value

    def ess_partials_global(self, max_lk, gms):
        gms = _f64(gms)
        k = gms.shape[0]
        sw, sw2 = np.empty(k), np.empty(k)
        self._ck(self.L.smc_ess_partials_global(self.ctx, float(max_lk), _dp(gms), k, _dp(sw), _dp(sw2)),
                 "smc_ess_partials_global")
        return sw, sw2

    def ess_search_global(self, gms, with_max=True):
        """max(lk) and the weight sums of up to 32 candidate increments, all ranks, one synchronisation
        (include/smc_hip.h: smc_ess_search_global) -> (max_lk, sum_w, sum_w2)."""
        gms = _f64(gms)
        k = gms.shape[0]
        sw, sw2 = np.empty(k), np.empty(k)
        m = ctypes.c_double(0)
        self._ck(self.L.smc_ess_search_global(self.ctx, _dp(gms), k, int(bool(with_max)), ctypes.byref(m), _dp(sw), _dp(sw2)),
                 "smc_ess_search_global")
        return m.value, sw, sw2

    def resample_global(self, max_lk, gm, sum_w, wrand, first_step):
        o, cs = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self.L.smc_resample_global(self.ctx, float(max_lk), float(gm), float(sum_w), float(wrand),
                                            int(bool(first_step)), ctypes.byref(o), ctypes.byref(cs)), "smc_resample_global")
        return {"n_offspring": o.value, "count_sum": cs.value}

    def resample_enqueue(self, max_lk, gm, sum_w, wrand, first_step):
        """Resampling without a host synchronisation (one rank; include/smc_hip.h: smc_resample_enqueue); resample_result() later."""
        self._ck(self.L.smc_resample_enqueue(self.ctx, float(max_lk), float(gm), float(sum_w), float(wrand), int(bool(first_step))),
                 "smc_resample_enqueue")

    def resample_result(self):
        o, cs = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self.L.smc_resample_result(self.ctx, ctypes.byref(o), ctypes.byref(cs)), "smc_resample_result")
        return {"n_offspring": o.value, "count_sum": cs.value}

    def mh_iteration_device_rng(self, gamma, mhstep_ratio, w_cov, seed, stream, global_offset=0):
        """One fused Metropolis iteration (moments -> cov_m -> factor -> propose -> solve -> accept -> counts), all ranks
        together.  Accept counts and n_failed are totals over all ranks."""
        w_cov = _f64(w_cov, (self.dim, self.dim))
        o = self._mh_out()
        cov = np.empty((self.dim, self.dim))
        self._ck(self.L.smc_mh_iteration_device_rng(self.ctx, float(gamma), float(mhstep_ratio), _dp(w_cov), int(seed),
                                                    int(stream), int(global_offset), *[ctypes.byref(x) for x in o], _dp(cov)),
                 "smc_mh_iteration_device_rng")
        return {"accepted_now": o[0].value, "accepted_ever": o[1].value, "n_failed": o[2].value,
                "rk_attempts": o[3].value, "cov_m": cov}

    def mh_sweeps_device_rng(self, gamma, mhstep_ratio, w_cov, seed, stream0, n_iter, thr_stop, thr_halve, global_offset=0):
        """A batch of n_iter fused Metropolis iterations with the loop control (break / halve mhstep_ratio, Micmem_SMC_main.py:
        243-249) on the device and ONE synchronisation (include/smc_hip.h: smc_mh_sweeps_device_rng).  Returns how many ran,
        whether the loop ended by its break, the mhstep_ratio a following batch starts with, and one record per iteration that
        ran (counts are totals over all ranks, rk_attempts this rank's)."""
        w_cov = _f64(w_cov, (self.dim, self.dim))
        k = int(n_iter)
        nd, st, rn = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_double(0)
        an, ae, nf, at = (np.zeros(k, dtype=np.int64) for _ in range(4))
        ru, cov = np.zeros(k), np.zeros((k, self.dim, self.dim))
        sc = np.zeros((k, B.SMC_SWEEP_COUNTER_WORDS), dtype=np.int64)
        self._ck(self.L.smc_mh_sweeps_device_rng(self.ctx, float(gamma), float(mhstep_ratio), _dp(w_cov), int(seed), int(stream0), k,
                                                 float(thr_stop), float(thr_halve), int(global_offset), ctypes.byref(nd),
                                                 ctypes.byref(st), ctypes.byref(rn), an.ctypes.data_as(B.c_i64p),
                                                 ae.ctypes.data_as(B.c_i64p), nf.ctypes.data_as(B.c_i64p),
                                                 at.ctypes.data_as(B.c_i64p), _dp(ru), _dp(cov), sc.ctypes.data_as(B.c_i64p)),
                 "smc_mh_sweeps_device_rng")
        its = [{"accepted_now": int(an[i]), "accepted_ever": int(ae[i]), "n_failed": int(nf[i]), "rk_attempts": int(at[i]),
                "mhstep_ratio": float(ru[i]), "cov_m": cov[i].copy()} for i in range(nd.value)]
        if self.model is not None and self.model[0] == "methanation":     # K8's work and completeness counters per sweep (the driver's books)
            for i, it in enumerate(its):
                it["counters"] = {nm: int(sc[i, q]) for q, nm in enumerate(B.SWEEP_COUNTER_NAMES)}
        return {"n_done": nd.value, "stopped": bool(st.value), "ratio_next": rn.value, "iterations": its}

    def proposal_factor_device(self, w_cov):
        """cov_m = np.cov(p_filt.T, bias=True) * w_cov over all ranks and its multivariate_normal factor, both on the device."""
        w_cov = _f64(w_cov, (self.dim, self.dim))
        cov, xf = np.empty((self.dim, self.dim)), np.empty((self.dim, self.dim))
        self._ck(self.L.smc_proposal_factor_device(self.ctx, _dp(w_cov), _dp(cov), _dp(xf)), "smc_proposal_factor_device")
        return cov, xf

    def mh_iteration_last_transform(self):
        out = np.empty((self.dim, self.dim))
        self._ck(self.L.smc_mh_iteration_last_transform(self.ctx, _dp(out)), "smc_mh_iteration_last_transform")
        return out

    # ---- resampling ----------------------------------------------------------------------------
    def resample_phase1(self, max_lk, gm, sum_w):
        r, c = ctypes.c_double(0), ctypes.c_int64(0)
        self._ck(self.L.smc_resample_phase1(self.ctx, float(max_lk), float(gm), float(sum_w), ctypes.byref(r),
                                            ctypes.byref(c)), "smc_resample_phase1")
        return r.value, c.value

    def resample_phase2(self, max_lk, gm, sum_w, residual_prefix, wrand):
        o = ctypes.c_int64(0)
        self._ck(self.L.smc_resample_phase2(self.ctx, float(max_lk), float(gm), float(sum_w), float(residual_prefix),
                                            float(wrand), ctypes.byref(o)), "smc_resample_phase2")
        return o.value

    def download_offspring(self):
        out = np.empty(self.n_local, dtype=np.int64)
        self._ck(self.L.smc_download_offspring(self.ctx, out.ctypes.data_as(B.c_i64p), self.n_local),
                 "smc_download_offspring")
        return out

    def resample_phase3(self, out_base_all, offspring_all, first_step):
        b = np.ascontiguousarray(out_base_all, dtype=np.int64)
        o = np.ascontiguousarray(offspring_all, dtype=np.int64)
        assert b.shape == o.shape == (self.world,)
        self._ck(self.L.smc_resample_phase3(self.ctx, b.ctypes.data_as(B.c_i64p), o.ctypes.data_as(B.c_i64p),
                                            int(bool(first_step))), "smc_resample_phase3")
        if self._peer_barrier is not None:      # loopback rehearsal: pack | barrier | pull | barrier
            self._peer_barrier()
            self._ck(self.L.smc_resample_phase3_pull(self.ctx), "smc_resample_phase3_pull")
            self._peer_barrier()

    def debug_set_local_peers(self, engines, rank, barrier):
        """Loopback rehearsal of the multi-rank path on one device (see include/smc_hip.h)."""
        arr = (B.c_ctx * len(engines))(*[e.ctx for e in engines])
        self._ck(self.L.smc_debug_set_local_peers(self.ctx, arr, int(rank), len(engines)), "smc_debug_set_local_peers")
        self.rank, self.world = int(rank), len(engines)
        self._peer_barrier = barrier

    def debug_peer_collectives(self, enable=True):
        """Loopback rehearsal with the collectives inside the engine (include/smc_hip.h: smc_debug_peer_collectives): the
        *_global entry points reduce among the local peers; the Python-side pull of resample_phase3 is no longer needed."""
        self._ck(self.L.smc_debug_peer_collectives(self.ctx, int(bool(enable))), "smc_debug_peer_collectives")
        if enable:
            self._peer_barrier = None

    # ---- moments -------------------------------------------------------------------------------
    def moment_sums_local(self):
        out = np.empty(self.dim)
        self._ck(self.L.smc_moment_sums_local(self.ctx, _dp(out)), "smc_moment_sums_local")
        return out

    def moment_centered_local(self, mean):
        mean = _f64(mean, (self.dim,))
        out = np.empty((self.dim, self.dim))
        self._ck(self.L.smc_moment_centered_local(self.ctx, _dp(mean), _dp(out)), "smc_moment_centered_local")
        return out

    # ---- MH ------------------------------------------------------------------------------------
    def reset_accept_flags(self):
        self._ck(self.L.smc_reset_accept_flags(self.ctx), "smc_reset_accept_flags")

    def _mh_out(self):
        return [ctypes.c_int64(0) for _ in range(4)]

    def mh_step_host_rng(self, gamma, mhstep_ratio, noise, rr):
        noise = _f64(noise, (self.n_local, self.dim))
        rr = _f64(rr, (self.n_local,))
        o = self._mh_out()
        self._ck(self.L.smc_mh_step_host_rng(self.ctx, float(gamma), float(mhstep_ratio), _dp(noise), _dp(rr),
                                             self.n_local, *[ctypes.byref(x) for x in o]), "smc_mh_step_host_rng")
        return {"accepted_now": o[0].value, "accepted_ever": o[1].value, "n_failed": o[2].value,
                "rk_attempts": o[3].value}

    def mh_step_device_rng(self, gamma, mhstep_ratio, transform, seed, stream, global_offset=0):
        transform = _f64(transform, (self.dim, self.dim))
        o = self._mh_out()
        self._ck(self.L.smc_mh_step_device_rng(self.ctx, float(gamma), float(mhstep_ratio), _dp(transform), int(seed),
                                               int(stream), int(global_offset), *[ctypes.byref(x) for x in o]),
                 "smc_mh_step_device_rng")
        return {"accepted_now": o[0].value, "accepted_ever": o[1].value, "n_failed": o[2].value,
                "rk_attempts": o[3].value}

    def set_debug_capture(self, enable=True):
        self._ck(self.L.smc_set_debug_capture(self.ctx, int(enable)), "smc_set_debug_capture")

    def download_debug_proposals(self):
        n = self.n_local
        aos, lk2 = np.empty((n, self.dim)), np.empty(n)
        p0, r = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8)
        self._ck(self.L.smc_download_debug_proposals(self.ctx, _dp(aos), _dp(lk2), p0.ctypes.data_as(B.c_u8p),
                                                     r.ctypes.data_as(B.c_u8p), n), "smc_download_debug_proposals")
        return aos, lk2, p0, r

    def debug_rccl_self_exchange(self, row, cnt, dst_row):
        self._ck(self.L.smc_debug_rccl_self_exchange(self.ctx, int(row), int(cnt), int(dst_row)), "smc_debug_rccl_self_exchange")

    # ---- collectives (RCCL) --------------------------------------------------------------------
    @staticmethod
    def comm_get_unique_id() -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        st = lib().smc_comm_get_unique_id(buf)
        if st != 0:
            msg = lib().smc_last_error(None)
            raise SmcError(f"smc_comm_get_unique_id: {msg.decode() if msg else ''}")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self.L.smc_comm_init(self.ctx, buf, int(rank), int(world)), "smc_comm_init")
        self.rank, self.world = int(rank), int(world)

    def comm_info(self):
        """RCCL's own view of this context's communicator: {"count", "user_rank", "device"} (count 0: none)."""
        n, r, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._ck(self.L.smc_comm_info(self.ctx, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "smc_comm_info")
        return {"count": n.value, "user_rank": r.value, "device": d.value}

    def comm_allreduce_sum_f64(self, x):
        x = _f64(x).copy()
        self._ck(self.L.smc_comm_allreduce_sum_f64(self.ctx, _dp(x), x.size), "smc_comm_allreduce_sum_f64")
        return x

    def comm_allreduce_max_f64(self, x):
        x = _f64(x).copy()
        self._ck(self.L.smc_comm_allreduce_max_f64(self.ctx, _dp(x), x.size), "smc_comm_allreduce_max_f64")
        return x

    def comm_allreduce_sum_i64(self, x):
        x = np.ascontiguousarray(x, dtype=np.int64).copy()
        self._ck(self.L.smc_comm_allreduce_sum_i64(self.ctx, x.ctypes.data_as(B.c_i64p), x.size),
                 "smc_comm_allreduce_sum_i64")
        return x

    def comm_allgather_f64(self, x):
        x = _f64(x)
        out = np.empty((self.world,) + x.shape)
        self._ck(self.L.smc_comm_allgather_f64(self.ctx, _dp(x), x.size, _dp(out)), "smc_comm_allgather_f64")
        return out

    def comm_allgather_i64(self, x):
        x = np.ascontiguousarray(x, dtype=np.int64)
        out = np.empty((self.world,) + x.shape, dtype=np.int64)
        self._ck(self.L.smc_comm_allgather_i64(self.ctx, x.ctypes.data_as(B.c_i64p), x.size,
                                               out.ctypes.data_as(B.c_i64p)), "smc_comm_allgather_i64")
        return out

    def comm_barrier(self):
        self._ck(self.L.smc_comm_barrier(self.ctx), "smc_comm_barrier")

    # ---- timing --------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._ck(self.L.smc_timing_enable(self.ctx, int(on)), "smc_timing_enable")

    def timing_reset(self):
        self._ck(self.L.smc_timing_reset(self.ctx), "smc_timing_reset")

    def timing_get(self):
        out = {}
        for which, name in B.TIMING_NAMES.items():
            n, ms = ctypes.c_int64(0), ctypes.c_double(0)
            self._ck(self.L.smc_timing_get(self.ctx, which, ctypes.byref(n), ctypes.byref(ms)), "smc_timing_get")
            out[name] = {"launches": n.value, "ms": ms.value}
        return out

    def work_totals(self):
        """Device-counted Michaelis-Menten work since timing_reset(): solves that produced their outputs, RK45 attempts, solve
        launches with work, speculative launches that found the loop ended (smc_work_totals)."""
        if "smc_work_totals" in B.MISSING:         # A/B build of an older revision (SMC_HIP_LIB)
            return None
        out = (ctypes.c_int64 * 4)()
        self._ck(self.L.smc_work_totals(self.ctx, out), "smc_work_totals")
        return {"solved_items": out[0], "rk_attempts": out[1], "solve_launches": out[2], "noop_launches": out[3]}
