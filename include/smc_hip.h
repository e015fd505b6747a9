/*
 * include/smc_hip.h -- C ABI of libsmc_hip.so, the MI355X (gfx950) engine for the particle
 * inner loop of adaptive likelihood-tempered SMC.
 *
 * The reference (maruchitatsuki/python-based-Sequential-Monte-Carlo-method-with-likelihood-tempering)
 * has no FFI: its boundary is a set of Python names.  Each entry point
 * below names the reference statements it replaces (file:line relative to the reference root).
 * The Python host side (python-based-..._amd/binding.py) binds exactly these symbols with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, non-zero on error, and
 *     smc_last_error(ctx) returns the message (ctx may be NULL for creation errors);
 *   - the caller owns host buffers (C-contiguous float64 / int64 / uint8); the library owns all
 *     device memory inside the opaque context; one context per (process, device); a context is
 *     not thread-safe;
 *   - particles are AoS (N,d) at this edge, exactly like the reference's p_pred / p_filt arrays,
 *     and SoA d x N in HBM (transposed by the upload / download calls);
 *   - the two particle sets of the reference are addressed by SMC_SET_PRED (p_pred with lk,
 *     Micmem_settings.py:85; Micmem_SMC_main.py:98) and SMC_SET_FILT (p_filt with lk1,
 *     Micmem_settings.py:118,127);
 *   - every arithmetic step is IEEE float64; integers are used only for offspring counts and flags;
 *   - multi-GPU: one context per rank, each owning the contiguous block
 *     [rank*n_local, (rank+1)*n_local) of the global particle order.  Entry points that take
 *     "_local"/"_global" arguments compute this rank's partial; the host combines partials with the
 *     smc_comm_* collectives (RCCL).  With one rank the same calls are used with world size 1.
 */
#ifndef SMC_HIP_H
#define SMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMC_ABI_VERSION 3    /* 2: smc_meth_sweep_check writes FIVE words (round 3 added the cancelled count); smc_mh_sweeps_device_rng
                              * 3: smc_work_totals; smc_mh_sweeps_device_rng takes the methanation model too and returns the sweep counters;
                              *    smc_meth_dae_host's stats have five words (round 5) */
#define SMC_MAX_DIM 8        /* parameters per particle (3 for Michaelis-Menten, 5 for methanation) */
#define SMC_MAX_ESS_CAND 16  /* tempering candidates evaluated by one smc_ess_partials call */
#define SMC_MAX_RANKS 64

#define SMC_SET_PRED 0
#define SMC_SET_FILT 1

#define SMC_PRIOR_UNIFORM 0 /* {"dist":"uniform","low","high"}  Micmem_settings.py:63-67 */
#define SMC_PRIOR_NORMAL 1  /* {"dist":"normal","mu","sigma"}   Micmem_settings.py:55-59 */
#define SMC_PRIOR_FLAT 2    /* no factor in cal_prior (sigma under normal_pred without taylor,
                             * methanation_functions.py:132-138 multiplies num_est_params-1 densities) */

/* what the Metropolis acceptance does with the prior */
#define SMC_PRIOR_MODE_MASK 0        /* pp = exp(px*g) * p0            Micmem_SMC_main.py:224-233, SMC_methanation_main.py:376-389 */
#define SMC_PRIOR_MODE_RATIO_MASK 1  /* pp = exp(px*g) * (p0_2/p0_1) * p0   SMC_methanation_main.py:320-349 (normal_pred, taylor) */
#define SMC_PRIOR_MODE_RATIO 2       /* pp = exp(px*g) * (p0_2/p0_1), no reset of proposals   :358-374 (normal_pred) */

typedef struct smc_ctx smc_ctx;

int smc_abi_version(void);
/* NULL ctx: message of the last failed smc_create / context-less call in this thread. */
const char *smc_last_error(const smc_ctx *ctx);

/* Context with capacity for n_local particles of dimension dim on HIP device `device`.
 * n_global = world * n_local (pass n_local for a single GPU).  Replaces the buffer allocations of
 * Micmem_settings.py:85,118-127. */
int smc_create(smc_ctx **out, int device, int64_t n_local, int64_t n_global, int dim);
void smc_destroy(smc_ctx *ctx);
int smc_synchronize(smc_ctx *ctx);
/* Name/arch of the device the context runs on (e.g. "gfx950"). */
int smc_device_info(smc_ctx *ctx, char *name, int name_len, char *arch, int arch_len, int *cu_count);

/* ---- model + prior ------------------------------------------------------------------------ */
/* Michaelis-Menten data set (Micmem_settings.py:103-115) and solver tolerances (SciPy defaults
 * rtol=1e-3, atol=1e-6 of the solve_ivp call at Micmem_likelihood.py:24-30).  t and P_obs are
 * n_ex x n_t row-major, S0 has n_ex entries.  n_ex <= 16, n_t <= 256. */
int smc_set_model_mm(smc_ctx *ctx, const double *t, const double *P_obs, const double *S0, int n_ex, int n_t,
                     int est_sigma, double sigma_fixed, double rtol, double atol);
/* Methanation model (configs 4-5) for the resident sets: smc_loglik and smc_mh_step_* then evaluate
 * cal_parallel_new (methanation_functions.py:44-65) per particle: the particle's `dim` estimated values are
 * scattered into the 9-vector base_params at est_position (:80), my_model integrates the DAE of each of the
 * n_data experiments (K8, parity unpinned against IDA) and my_loglike compares the outlet flows with obs.
 * cond: n_data x 10 (Ca_in,Cb_in,Cc_in,Cd_in,Ce_in,T_in,T_jacket,u_in,void,dz = the first ten entries of p0,
 * methanation_set_likelihood.py:164); guess: n_data x 357; obs: 5 x n_data; base_params: 9. */
int smc_set_model_methanation(smc_ctx *ctx, const double *cond, const double *guess, const double *obs, int n_data,
                              const double *base_params, const int *est_position, int est_sigma, double sigma_fixed,
                              double tf, double rtol, double atol);
/* User model (the reference's README.md:4 "modify for your problem", i.e. a new
 * Micmem_likelihood.py): `source` is HIP device code defining, for the state y[n_states] of an ODE,
 *   __device__ void   smc_user_y0 (const double *theta, const double *cond, double *y);            initial state at t[e][0]
 *   __device__ void   smc_user_rhs(double t, const double *y, const double *theta, const double *cond, double *dydt);
 *   __device__ double smc_user_obs(double t, const double *y, const double *theta, const double *cond);   what obs is compared with
 * (theta: the particle's `dim` values; cond: the n_cond numbers of experiment e).  It is compiled at run time (hiprtc,
 * gfx950) into a kernel that restates solve_ivp(RK45, t_eval = t[e], rtol, atol) and the Gaussian log-likelihood of
 * Micmem_likelihood.py:17-33,62-73 with sigma = the last parameter (est_sigma) or sigma_fixed; smc_loglik and
 * smc_mh_step_* then use it.  t, obs: n_ex x n_t; cond: n_ex x n_cond.  A source that does not compile fails with
 * hiprtc's log in smc_last_error.  smc_user_model_check only compiles (no GPU needed): 0 ok, 1 compile error (log).
 * Optional fourth ingredient, a COST HINT (a source that contains the name must define it):
 *   __device__ double smc_user_cost(const double *theta);      rough number of RK45 step attempts of one solve with theta
 * Sweeps then hand the solves of particles above 220 attempts out before the index-ordered ones and run those above 3700
 * one per wave on wave-uniform operands (the built-in Michaelis-Menten kernel's stiff list and solo phase, smc_set_stiff_first
 * switches both off) - a sweep over a prior population is bounded by its longest serial solve, which should start first -
 * and heterogeneous Metropolis sweeps of 16 384 particles or more are handed out by cost class and in phase (smc_set_cost_order).
 * The hint changes the order of independent solves only, never a result; it may be crude, and NaN counts as cheap.
 * The functions may call   double smc_div(double a, double b)   for a / b: the source is compiled twice, in two namespaces
 * (so: device functions and constants only, nothing extern "C") - once with the six-operation division of the built-in
 * kernel (v_rcp, one Newton step, one correction: equal to a / b for normal operands and quotients up to a last bit in about one pair of 2^44, NaN where a / b
 * needs a subnormal or infinite divisor or a * (1 / b) overflows), once with IEEE division; a step attempt runs on the first
 * and is repeated on the second whenever its error norm is not finite.  smc_user_y0, smc_user_obs and smc_user_cost always
 * run with IEEE division. */
#define SMC_USER_MAX_STATES 8
int smc_set_model_user(smc_ctx *ctx, const char *source, int n_states, const double *t, const double *obs, const double *cond,
                       int n_ex, int n_t, int n_cond, int est_sigma, double sigma_fixed, double rtol, double atol);
int smc_user_model_check(const char *source, int n_states, int dim, char *log, int log_cap);
/* Writes exactly what hiprtc is given for `source` into the existing directory `dir`: smc_user_model.hip (the user's functions
 * followed by the library's kernel) and the four headers it includes - to read, or to compile off line
 * (`hipcc --offload-arch=gfx950 -O3 -ffp-contract=on -fno-fast-math -I dir -S dir/smc_user_model.hip`; the build container's
 * tests run the compiler's uniformity analysis on it).  0 ok, 1 a file could not be written, 2 bad arguments.  No GPU needed. */
int smc_user_model_dump_source(const char *source, int n_states, int dim, const char *dir);
/* Methanation model: device-counted work of the LAST smc_loglik / smc_mh_step_* call (the counts the K8
 * roofline is built from): out = {accepted BDF steps, Newton iterations, Jacobian factorisations, failed solves}. */
int smc_meth_sweep_counters(smc_ctx *ctx, int64_t out[4]);
/* Completeness of the LAST methanation sweep: out = {DAE solves asked for (live (particle, experiment) pairs), solves
 * finished, live items whose status was still the pre-sweep poison value when the likelihood was formed, waves that were
 * incomplete at a dequeue, solves NOT started because their proposal was already certain to be rejected}.  A sweep with
 * out[1] + out[4] != out[0] or out[2], out[3] != 0 makes smc_loglik / smc_mh_step_* fail (the reference's counterpart is
 * the bare except of methanation_set_likelihood.py:234-254: there a lost solve would surface as an exception at ray.get).
 * Exact early rejection (smc_set_early_reject, default on) also covers the methanation sweeps: the accept test
 * exp((lk2 - lk1) * gamma) >= rr (SMC_methanation_main.py:376-383) has lk1 and rr fixed beforehand and my_loglike only
 * falls with every experiment added to it, so a proposal that fails the test on the experiments finished SO FAR is
 * rejected exactly as the full computation would reject it and its remaining DAE solves are skipped (out[4]); the sweep
 * runs experiment-major so that a proposal's experiments come up one after the other.  p_filt, lk1, accept flags and
 * counts are unchanged; lk2 of such a proposal is never formed (debug capture switches the feature off). */
int smc_meth_sweep_check(smc_ctx *ctx, int64_t out[5]);
/* Outlet flows (n x n_data x 5, the F_k of methanation_set_likelihood.py:204-208; -10000 where the solve failed, :244-249)
 * and solver status (n x n_data: 0 solved, 1 given up, -1 not solved in this sweep = masked proposal) of the LAST sweep -
 * what my_model returns per particle before my_loglike reduces it (the reference keeps them as C_l_ for its plots). */
int smc_meth_download_solves(smc_ctx *ctx, double *flows, int32_t *status, int64_t n);
/* Independent priors, one per parameter: kind[i] in {SMC_PRIOR_UNIFORM, SMC_PRIOR_NORMAL, SMC_PRIOR_FLAT}; (a,b) =
 * (low,high) or (mu,sigma).  Used by the support mask of cal_prior (Micmem_SMC_main.py:60-90,
 * 224-228) and by smc_sample_prior_device. */
int smc_set_prior(smc_ctx *ctx, const int *kind, const double *a, const double *b, int dim);
/* SMC_PRIOR_MODE_*: default SMC_PRIOR_MODE_MASK (the live branch of both reference drivers). */
int smc_set_prior_mode(smc_ctx *ctx, int mode);

/* Resampling scheme of smc_resample_phase1/2 (BASELINE.json names systematic resampling; the reference implements
 * only the residual-systematic variant, Micmem_SMC_main.py:147-184, which stays the default):
 *   SMC_RESAMPLE_RESIDUAL_SYSTEMATIC  trunc(N w_i) copies + systematic draws on the residuals
 *   SMC_RESAMPLE_SYSTEMATIC           systematic draws on the weights themselves (thresholds (u + k)/N)
 *   SMC_RESAMPLE_MULTINOMIAL          N iid draws: the thresholds are the order statistics of N uniforms, produced on
 *                                     the device as normalised partial sums of N+1 exponentials (Philox, seeded by
 *                                     the bits of `wrand`); offspring land sorted by ancestor like the other schemes */
#define SMC_RESAMPLE_RESIDUAL_SYSTEMATIC 0
#define SMC_RESAMPLE_SYSTEMATIC 1
#define SMC_RESAMPLE_MULTINOMIAL 2
/* Exact early rejection in the Metropolis sweeps of the Michaelis-Menten model (default: on).  The accept test
 * exp((lk2-lk1)*gamma)*p0 >= rr (Micmem_SMC_main.py:231-236) has lk1 and rr fixed before the proposal is solved, and lk2 = sum
 * over the experiments of c0 - sum(residual^2)/(2 sigma^2) only decreases while a solve accumulates residuals.  A solve
 * whose proposal fails the test even with the sums accumulated SO FAR (0 for experiments still running) is stopped: the
 * proposal is rejected exactly as the completed computation would reject it - p_filt, lk1, accept flags and counts are
 * unchanged; only rk_attempts is smaller, a step-size underflow in the skipped part of a solve cannot be reported (the
 * reference would raise there; none occurs on this model), and the lk2 of such a proposal is never formed (the debug
 * capture therefore switches the feature off).  It removes the long solves that dominate the early tempering steps:
 * they are proposals with Vmax/Km in the thousands whose other experiments already rule them out. */
int smc_set_early_reject(smc_ctx *ctx, int enable);
/* Michaelis-Menten sweeps hand the predictably long solves (Vmax > 60 Km: RK45 runs on its stability limit for ~3.7 Vmax/Km
 * attempts) out BEFORE the index-ordered items (default: on), so that the serial chains that bound a sweep start at its
 * beginning, and run the stiffest of them (Vmax > 1000 Km, at most one solve per wave of the sweep's grid) SOLO: one wave per solve on
 * wave-uniform operands from the first attempt (0.33 - smc_set_fast_tail - instead of ~0.55 us per attempt while the rest of the population
 * keeps the other lanes busy).  The order in which independent (particle, experiment) solves run - and the lane count they
 * run on - changes no result: the reference's one-Ray-task-per-particle fan-out (Micmem_likelihood.py:83-87) leaves it to its
 * scheduler too; 0 restores round 2's plain index order (A/B timing, tests).  Methanation: the same switch selects the
 * misfit order of the experiments in the early-rejection sweeps (smc_meth_sweep_check). */
int smc_set_stiff_first(smc_ctx *ctx, int enable);
/* Michaelis-Menten Metropolis sweeps over a heterogeneous population hand their index-ordered solves out by cost class
 * (Vmax / Km of the proposal, four classes per octave, counting sort on the device) and in phase, so that the 64 solves a wave
 * starts together are alike: same results (the order of independent solves is free, as for smc_set_stiff_first), 20-25 % less
 * time for the sweeps of the middle of a run.  Default: on; needs smc_set_in_phase on.  0: A/B timing, tests. */
int smc_set_cost_order(smc_ctx *ctx, int enable);
/* A Michaelis-Menten solve that runs alone in its wave (a solo solve, the last survivor of a sweep) performs the attempts that
 * neither produce an output nor meet a special case in a hand-written instruction sequence (csrc/mm_rk45.h:
 * mm_fast_uniform_attempts: ~140 instead of ~190 instructions per attempt, the Dormand-Prince tableau resident in scalar
 * registers); every other attempt goes through the compiled step function.  Default: on.  The results are the same bit for
 * bit (tests/test_gpu_parity.py compares on and off); 0 is for that comparison and for A/B timing.  Parity mode
 * (smc_set_exact_pow) never uses it. */
int smc_set_fast_tail(smc_ctx *ctx, int enable);
/* Michaelis-Menten Metropolis sweeps over a homogeneous population (the previous sweep of the context had fewer than one
 * (particle, experiment) solve in 20 000 with more than 64 RK45 attempts - counted on the device) run their waves IN PHASE
 * (default: on): a wave waits up to 12 attempts for all 64 lanes to finish before it starts its next 64 items, so that the
 * lanes stay at the same point of their trajectories and the dense-output loop runs as long as the average lane needs, not
 * the busiest (csrc/solve_sched.h).  Scheduling only: no result changes.  0 = hand out as soon as 24 lanes are idle, always. */
int smc_set_in_phase(smc_ctx *ctx, int enable);
/* Parity mode of the Michaelis-Menten step controller (default: off).  SciPy evaluates error_norm ** -0.2 (rk.py:155,169)
 * and x ** (1 / 5) (common.py:130) with libm's pow and the DOUBLE exponents -0.2 / 0.2 (= 1/5 + 1.1e-17).  The fast
 * device form (hardware log2 / exp2 seed + one correction, <= 1.5 ulp about the fifth root) differs from that in the last
 * bit of many arguments; on RK45's stability limit one such bit can flip one accept / reject decision and the solve then
 * follows another, equally valid step sequence (logL equal to 1e-8 ... 1e-6 only).  enable != 0 finishes the fast value to
 * the CORRECTLY ROUNDED pow(x, -0.2) / pow(x, 0.2) (double-double residual, csrc/pow_fifth_exact.h; checked against a 113-bit
 * reference on the CPU) - what libm returns except for about 8 of 10^4 arguments where glibc's own pow is not correctly
 * rounded - AND evaluates the Runge-Kutta stages, the error estimate and select_initial_step with every product and sum
 * rounded separately, in NumPy's order (no fused multiply-add; mm_rk45.h: rk_attempt_core_exact), as the CPU checker does:
 * the device then walks the checker's step sequence attempt for attempt, also in the stiff band
 * (test_stiff_band_parity_and_its_tolerance).  About 60 more operations per attempt.  run_smc switches it on with
 * rng="numpy" (the reference's stream) and the drop-in sim_particle uses it; the device-RNG default keeps the fast form. */
int smc_set_exact_pow(smc_ctx *ctx, int enable);
int smc_set_resampling(smc_ctx *ctx, int scheme);

/* ---- particle movement -------------------------------------------------------------------- */
int smc_upload_particles(smc_ctx *ctx, int set, const double *aos, int64_t n);   /* (n,d) -> SoA */
int smc_download_particles(smc_ctx *ctx, int set, double *aos, int64_t n);
int smc_upload_lk(smc_ctx *ctx, int set, const double *lk, int64_t n);
int smc_download_lk(smc_ctx *ctx, int set, double *lk, int64_t n);
/* r_ac, the ever-accepted flags of the current tempering step (Micmem_SMC_main.py:187,241). */
int smc_download_accept_flags(smc_ctx *ctx, uint8_t *flags, int64_t n);
/* Diagnostics (tools/sweep_tail_census.py): the per-(experiment, particle) record of the last Michaelis-Menten sweep over n
 * particles, info[e * n + p] = RK45 attempts | cancelled by early rejection << 29 | failed << 30.  n_ex * n entries. */
int smc_download_item_info(smc_ctx *ctx, int32_t *info, int64_t n);
/* p_pred, lk = p_filt.copy(), lk1.copy() (Micmem_SMC_main.py:251-252): device-to-device. */
int smc_commit_filt_to_pred(smc_ctx *ctx);
/* Device-RNG prior draw into SMC_SET_PRED (replaces sample_prior, Micmem_settings.py:69-87, in
 * device-RNG mode): Philox4x32-10 keyed by (seed, global particle index, parameter). */
int smc_sample_prior_device(smc_ctx *ctx, uint64_t seed, int64_t global_offset);

/* ---- A2: likelihood sweep ----------------------------------------------------------------- */
/* sim_particle (Micmem_likelihood.py:79-92) on the resident set: lk[i] = log_likelihood_mm_multi(
 * theta_i) (Micmem_likelihood.py:35-77).  n_failed counts particles with a failed RK45 solve (their
 * lk is NaN; the reference raises there).  rk_attempts (optional) receives the device-counted number
 * of RK45 step attempts summed over particles x experiments (flop accounting). */
int smc_loglik(smc_ctx *ctx, int set, int64_t *n_failed, int64_t *rk_attempts);
/* Same for a host batch (the drop-in sim_particle surface): particle (n,3) in, lk[n] out and, when
 * pred != NULL, the model predictions C_l_ as n x n_ex x n_t (P_model, Micmem_likelihood.py:32,74).
 * Any n up to the context capacity. Does not disturb the resident sets. */
int smc_mm_loglik_host(smc_ctx *ctx, const double *particle, int64_t n, double *lk, double *pred, int64_t *n_failed,
                       int64_t *rk_attempts);

/* ---- A3/A4: tempered weights, ESS ---------------------------------------------------------- */
/* max(lk) over this rank's SMC_SET_PRED block (Micmem_SMC_main.py:116). */
int smc_max_lk_local(smc_ctx *ctx, double *max_lk);
/* For k < n_cand: sum_w[k] = sum_i exp((lk_i-max_lk)*gm[k]), sum_w2[k] = sum_i exp(...)^2 over this
 * rank's block (Micmem_SMC_main.py:118,124-134; ess = sum_w^2 / sum_w2 / N).  n_cand <= SMC_MAX_ESS_CAND. */
int smc_ess_partials(smc_ctx *ctx, double max_lk, const double *gm, int n_cand, double *sum_w, double *sum_w2);
/* The same two reductions over ALL ranks: the partial stays on the device, RCCL reduces it
 * in place on the context's stream (allreduce MAX of 1 f64 / SUM of 2 n_cand f64) and ONE read-back follows - no
 * device->host->device->host round trip per collective.  With one rank (no communicator) they equal the calls above. */
int smc_max_lk_global(smc_ctx *ctx, double *max_lk);
int smc_ess_partials_global(smc_ctx *ctx, double max_lk, const double *gm, int n_cand, double *sum_w, double *sum_w2);
/* One call, ONE synchronisation for a batch of the back-off search (Micmem_SMC_main.py:116-134): max(lk) over all ranks
 * (with_max != 0; it stays on the device), then the sums of up to 32 candidate increments in two back-to-back 16-candidate
 * passes that read the maximum from device memory, one all-reduce, one read-back.  sum_w / sum_w2: n_cand values, the same
 * numbers smc_max_lk_global + smc_ess_partials_global return (same kernels, same order of additions).  with_max == 0 reuses
 * the maximum of the previous call (the search going on beyond its first 32 candidates). */
int smc_ess_search_global(smc_ctx *ctx, const double *gm, int n_cand, int with_max, double *max_lk, double *sum_w,
                          double *sum_w2);

/* ---- A5: residual-systematic resampling (Micmem_SMC_main.py:147-184) ------------------------ */
/* Phase 1: with w_i = exp((lk_i-max_lk)*gm)/sum_weight_global, p_is_i = trunc(w_i*N_global) and the
 * residual w_i - p_is_i/N_global: returns this rank's sum of residuals and of p_is. */
int smc_resample_phase1(smc_ctx *ctx, double max_lk, double gm, double sum_weight_global, double *residual_sum_local,
                        int64_t *count_sum_local);
/* Phase 2: residual_prefix = sum of the residual sums of lower ranks; wrand = rand()/N_global (:156).
 * Adds the systematic offspring (:165-174) and builds the inclusive offspring scan.  Returns this
 * rank's total offspring. */
int smc_resample_phase2(smc_ctx *ctx, double max_lk, double gm, double sum_weight_global, double residual_prefix,
                        double wrand, int64_t *offspring_local);
/* Offspring counts p_is (after phase 2), for inspection / parity tests. */
int smc_download_offspring(smc_ctx *ctx, int64_t *p_is, int64_t n);
/* Phase 3 (:178-184): this rank's offspring occupy global output slots [out_base, out_base+offspring)
 * in ancestor order.  Slots owned by this rank are written straight into SMC_SET_FILT; slots owned by
 * other ranks are exchanged with grouped RCCL send/recv (smc_comm_init must have been called when
 * world > 1).  out_base_all / offspring_all: arrays of length world (the allgathered values).
 * Slots >= total offspring keep the content the reference's persistent p_filt would hold
 * (zeros before the first tempering step, the previous p_pred row afterwards). */
int smc_resample_phase3(smc_ctx *ctx, const int64_t *out_base_all, const int64_t *offspring_all, int first_step);
/* The send / receive plan phase 3 executes on rank `rank` of `world` (host arithmetic only: needs neither a context nor a
 * GPU).  Arrays of length world: send_off / send_cnt (one contiguous block per peer in the send staging, in particles),
 * src_lo (first local offspring index of the block for peer q; for q == rank the block that stays), recv_off / recv_cnt
 * (receive staging) and recv_row (first row of SMC_SET_FILT the block of peer q is spread over); own[3] = {src_lo, count,
 * first row} of the offspring that stay on the rank; stale_lo = first row nobody writes (n_local if none).  Any output
 * pointer may be NULL.  Returns 1 if out_base_all is not the exclusive prefix of offspring_all, 2 on bad arguments.
 * Sender and receiver plans must agree - send_cnt of rank s towards r == recv_cnt of rank r from s - and every row of every
 * rank must be written exactly once: tests/test_exchange_plan.py checks both for world sizes 2..8 without a GPU. */
int smc_exchange_plan(int world, int rank, int64_t n_local, const int64_t *out_base_all, const int64_t *offspring_all,
                      int64_t *send_off, int64_t *send_cnt, int64_t *src_lo, int64_t *recv_off, int64_t *recv_cnt,
                      int64_t *recv_row, int64_t *own, int64_t *stale_lo);
/* Micmem_SMC_main.py:147-184 across all ranks in one call: phase 1 -> all-gather of (residual sum, integer copies) ->
 * residual prefix of the lower ranks as a running sum in rank order (:165-167) -> phase 2 -> all-gather of the offspring
 * counts -> phase 3.  Two synchronisations instead of six.  n_offspring = sum of p_is over all particles (== N unless the
 * reference's stale-row case, :178-184, occurs), count_sum = sum of trunc(w*N) (the reference prints N - count_sum as n_tmp). */
int smc_resample_global(smc_ctx *ctx, double max_lk, double gm, double sum_weight_global, double wrand, int first_step,
                        int64_t *n_offspring, int64_t *count_sum);

/* The same resampling ENQUEUED: with one rank no number of it has to visit the host (the residual prefix of the lower ranks is
 * zero, the gather kernel reads the offspring total on the device), so the call returns without a synchronisation and the
 * Metropolis sweeps can be enqueued right behind it.  smc_resample_result - call it after your next synchronisation - returns
 * the two numbers the driver logs (n_offspring, count_sum as above) and fails if the resampler produced more offspring than
 * particles (the reference's IndexError, :180).  With several ranks smc_resample_enqueue is smc_resample_global. */
int smc_resample_enqueue(smc_ctx *ctx, double max_lk, double gm, double sum_weight_global, double wrand, int first_step);
int smc_resample_result(smc_ctx *ctx, int64_t *n_offspring, int64_t *count_sum);

/* Page-locked host memory for results (process-wide, valid until smc_pinned_free - also after smc_destroy): a download into it
 * is one DMA transfer; into pageable memory the runtime stages through its own buffers and copies on the host. */
int smc_pinned_alloc(size_t bytes, void **out);
int smc_pinned_free(void *p);

/* ---- A6: proposal covariance (np.cov(p_filt.T, bias=True), Micmem_SMC_main.py:212) ---------- */
/* sums[d] = sum_i theta_i over this rank's SMC_SET_FILT block. */
int smc_moment_sums_local(smc_ctx *ctx, double *sums);
/* centered[d*d] (row-major, symmetric) = sum_i (theta_i-mean)(theta_i-mean)^T over this rank's block. */
int smc_moment_centered_local(smc_ctx *ctx, const double *mean, double *centered);

/* ---- A7-A9: one random-walk Metropolis iteration, fused with the likelihood ------------------ */
/* Host-RNG (parity) mode.  noise is the (n,d) array np.random.multivariate_normal returned
 * (Micmem_SMC_main.py:220), rr the n uniforms of :235.  Performs :220-241 on SMC_SET_FILT in place:
 * proposal = p_filt + noise*mhstep_ratio; support mask p0; proposal reset where p0 == 0; lk2 =
 * likelihood(proposal); accept r = exp((lk2-lk1)*gamma)*p0 >= rr; select p_filt, lk1; r_ac = max(r_ac, r).
 * Outputs (local): accepted_now = sum(r), accepted_ever = sum(r_ac). */
int smc_mh_step_host_rng(smc_ctx *ctx, double gamma, double mhstep_ratio, const double *noise, const double *rr,
                         int64_t n, int64_t *accepted_now, int64_t *accepted_ever, int64_t *n_failed,
                         int64_t *rk_attempts);
/* Device-RNG mode.  noise_i = z_i @ transform (transform is d x d row-major, i.e. NumPy's
 * sqrt(s)[:,None]*v factor of cov_m), z_i and rr_i from Philox4x32-10 keyed by (seed, global particle
 * index = global_offset+i, stream).  Everything else as above. */
int smc_mh_step_device_rng(smc_ctx *ctx, double gamma, double mhstep_ratio, const double *transform, uint64_t seed,
                           uint64_t stream, int64_t global_offset, int64_t *accepted_now, int64_t *accepted_ever,
                           int64_t *n_failed, int64_t *rk_attempts);
/* One whole Metropolis iteration of the device-RNG mode on the stream (Micmem_SMC_main.py:212-241), every rank calling it
 * together: cov_m = np.cov(p_filt.T, bias=True) * w_cov (:212-215), the factor NumPy's legacy multivariate_normal multiplies
 * standard normals with - (u,s,v) = svd(cov_m), sqrt(s)[:,None]*v - computed on the device (cov_m is symmetric: a Jacobi
 * eigen-decomposition, |lambda| sorted descending, largest component of each row positive), then proposal, support mask,
 * likelihood, accept/select as in smc_mh_step_device_rng, then the accept counts summed over the ranks.  ONE synchronisation.
 * Moments: the first call after anything else wrote SMC_SET_FILT (upload, resampling, smc_mh_step_*) takes np.cov's own
 * two-pass route (column sums -> all-reduce -> sums centred about the global mean -> all-reduce); for the Michaelis-Menten
 * model every later call gets them from the accept kernel of the call before, which accumulates the moments of the
 * particles it selects about the previous mean - no pass over the particles, and moments + accept counts travel in ONE
 * all-reduce of d + d(d+1)/2 + 3 doubles.  w_cov: d x d (Micmem_settings.py:94-97).  accepted_now, accepted_ever and
 * n_failed are totals over ALL ranks, rk_attempts_local is this rank's; cov_m (optional, d x d) receives the global cov_m. */
int smc_mh_iteration_device_rng(smc_ctx *ctx, double gamma, double mhstep_ratio, const double *w_cov, uint64_t seed,
                                uint64_t stream, int64_t global_offset, int64_t *accepted_now, int64_t *accepted_ever,
                                int64_t *n_failed, int64_t *rk_attempts_local, double *cov_m);
/* A BATCH of those iterations with the loop control of Micmem_SMC_main.py:243-249 on the device: n_iter (1 .. 32) iterations are
 * enqueued back to back and the call synchronises ONCE.  After every iteration a one-block kernel takes the reference's
 * decision on the counts summed over all ranks: `break` when r_ac.sum() > thr_stop (= r_th * n_particle, :243-246) - every
 * kernel of the remaining iterations then returns at once - else mhstep_ratio *= 0.5 when r_ac.sum() < thr_halve
 * (= r_threshold_min * n_particle, :247-249); a failed solve also ends the loop.  Pass the two thresholds as the doubles
 * Python computes; counts are compared as exact doubles.  Iteration i draws with Philox stream stream0 + i and otherwise is
 * smc_mh_iteration_device_rng's iteration, bit for bit.  Outputs: n_done iterations ran (1 <= n_done <= n_iter), stopped != 0
 * if the loop ended by its break (or a failure) - otherwise the caller may enqueue the next batch with mhstep_ratio =
 * ratio_next; per iteration i < n_done (arrays of n_iter entries, each optional): accepted_now/ever and n_failed (all ranks),
 * rk_attempts_local (this rank), ratio_used (the mhstep_ratio it drew with), cov_m (n_iter x d x d, row-major).
 * sweep_counters (optional, n_iter x SMC_SWEEP_COUNTER_WORDS): this rank's device counters as iteration i left them, in the
 * order n_failed, rk_attempts (methanation: BDF steps), accepted_now, accepted_ever, newton_iters, factorisations, failed_solves,
 * expected_solves, completed_solves, unsolved_items, wave_split, cancelled_solves, long_items, solved_items - what
 * smc_meth_sweep_counters / smc_meth_sweep_check return after a single sweep (after the batch they describe its last sweep).
 * Michaelis-Menten (carried moments: one all-reduce per iteration) and methanation (np.cov's two passes per iteration: two
 * all-reduces, the second one carrying the counts); every kernel of an iteration tests the stop flag first, the experiment
 * order of the methanation sweeps is formed on the device.  A user model takes smc_mh_iteration_device_rng. */
#define SMC_SWEEP_COUNTER_WORDS 14
int smc_mh_sweeps_device_rng(smc_ctx *ctx, double gamma, double mhstep_ratio, const double *w_cov, uint64_t seed,
                             uint64_t stream0, int n_iter, double thr_stop, double thr_halve, int64_t global_offset,
                             int *n_done, int *stopped, double *ratio_next, int64_t *accepted_now, int64_t *accepted_ever,
                             int64_t *n_failed, int64_t *rk_attempts_local, double *ratio_used, double *cov_m,
                             int64_t *sweep_counters);
/* The first half of that iteration on its own (no model needed): cov_m = np.cov(p_filt.T, bias=True) * w_cov over all ranks
 * (:212-215) and the factor sqrt(s)[:,None]*v of its SVD, both d x d row-major; either output may be NULL. */
int smc_proposal_factor_device(smc_ctx *ctx, const double *w_cov, double *cov_m, double *transform);
/* The d x d factor (row-major) the last smc_mh_iteration_device_rng drew its proposals with. */
int smc_mh_iteration_last_transform(smc_ctx *ctx, double *transform);
/* r_ac = zeros (Micmem_SMC_main.py:187). */
int smc_reset_accept_flags(smc_ctx *ctx);
/* Copies of the last proposals/lk2 for parity tests (valid after an MH step when enabled). */
int smc_set_debug_capture(smc_ctx *ctx, int enable);
int smc_download_debug_proposals(smc_ctx *ctx, double *aos, double *lk2, uint8_t *p0, uint8_t *r, int64_t n);

/* ---- collectives (RCCL over xGMI) ------------------------------------ */
/* unique id = 128 bytes from smc_comm_get_unique_id on rank 0, broadcast by the launcher. */
int smc_comm_get_unique_id(uint8_t id[128]);
int smc_comm_init(smc_ctx *ctx, const uint8_t id[128], int rank, int world);
/* What RCCL itself says about the communicator of this context: ncclCommCount / ncclCommUserRank / ncclCommCuDevice
 * (count = 0, user_rank = device = -1 when there is none: a single rank needs no communicator).  A launcher can check
 * that RCCL saw as many ranks as it started. */
int smc_comm_info(smc_ctx *ctx, int *count, int *user_rank, int *device);
int smc_comm_allreduce_sum_f64(smc_ctx *ctx, double *inout, int n);
int smc_comm_allreduce_max_f64(smc_ctx *ctx, double *inout, int n);
int smc_comm_allreduce_sum_i64(smc_ctx *ctx, int64_t *inout, int n);
int smc_comm_allgather_f64(smc_ctx *ctx, const double *in, int n, double *out /* world*n */);
int smc_comm_allgather_i64(smc_ctx *ctx, const int64_t *in, int n, int64_t *out /* world*n */);
int smc_comm_barrier(smc_ctx *ctx);

/* ---- test and probe hooks (NOT part of the drop-in surface) -------------------------------------
 * Compiled into the library always (tests/ call them through the same C ABI), declared only when the includer asks:
 * a reference-side binding never needs them. */
#ifdef SMC_ENABLE_DEBUG_API
/* One-rank rehearsal of the particle exchange with real RCCL calls: rows [row, row+cnt) of the PRED set go through
 * the send staging, an ncclSend/ncclRecv pair addressed to this rank itself and the receive staging into rows
 * [dst_row, dst_row+cnt) of the FILT set (theta and lk) - step 3 of smc_resample_phase3 for a self-addressed block.
 * Needs a communicator (smc_comm_init with SMC_FORCE_RCCL=1 on one rank). */
/* Probes (tools/sort_probe.py): every Michaelis-Menten likelihood sweep hands its index-ordered items out in the uploaded order
 * (a permutation of 0 .. n-1: position -> particle) with the given in-phase patience; order == NULL switches it off. */
int smc_debug_set_order(smc_ctx *ctx, const int32_t *order, int64_t n, int patience);
int smc_debug_rccl_self_exchange(smc_ctx *ctx, int64_t row, int64_t cnt, int64_t dst_row);
/* Loopback rehearsal of the multi-rank path on ONE device: `peers` are the `world` contexts of one process
 * (peers[rank] == ctx), typically one host thread per rank.  smc_resample_phase3 then only packs (gathers own
 * slots, stages remote ones, synchronises its stream); after a barrier between the threads every rank calls
 * smc_resample_phase3_pull, which copies its incoming slots out of the peers' staging buffers (device-to-
 * device) instead of the RCCL send/recv pairs.  Everything else (kernels, offsets, counts) is the RCCL path. */
int smc_debug_set_local_peers(smc_ctx *ctx, smc_ctx **peers, int rank, int world);
int smc_resample_phase3_pull(smc_ctx *ctx);
/* ... and with the collectives INSIDE the engine (after smc_debug_set_local_peers on every peer, at most 8): the *_global entry
 * points, smc_resample_enqueue, smc_mh_iteration_device_rng and smc_mh_sweeps_device_rng then take their world > 1 branches with
 * every ncclAllReduce / ncclAllGather / send-recv pair replaced by its counterpart among the peer contexts (events between
 * their streams, a host barrier between their threads; sums in rank order) - the stop / no-op logic of a speculative batch and
 * the matched reductions run bit for bit as they would over RCCL.  Every peer thread must make the same calls. */
int smc_debug_peer_collectives(smc_ctx *ctx, int enable);
#endif /* SMC_ENABLE_DEBUG_API */

/* ---- measurement ------------------------------------------------------------------------------ */
/* HIP-event timing of the kernels launched on the context's stream: smc_timing_reset clears the
 * accumulators; smc_timing_get returns, for kernel class `which`, launches and total milliseconds
 * measured with hipEvents recorded on that stream around each launch (enabled by smc_timing_enable). */
#define SMC_T_LOGLIK 0
#define SMC_T_MH 1
#define SMC_T_ESS 2
#define SMC_T_RESAMPLE 3
#define SMC_T_MOMENTS 4
#define SMC_T_MAX 5
#define SMC_T_SOLVE 6 /* the persistent RK45 solve kernel alone (inside LOGLIK / MH sweeps) */
#define SMC_T_COUNT 7
int smc_timing_enable(smc_ctx *ctx, int enable);
int smc_timing_reset(smc_ctx *ctx);
int smc_timing_get(smc_ctx *ctx, int which, int64_t *launches, double *total_ms);
/* Device-counted work of the Michaelis-Menten sweeps since the last smc_timing_reset (always on; bench.py's roofline numerator):
 * out[0] (particle, experiment) solves that ran to t_bound and produced their n_t dense outputs - cancelled (exact early
 *        rejection) and masked (out of support) proposals do not count;
 * out[1] RK45 attempts of all solves, finished or cancelled;
 * out[2] launches of the solve kernel that had work;  out[3] launches of a speculative batch (smc_mh_sweeps_device_rng) that
 *        found the loop already ended and returned at once. */
int smc_work_totals(smc_ctx *ctx, int64_t out[4]);

/* ---- methanation model (configs 4-5): context-free building blocks ---------------------------------------
 * State layout as in the reference: 357 = 7 fields x 51 axial nodes, field-major X[f*51+i],
 * f in {Ca,Cb,Cc,Cd,Ce,T,u} (SMC_methanation/methanation_set_likelihood.py:85-91); params = the
 * 18-tuple p0 of my_model (:164): 10 inlet/geometry scalars then the 8 kinetic parameters.
 * Context-free calls on host buffers (they allocate and free device memory per call), parity-tested through the
 * ABI: residual, rate law and likelihood are pinned by the reference's own functions; smc_meth_dae_host (below)
 * is the DAE integration replacing my_model/IDA (:144-277), parity-unpinned (no IDA in the image). */
const char *smc_meth_last_error(void);
/* res[n][357] = reaction(t, X, dX, params) (methanation_set_likelihood.py:69-139). */
int smc_meth_residual_host(int device, const double *X, const double *dX, const double *params, int64_t n,
                           double *res);
/* out[j] = func_rCH4(T, Ca, Cb, Cc, Cd, kin) (:44-58); in is n x 5 (T,Ca,Cb,Cc,Cd), kin n x 8. */
int smc_meth_rate_host(int device, const double *in, const double *kin, int64_t n, double *out);
/* lk[j] = my_loglike(y_j, data, sigma_j, n_data) (:280-300); y is n x 5 x n_data, data 5 x n_data. */
int smc_meth_loglike_host(int device, const double *y, const double *data, const double *sigma, int64_t n, int n_data,
                          double *lk);

/* The body of my_model (:161-208) for n_solves independent (particle, experiment) pairs: integrate
 * F(t,X,X';p0) = 0 from X(0) = y0 (the driver's guess, SMC_methanation_main.py:47-58), X'(0) = 0 to tf with a
 * variable-order BDF (rtol/atol as Assimulo's IDA defaults 1e-6) and map the outlet node to the five
 * standard-state flows (:204-208).  PARITY UNPINNED against IDA (see csrc/meth_dae.h).  A failed solve
 * (status != 0) returns the reference's sentinel flows -10000 (:244-249).
 * p0_all: n_solves x 18, y0_all: n_solves x 357, flows: n_solves x 5, y_final (optional): n_solves x 357,
 * status: n_solves, stats (optional, FIVE words since ABI 3): sums of steps, rejected steps, Newton failures, Newton iterations,
 * factorisations of the iteration matrix. */
int smc_meth_dae_host(int device, const double *p0_all, const double *y0_all, int64_t n_solves, double tf, double rtol,
                      double atol, double h0, double S, double P_stp, double *flows, double *y_final, int32_t *status,
                      int64_t *stats, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* SMC_HIP_H */
