// smc_internal.h -- context layout and launch helpers shared by the translation units of
// libsmc_hip.so.  Not part of the ABI (that is include/smc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/smc_hip.h"
#include "sweep_args.h"   // RejectArgs, UserSolveArgs

int smc_fail(smc_ctx *ctx, const char *msg);   // records the message (smc_last_error) and returns 1

namespace smc {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kMaxEx = 16;           // experiments per data set (waves per sweep block)
constexpr int kMaxNt = 256;          // data times per experiment
constexpr int kScanBlock = 256;      // threads per block in the scan / reduction kernels
constexpr int kScanItems = 4;        // consecutive items per thread in the resampling scans
constexpr int kScanTile = kScanBlock * kScanItems;

struct MMModel {     // passed by value to the sweep kernel
    const double *t;      // n_ex*n_t (device)
    const double *P_obs;  // n_ex*n_t (device)
    const double *S0;     // n_ex (device)
    int n_ex, n_t, est_sigma;
    double sigma_fixed, rtol, atol;
};

struct MethModel {   // methanation model (configs 4-5), passed by value to its kernels
    const double *cond;   // n_data x 10: Ca_in,Cb_in,Cc_in,Cd_in,Ce_in,T_in,T_jacket,u_in,void,dz (the first 10 of p0, :164)
    const double *guess;  // n_data x 357 initial states (SMC_methanation_main.py:47-58)
    const double *obs;    // 5 x n_data observed flows (obs_data)
    int n_data, dim, est_sigma;
    double base[9];       // baseparams + sigma_true: values of the parameters that are not estimated
    int est_pos[SMC_MAX_DIM];  // est_position (methanation_set_conditon.py:34)
    double sigma_fixed, tf, rtol, atol, h0, S, P_stp;
};

struct Prior {       // passed by value
    int kind[SMC_MAX_DIM];
    double a[SMC_MAX_DIM], b[SMC_MAX_DIM];
    int d;
};

struct ParticleSet {  // SoA view: theta[c][i] = theta_base[c*stride + i]
    double *theta;
    double *lk;
    int64_t stride;
};

struct SweepCounters;

// The list of predictably long solves ("stiff list") of a sweep: particles with Vmax > kStiffRatio * Km.  Built by the
// propose kernel (Metropolis sweeps) or by mm_stiff_scan_kernel (likelihood sweeps), handed out by the solve kernel BEFORE
// the index-ordered items so that the serial chains that bound a sweep start at its very beginning.  Two counters used
// alternately: the kernel that builds the list of sweep k clears the counter of sweep k + 1.
struct StiffList {
    int32_t *particles;         // capacity = item_cap: the stiff list grows from the front, the solo list from the back
    unsigned *count;            // [0] stiff entries, [1] solo entries of THIS sweep
    unsigned *count_next;       // the other pair: cleared while this sweep's lists are built
    int64_t cap;                // entries in `particles`
    unsigned solo_cap;          // solo entries this sweep's grid can run at once (one per wave): the rest join the ordinary list
};

struct SweepCounters {  // device-side integer counters (order-independent atomics)
    unsigned long long n_failed, rk_attempts, accepted_now, accepted_ever;
    unsigned long long newton_iters, factorisations, failed_solves;   // K8 only (methanation)
    // K8 sweep bookkeeping: solves the DAE kernel was asked for / finished, live items the likelihood kernel found
    // unsolved (status still poisoned), waves that were incomplete at a dequeue.  A sweep is valid only if
    // completed == expected and the other two are zero (checked on the host after every sweep).
    unsigned long long expected_solves, completed_solves, unsolved_items, wave_split;
    unsigned long long cancelled_solves;   // solves not started because their proposal was already certain to be rejected
    unsigned long long long_items;         // Michaelis-Menten: items that needed more than kLongItemAttempts attempts (mm_kernels.hip)
    unsigned long long solved_items;       // Michaelis-Menten: (particle, experiment) solves that reached t_bound and produced their dense outputs
};

// Loop control of a batch of Metropolis iterations on the device (stage_kernels.hip: mh_control_kernel;
// Micmem_SMC_main.py:243-249).  One block in device memory per context; every kernel of an iteration that belongs to a batch
// reads `stop` first thing and returns at once when it is set.
struct MHControl {
    int stop;                   // the loop has ended (break at :243-246, or a failed solve)
    int n_done;                 // iterations of the batch that ran
    double ratio;               // mhstep_ratio of the next iteration (:190, halved at :247-249)
    double thr_stop, thr_halve; // r_th * n_particle, r_threshold_min * n_particle
};
struct MHLogEntry {             // what the driver's Python kept per iteration (mh_log) and counted (account)
    double ratio;               // mhstep_ratio the iteration drew its proposals with
    double accepted_now, accepted_ever, n_failed;   // totals over all ranks
    unsigned long long rk_attempts, long_items;     // this rank's
    unsigned long long solved_items;                // this rank's: solves of the iteration that ran to the end (not cancelled, not masked)
    SweepCounters snap;                             // this rank's counters as the iteration left them (methanation: K8's work and the
                                                    // completeness bookkeeping the host checks per sweep)
    double cov[SMC_MAX_DIM * SMC_MAX_DIM];          // cov_m of the iteration (:212-215)
};
constexpr int kMHBatchMax = 32;                     // iterations per batch (ad_mhstep_num is 20)
constexpr int kCtlInit = 1, kCtlDecide = 2, kCtlTransform = 4;
struct SweepCounters;
struct MHControlArgs {
    MHControl *ctl;
    MHLogEntry *log;
    int mode;                   // kCtl* flags
    int iteration;              // DECIDE: iteration - 1 has just ended; TRANSFORM: iteration is about to start (within the batch)
    double ratio0, thr_stop, thr_halve;   // INIT
    // DECIDE
    const double *rows;         // one rank: per-block moment rows of the accept kernel (nullptr: vec is already reduced)
    int n_rows, nv;
    int counts_local;           // one rank, no rows (models without carried moments): the counts come straight from `counters`
    double *vec;                // [moments (nv) | accepted_now, accepted_ever, n_failed]
    const SweepCounters *counters;
    // TRANSFORM (mh_transform_body)
    const double *mom, *sums;
    double n_global;
    double *shift_io, *cov_out, *xform_out;
};

struct MHParams {    // passed by value to the fused MH kernel
    double gamma, ratio;
    const double *noise;  // host-RNG mode: SoA d x n (device); nullptr in device-RNG mode
    const double *rr;     // host-RNG mode: n uniforms (device)
    double transform[SMC_MAX_DIM * SMC_MAX_DIM];  // device-RNG mode: z @ transform
    const double *transform_dev;  // if set: the same d x d factor in device memory (fused iteration), read instead
    // fused iteration, Michaelis-Menten path: the accept kernel also accumulates the moments the NEXT iteration's proposal
    // covariance needs, about the shift vector at moment_shift (d doubles, device), into per-block rows of moment_rows
    const double *moment_shift;
    double *moment_rows;
    // ... and the propose kernel clears the sweep counters and the work queue (saves two memset launches per iteration)
    SweepCounters *zero_counters;
    unsigned long long *zero_queue;
    // ... and, with early rejection on, marks every item of the sweep as not finished yet
    double *pending_sums;
    int pending_n_ex;
    uint64_t seed, stream;
    int64_t global_offset;
    int device_rng;
    int prior_mode;       // SMC_PRIOR_MODE_*
    double *pratio;       // prior_mode != MASK: p0_2 / p0_1 per particle (written by propose, read by accept)
    // Michaelis-Menten path: the propose kernel also publishes the early-rejection arguments and builds the stiff list
    RejectArgs *reject_out;     // nullptr: early rejection off in this sweep
    const double *reject_lk1;
    StiffList stiff;            // particles == nullptr: no list
    const MHControl *ctl;       // batch of iterations under device control: stop flag and mhstep_ratio live here (nullptr: mh.ratio)
    uint8_t *cost_bucket;       // cost order of the sweep (mm_kernels.hip: mm_cost_bucket): one byte per proposal, or nullptr
    unsigned *cost_table;       // ... and the counting sort's table, whose histogram rows the propose kernel's blocks add to
    double *done_sums;          // ... whose out-of-support proposals the propose kernel publishes itself ([e * n + p])
    int *done_info;
};


struct EventPair {
    hipEvent_t a, b;
    int which;
};

}  // namespace smc

struct smc_ctx {
    int device = 0;
    int dim = 0;
    int64_t n_local = 0, n_global = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // particle sets (SoA), lk arrays
    smc::ParticleSet set[2]{};
    uint8_t *r_ac = nullptr;

    // model
    int model_kind = 0;   // 0 none, 1 Michaelis-Menten, 2 methanation, 3 user model (hiprtc)
    void *user = nullptr; // UserModel (user_model.hip)
    bool launch_failed = false;   // a launcher without a status (module launch) failed: the message is in err
    smc::MethModel meth{};
    double *d_mcond = nullptr, *d_mguess = nullptr, *d_mobs = nullptr, *d_mflows = nullptr, *d_mlk2 = nullptr;
    int *d_mstatus = nullptr;
    int64_t *d_mwork = nullptr;      // work list of an MH sweep: the live proposals, compacted
    double *d_mstat = nullptr;       // 2 x n_data: per experiment, sum of log1p(misfit) and count over the last sweep's solves
    int *d_morder = nullptr;         // n_data: the order in which the next sweep solves the experiments (most misfit first)
    bool have_model = false, have_prior = false;
    smc::MMModel mm{};
    double *d_t = nullptr, *d_P = nullptr, *d_S0 = nullptr;
    smc::Prior prior{};

    // scratch
    smc::SweepCounters *d_counters = nullptr;
    smc::SweepCounters *h_counters = nullptr;   // pinned
    double *d_noise = nullptr, *d_rr = nullptr;  // host-RNG staging: (d+1) x n_local
    double *d_stage = nullptr;                   // AoS staging for upload/download: n_local*d
    double *d_partials = nullptr;                // reduction partials
    int64_t partials_cap = 0;
    double *d_small = nullptr, *h_small = nullptr;  // small results (device / pinned host), 4096 doubles
    double *d_fused = nullptr, *h_fused = nullptr;  // the fused Metropolis iteration's own 256 doubles (carried moments, cov_m, factor)
    double *d_ess = nullptr, *h_ess = nullptr;      // the fused ESS search's own 128 doubles: max(lk) and the candidates' sums
    bool ess_max_valid = false;                     // d_ess[0] holds max over the PRED set's lk as it is now
    smc::MHControl *d_mhctl = nullptr;              // device-side loop control of a batch of Metropolis iterations
    smc::MHLogEntry *d_mhlog = nullptr, *h_mhlog = nullptr;   // its per-iteration log (kMHBatchMax + 1 entries; host copy pinned)
    smc::MHControl *h_mhctl = nullptr;              // pinned copy of the control block, read back with the log
    // resampling
    int32_t *d_oscan = nullptr;      // inclusive offspring scan (n_local)
    double *d_blk_r = nullptr;       // per tile: residual sums, then exclusive prefix
    int64_t *d_blk_c = nullptr;      // per tile: integer sums, then exclusive prefix
    int64_t *d_blk_o = nullptr;      // per tile: offspring totals of phase 2, then exclusive prefix
    int64_t *d_rs = nullptr, *h_rs = nullptr;   // (sum of trunc(w N), offspring total) of an enqueued resampling; host copy pinned
    int rs_pending = 0;              // smc_resample_enqueue has left its two numbers for smc_resample_result (1: on the device, 2: in h_rs)
    int64_t n_tiles = 0;
    double *d_sendbuf = nullptr;     // (d+1) x capacity staging for remote offspring
    int64_t sendbuf_cap = 0;
    double *d_recvbuf = nullptr;     // (d+1) x capacity staging for offspring arriving from peers
    int64_t recvbuf_cap = 0;
    // sweep scratch: per (experiment, particle) sums of squared residuals and solver info, support
    // flags of the proposals, the global work counter of the persistent solve kernel
    double *d_sum_r2 = nullptr;
    int *d_info = nullptr;
    int64_t item_cap = 0;            // particles the two arrays above can hold (x kMaxEx experiments)
    uint8_t *d_p0 = nullptr;
    double *d_pratio = nullptr;      // prior density ratio of the proposals (allocated on first use)
    int prior_mode = 0;
    int resampling = 0;              // SMC_RESAMPLE_*
    int early_reject = 1;            // stop a solve whose proposal is certainly rejected (mm_kernels.hip: mm_certainly_rejected)
    double *d_mn_thr = nullptr;      // multinomial resampling: n_global + 1 thresholds
    double *d_mn_blk = nullptr;      // ... and their per-tile sums
    unsigned long long *d_queue = nullptr;
    smc::RejectArgs *d_reject = nullptr;   // early-rejection arguments of the running sweep (written by its propose kernel)
    int32_t *d_stiff_list = nullptr;       // stiff list (item_cap entries) and its two alternating counters
    unsigned *d_stiff_count = nullptr;
    int stiff_parity = 0;
    int32_t *d_order = nullptr;            // cost order of a sweep (n_local entries): position -> particle
    uint8_t *d_bucket = nullptr;           // ... its cost class per proposal, and the counting sort's (blocks + 1) x buckets table
    unsigned *d_order_hist = nullptr;
    void *d_sorted = nullptr;              // ... and the proposals in cost order (n_local 32-byte records, mm_kernels.hip: SortedProposal)
    int cost_order = 1;                    // hand a Metropolis sweep's index-ordered items out by cost class (smc_set_cost_order)
    int order_debug = 0, order_debug_patience = 0;   // smc_debug_set_order: every MM sweep uses the uploaded order (probes)
    int fast_tail = 1;                     // hand-written lone-chain loop in the Michaelis-Menten solve kernel (smc_set_fast_tail)
    int stiff_first = 1;                   // hand the predictably long solves out first (smc_set_stiff_first)
    int in_phase = 1;                      // let homogeneous Metropolis sweeps run their waves in phase (solve_sched.h: patience)
    int64_t last_sweep_items = 0, last_sweep_long_items = 0, pending_sweep_items = 0;   // of the last finished MM Metropolis sweep
    int exact_pow = 0;                     // parity mode: correctly rounded pow(x, -0.2) in the step controller (smc_set_exact_pow)
    bool solve_lds_raised = false;         // hipFuncAttributeMaxDynamicSharedMemorySize raised on THIS device
    int cu_count = 0, solve_blocks_per_cu = 0, solve_blocks_per_cu_fast = 0;   // persistent blocks per CU of the two kinds of mm_solve_kernel
    // debug capture of the last MH iteration (lk2, accept flags; proposals live in SMC_SET_PRED)
    int debug_capture = 0;
    double *dbg_lk2 = nullptr;
    uint8_t *dbg_r = nullptr;
    // host batch sweeps (drop-in sim_particle)
    double *d_hb_theta = nullptr, *d_hb_lk = nullptr, *d_hb_pred = nullptr;
    int64_t hb_cap = 0, hb_pred_cap = 0;
    // fused Metropolis iterations: d_fused holds the moments of the FILT set that the last accept kernel accumulated (valid
    // until anything else writes the FILT set)
    bool moments_valid = false;
    int moment_rows_n = 0;           // per-block partial rows the last accept kernel wrote to d_partials

    // comm
    void *nccl_comm = nullptr;
    int rank = 0, world = 1;
    // loopback rehearsal (smc_debug_set_local_peers): peer contexts on the same device, last exchange plan
    std::vector<smc_ctx *> peers;
    std::vector<int64_t> plan_send_off, plan_send_cnt, plan_base, plan_cnt;
    // ... with the collectives INSIDE the engine (smc_debug_peer_collectives): every ncclAllReduce / ncclAllGather / send-recv of
    // the *_global entry points replaced by its counterpart among the peer contexts - events between their streams, a host
    // barrier between their threads - so that the world > 1 branches of those entry points run on a one-GPU box
    bool peer_collectives = false;
    double *d_peer_contrib = nullptr;           // 2 x kPeerWords: this rank's contribution to a collective, two in turn
    hipEvent_t peer_ev[2]{};                    // ... recorded once it is written
    hipEvent_t peer_pack_ev = nullptr, peer_pull_ev = nullptr;   // exchange: send staging packed / peers' blocks pulled
    bool peer_pull_recorded = false;
    int peer_parity = 0;
    void *peer_barrier = nullptr;               // host barrier of the peer threads (rank 0's object is the one in use)

    // timing
    int timing = 0;
    std::vector<smc::EventPair> ev_used;
    std::vector<smc::EventPair> ev_free;
    int64_t t_launches[SMC_T_COUNT]{};
    double t_ms[SMC_T_COUNT]{};
    // work totals since smc_timing_reset (smc_work_totals): Michaelis-Menten solves that produced their outputs, RK45 attempts, solve
    // launches that had work, solve launches of a speculative batch that found the loop ended
    int64_t w_solved_items = 0, w_rk_attempts = 0, w_solve_launches = 0, w_noop_launches = 0;
};

namespace smc {

// kernel launchers implemented in mm_kernels.hip
void launch_mm_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk, double *pred);
void launch_mm_mh(smc_ctx *ctx, int64_t n, const MHParams &mh);   // mh.moment_rows set: returns the row count in ctx->moment_rows_n
int query_solve_blocks_per_cu(bool fast);
// in-phase patience of homogeneous and of cost-ordered sweeps (solve_sched.h; profiles/r03_ab_patience.log, r03_ab_cost_order.log)
constexpr int kInPhasePatience = 12;
size_t cost_table_bytes();                  // mm_kernels.hip: size of ctx->d_order_hist
const unsigned *cost_n_ordered(const smc_ctx *ctx);   // ... and where the solve kernels find the number of ordered positions
const unsigned *launch_cost_sort_order(smc_ctx *ctx, int64_t n);   // mm_kernels.hip: counting sort of ctx->d_bucket into ctx->d_order
// implemented in meth_smc.hip
void launch_meth_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk);
void launch_meth_mh(smc_ctx *ctx, int64_t n, const MHParams &mh);   // mh.ctl set: every kernel of the sweep is a no-op once the loop has ended
// any dimension: proposal + support mask into the PRED set / d_p0; accept-select from a per-particle lk2 array
void launch_generic_propose(smc_ctx *ctx, int64_t n, const MHParams &mh);
void launch_generic_accept(smc_ctx *ctx, int64_t n, const MHParams &mh, const double *lk2);
// user model (user_model.hip)
void launch_user_loglik(smc_ctx *ctx, const double *theta, int64_t stride, int64_t n, double *lk);
void launch_user_mh(smc_ctx *ctx, int64_t n, const MHParams &mh);
void user_model_release(smc_ctx *ctx);

struct ScopedTimer {
    smc_ctx *c;
    EventPair ep{};
    bool on;
    ScopedTimer(smc_ctx *ctx, int which);
    ~ScopedTimer();
};

}  // namespace smc
