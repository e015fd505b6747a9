// sweep_args.h -- plain argument blocks shared between the library and the run-time compiled user-model kernel (this file
// is also handed to hiprtc as an in-memory header, so: built-in types only, no #include).
#pragma once

namespace smc {

// What the exact early-rejection bound of a solve kernel needs (mm_kernels.hip: mm_certainly_rejected; meth_smc.hip:
// meth_certainly_rejected; user_model.hip: UserOps::certainly_rejected).  Lives in DEVICE memory (ctx->d_reject) and is
// written by the propose kernel of the sweep, so that the solve kernel carries one pointer instead of nine kernel arguments
// in scalar registers through its attempt loops (144 SGPR spills).
struct RejectArgs {
    const double *lk1;          // likelihood of the current particles (lk1, Micmem_SMC_main.py:231)
    const double *rr;           // host-RNG mode: the uniforms of :235
    const double *pratio;       // prior_mode != MASK: p0_2 / p0_1
    double gamma;
    unsigned long long seed, stream;
    long long global_offset;
    int device_rng, prior_mode;
};

// Arguments of smc_user_solve_kernel (user_model.hip), passed by value.
struct UserSolveArgs {
    const double *theta;        // SoA rows [c * stride + p], c < dim
    long long stride, n;
    const unsigned char *p0;    // MH: support flags (0: masked, not solved); nullptr in a likelihood sweep
    const double *t, *obs, *cond;   // n_ex x n_t, n_ex x n_t, n_ex x n_cond
    int n_ex, n_t, n_cond, dim, est_sigma;
    double sigma_fixed, rtol, atol;
    double *sum_r2;             // [e * n + p]: NaN pending, -1 cancelled, >= 0 finished (early rejection: see mm_kernels.hip)
    int *info;                  // [e * n + p]: attempts | cancelled << 29 | failed << 30
    unsigned long long *queue;
    const RejectArgs *rej;      // nullptr: no early rejection in this sweep
    // models with a cost hint (smc_user_cost, include/smc_hip.h): the predictably long solves of the sweep, handed out first
    const unsigned char *listed;    // [p] != 0: particle p is on one of the two lists (the index-ordered pass skips it)
    const int *stiff_list;          // nullptr: no lists.  Ordinary list from the front, solo list from the back (stiff_cap - 1 down)
    const unsigned *stiff_count;    // [0] ordinary, [1] solo entries
    long long stiff_cap;
    unsigned solo_cap;              // solo entries the grid runs at once (one per wave); the overflow is on the ordinary list
    // ... and, for a heterogeneous Metropolis sweep, the cost order of its in-support proposals (mm_kernels.hip: counting sort)
    const int *order;               // position of the index-ordered pass -> particle (nullptr: identity)
    const unsigned *n_ordered;      // how many positions it has (the out-of-support proposals were published by the scan kernel)
    int patience;                   // solve_sched.h: in-phase waves (0: off)
};

// Arguments of smc_user_cost_scan_kernel (user_model.hip): builds the two lists and the flags of a sweep from the hint.
struct UserScanArgs {
    const double *theta;
    long long stride, n;
    const unsigned char *p0;    // masked proposals are not listed
    unsigned char *listed;
    int *stiff_list;
    unsigned *count, *count_next;   // this sweep's pair of counters; the other pair is cleared for the next sweep
    long long stiff_cap;
    unsigned solo_cap;
    // cost order of the sweep (nullptr: none): a class byte per proposal, 0 = longest ... 123 = shortest, 127 = out of support -
    // and the items of those, which no solve kernel will see, published here (sum 0, no attempts)
    unsigned char *bucket;
    double *done_sums;              // [e * n + p]
    int *done_info;
    int n_ex;
};

}  // namespace smc
