// smc_api.hip -- host side of the C ABI declared in include/smc_hip.h: context, device memory,
// stream, stage orchestration, RCCL collectives and HIP-event timing.  No torch, no Python types.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>

#include "exchange_plan.h"
#include "smc_internal.h"
#include "stage_kernels.h"

using namespace smc;

static thread_local std::string g_err;

#define HIPC(ctx, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                     __LINE__);                                                                           \
            if (ctx)                                                                                      \
                (ctx)->err = buf_;                                                                        \
            else                                                                                          \
                g_err = buf_;                                                                             \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

#define NCCLC(ctx, call)                                                                                  \
    do {                                                                                                  \
        ncclResult_t e_ = (call);                                                                         \
        if (e_ != ncclSuccess) {                                                                          \
            char buf_[512];                                                                               \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                           \
            if (ctx)                                                                                      \
                (ctx)->err = buf_;                                                                        \
            else                                                                                          \
                g_err = buf_;                                                                             \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

static int fail(smc_ctx *ctx, const char *msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_err = msg;
    return 1;
}
int smc_fail(smc_ctx *ctx, const char *msg) { return fail(ctx, msg); }

// ---- timing ---------------------------------------------------------------------------------
namespace smc {
ScopedTimer::ScopedTimer(smc_ctx *ctx, int which) : c(ctx), on(ctx->timing != 0) {
    if (!on) return;
    if (!c->ev_free.empty()) {
        ep = c->ev_free.back();
        c->ev_free.pop_back();
    } else {
        (void)hipEventCreate(&ep.a);
        (void)hipEventCreate(&ep.b);
    }
    ep.which = which;
    (void)hipEventRecord(ep.a, c->stream);
}
ScopedTimer::~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(ep.b, c->stream);
    c->ev_used.push_back(ep);
}
}  // namespace smc

static void timing_collect(smc_ctx *c) {
    for (auto &ep : c->ev_used) {
        (void)hipEventSynchronize(ep.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            c->t_ms[ep.which] += ms;
            c->t_launches[ep.which] += 1;
        }
        c->ev_free.push_back(ep);
    }
    c->ev_used.clear();
}

static int ensure_item_capacity(smc_ctx *c, int64_t n);

// ---- collectives among local peers (debug API; rehearsal of the RCCL paths on one device) ------------------------------------
// Contexts of one process on one device, one host thread each (tests/_thread_comm.py).  A collective = every rank copies its
// words into its own contribution slot and records an event; the threads meet at a host barrier (all events recorded); every
// rank makes its stream wait for the peers' events and runs ONE small kernel that reads all contributions in rank order.  Two
// slots used alternately: a rank can only write slot s again after its own next collective, which needed every peer's
// contribution to it, which a peer enqueues after its reads of slot s - so no second barrier.
namespace {
constexpr int kPeerWords = 2048, kPeerMax = 8;
struct PeerBarrier {
    std::mutex m;
    std::condition_variable cv;
    int count = 0;
    unsigned long gen = 0;
};
struct PeerPtrs { const double *p[kPeerMax]; };
__global__ void peer_reduce_kernel(PeerPtrs src, int world, int n, double *out, int is_max) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double v = src.p[0][i];
        for (int q = 1; q < world; ++q) v = is_max ? fmax(v, src.p[q][i]) : v + src.p[q][i];   // rank order: the same on every rank
        out[i] = v;
    }
}
__global__ void peer_gather_kernel(PeerPtrs src, int world, int n, double *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * world; i += gridDim.x * blockDim.x) out[i] = src.p[i / n][i % n];
}
}  // namespace
static int peer_host_barrier(smc_ctx *c) {
    PeerBarrier &b = *static_cast<PeerBarrier *>(c->peers[0]->peer_barrier);
    std::unique_lock<std::mutex> lk(b.m);
    const unsigned long g = b.gen;
    if (++b.count == c->world) {
        b.count = 0;
        ++b.gen;
        b.cv.notify_all();
        return 0;
    }
    if (!b.cv.wait_for(lk, std::chrono::seconds(120), [&] { return b.gen != g; }))
        return fail(c, "local-peer collective: a peer did not arrive within 120 s (the ranks disagree on a collective)");
    return 0;
}
// in place over `n` doubles at buf (device): SUM or MAX over the peers
static int peer_allreduce(smc_ctx *c, double *buf, size_t n, bool is_max, bool gather, double *gather_out) {
    if (n > (size_t)kPeerWords || (gather && n * c->world > (size_t)kPeerWords)) return fail(c, "local-peer collective: payload too large");
    const int s = c->peer_parity;
    c->peer_parity ^= 1;
    double *slot = c->d_peer_contrib + (size_t)s * kPeerWords;
    HIPC(c, hipMemcpyAsync(slot, buf, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipEventRecord(c->peer_ev[s], c->stream));
    if (peer_host_barrier(c)) return 1;
    PeerPtrs src{};
    for (int q = 0; q < c->world; ++q) {
        src.p[q] = c->peers[q]->d_peer_contrib + (size_t)s * kPeerWords;
        if (q != c->rank) HIPC(c, hipStreamWaitEvent(c->stream, c->peers[q]->peer_ev[s], 0));
    }
    if (gather)
        hipLaunchKernelGGL(peer_gather_kernel, dim3(1), dim3(256), 0, c->stream, src, c->world, (int)n, gather_out);
    else
        hipLaunchKernelGGL(peer_reduce_kernel, dim3(1), dim3(256), 0, c->stream, src, c->world, (int)n, buf, is_max ? 1 : 0);
    HIPC(c, hipGetLastError());
    return 0;
}
static bool peers_on(const smc_ctx *c) { return c->peer_collectives && c->world > 1 && !c->peers.empty(); }
// does a *_global entry point have ranks to reduce over, and the means to?  (RCCL, or the local peers of a rehearsal)
static bool can_reduce(const smc_ctx *c) { return c->nccl_comm != nullptr || peers_on(c); }
static int peer_exchange(smc_ctx *c);

extern "C" {

int smc_abi_version(void) { return SMC_ABI_VERSION; }

const char *smc_last_error(const smc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int smc_create(smc_ctx **out, int device, int64_t n_local, int64_t n_global, int dim) {
    if (!out) return fail(nullptr, "smc_create: out is NULL");
    *out = nullptr;
    if (n_local <= 0 || n_global < n_local) return fail(nullptr, "smc_create: need 0 < n_local <= n_global");
    if (n_global >= (int64_t)1 << 31) return fail(nullptr, "smc_create: n_global must be < 2^31");
    if (dim < 1 || dim > SMC_MAX_DIM) return fail(nullptr, "smc_create: dim out of range");
    int ndev = 0;
    HIPC((smc_ctx *)nullptr, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(nullptr, "smc_create: no such HIP device");
    HIPC((smc_ctx *)nullptr, hipSetDevice(device));
    smc_ctx *c = new (std::nothrow) smc_ctx();
    if (!c) return fail(nullptr, "smc_create: out of host memory");
    c->device = device;
    c->dim = dim;
    c->n_local = n_local;
    c->n_global = n_global;
#define CK(call)                                   \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) {                    \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_); \
            smc_destroy(c);                        \
            return 1;                              \
        }                                          \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t nb = (size_t)n_local * sizeof(double);
    for (int s = 0; s < 2; ++s) {
        CK(hipMalloc(&c->set[s].theta, nb * dim));
        CK(hipMalloc(&c->set[s].lk, nb));
        CK(hipMemsetAsync(c->set[s].theta, 0, nb * dim, c->stream));
        CK(hipMemsetAsync(c->set[s].lk, 0, nb, c->stream));
        c->set[s].stride = n_local;
    }
    CK(hipMalloc(&c->r_ac, (size_t)n_local));
    CK(hipMemsetAsync(c->r_ac, 0, (size_t)n_local, c->stream));
    CK(hipMalloc(&c->d_counters, sizeof(SweepCounters)));
    CK(hipMalloc(&c->d_queue, 2 * sizeof(unsigned long long)));
    CK(hipMalloc(&c->d_reject, sizeof(RejectArgs)));
    CK(hipMalloc(&c->d_stiff_count, 4 * sizeof(unsigned)));
    CK(hipMemsetAsync(c->d_stiff_count, 0, 4 * sizeof(unsigned), c->stream));
    CK(hipMalloc(&c->d_p0, (size_t)n_local));
    CK(hipMalloc(&c->d_order, (size_t)n_local * sizeof(int32_t)));          // cost order of a Metropolis sweep (mm_kernels.hip)
    CK(hipMalloc(&c->d_bucket, (size_t)n_local));
    CK(hipMalloc(&c->d_order_hist, cost_table_bytes()));
    CK(hipMemsetAsync(c->d_order_hist, 0, cost_table_bytes(), c->stream));   // the histogram rows are zero between sweeps
    CK(hipMalloc(&c->d_sorted, (size_t)n_local * 32));
    {
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, device));
        c->cu_count = prop.multiProcessorCount;
        c->solve_blocks_per_cu = query_solve_blocks_per_cu(false);
        c->solve_blocks_per_cu_fast = query_solve_blocks_per_cu(true);
    }
    CK(hipHostMalloc(&c->h_counters, sizeof(SweepCounters)));
    CK(hipMalloc(&c->d_noise, nb * (dim + 1)));
    c->d_rr = c->d_noise + (size_t)n_local * dim;
    CK(hipMalloc(&c->d_stage, nb * dim));
    c->partials_cap = 2048 * 64;
    CK(hipMalloc(&c->d_partials, (size_t)c->partials_cap * sizeof(double)));
    CK(hipMalloc(&c->d_small, 4096 * sizeof(double)));
    CK(hipMalloc(&c->d_fused, 256 * sizeof(double)));
    CK(hipMemsetAsync(c->d_fused, 0, 256 * sizeof(double), c->stream));
    CK(hipHostMalloc(&c->h_fused, 256 * sizeof(double)));
    CK(hipMalloc(&c->d_ess, 128 * sizeof(double)));          // the fused ESS search's own words (max(lk) + 2 x 32 sums)
    CK(hipMemsetAsync(c->d_ess, 0, 128 * sizeof(double), c->stream));
    CK(hipHostMalloc(&c->h_ess, 128 * sizeof(double)));
    CK(hipMalloc(&c->d_mhctl, sizeof(MHControl)));            // device-side loop control of a batch of Metropolis iterations
    CK(hipMemsetAsync(c->d_mhctl, 0, sizeof(MHControl), c->stream));
    CK(hipMalloc(&c->d_mhlog, (kMHBatchMax + 1) * sizeof(MHLogEntry)));
    CK(hipMemsetAsync(c->d_mhlog, 0, (kMHBatchMax + 1) * sizeof(MHLogEntry), c->stream));
    CK(hipHostMalloc(&c->h_mhlog, (kMHBatchMax + 1) * sizeof(MHLogEntry)));
    CK(hipHostMalloc(&c->h_mhctl, sizeof(MHControl)));
    CK(hipHostMalloc(&c->h_small, 4096 * sizeof(double)));
    c->n_tiles = (n_local + kScanTile - 1) / kScanTile;
    CK(hipMalloc(&c->d_oscan, (size_t)n_local * sizeof(int32_t)));
    CK(hipMalloc(&c->d_blk_r, (size_t)(c->n_tiles + 1) * sizeof(double)));
    CK(hipMalloc(&c->d_blk_c, (size_t)(c->n_tiles + 1) * sizeof(int64_t)));
    CK(hipMalloc(&c->d_blk_o, (size_t)(c->n_tiles + 1) * sizeof(int64_t)));
    CK(hipMalloc(&c->d_rs, 2 * sizeof(int64_t)));
    CK(hipHostMalloc(&c->h_rs, 2 * sizeof(int64_t)));
    CK(hipStreamSynchronize(c->stream));
#undef CK
    *out = c;
    return 0;
}

void smc_destroy(smc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl_comm) ncclCommDestroy((ncclComm_t)c->nccl_comm);
    for (auto &ep : c->ev_used) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (auto &ep : c->ev_free) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (int s = 0; s < 2; ++s) {
        (void)hipFree(c->set[s].theta);
        (void)hipFree(c->set[s].lk);
    }
    (void)hipFree(c->r_ac);
    (void)hipFree(c->d_t);
    (void)hipFree(c->d_P);
    (void)hipFree(c->d_S0);
    (void)hipFree(c->d_counters);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    (void)hipFree(c->d_noise);
    (void)hipFree(c->d_stage);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_small);
    (void)hipFree(c->d_fused);
    if (c->h_fused) (void)hipHostFree(c->h_fused);
    (void)hipFree(c->d_ess);
    if (c->h_ess) (void)hipHostFree(c->h_ess);
    (void)hipFree(c->d_mhctl);
    (void)hipFree(c->d_mhlog);
    if (c->h_mhlog) (void)hipHostFree(c->h_mhlog);
    if (c->h_mhctl) (void)hipHostFree(c->h_mhctl);
    if (c->h_small) (void)hipHostFree(c->h_small);
    (void)hipFree(c->d_oscan);
    (void)hipFree(c->d_blk_r);
    (void)hipFree(c->d_blk_c);
    (void)hipFree(c->d_blk_o);
    (void)hipFree(c->d_rs);
    if (c->h_rs) (void)hipHostFree(c->h_rs);
    (void)hipFree(c->d_sendbuf);
    (void)hipFree(c->d_recvbuf);
    (void)hipFree(c->dbg_lk2);
    (void)hipFree(c->dbg_r);
    (void)hipFree(c->d_sum_r2);
    (void)hipFree(c->d_info);
    (void)hipFree(c->d_p0);
    (void)hipFree(c->d_pratio);
    user_model_release(c);
    (void)hipFree(c->d_mn_thr);
    (void)hipFree(c->d_mn_blk);
    (void)hipFree(c->d_queue);
    (void)hipFree(c->d_reject);
    (void)hipFree(c->d_stiff_count);
    (void)hipFree(c->d_order);
    (void)hipFree(c->d_bucket);
    (void)hipFree(c->d_order_hist);
    (void)hipFree(c->d_sorted);
    (void)hipFree(c->d_stiff_list);
    (void)hipFree(c->d_mcond);
    (void)hipFree(c->d_mguess);
    (void)hipFree(c->d_mobs);
    (void)hipFree(c->d_mflows);
    (void)hipFree(c->d_mlk2);
    (void)hipFree(c->d_mstatus);
    (void)hipFree(c->d_mwork);
    (void)hipFree(c->d_mstat);
    (void)hipFree(c->d_morder);
    (void)hipFree(c->d_hb_theta);
    (void)hipFree(c->d_hb_lk);
    (void)hipFree(c->d_hb_pred);
    (void)hipFree(c->d_peer_contrib);
    for (int s = 0; s < 2; ++s)
        if (c->peer_ev[s]) (void)hipEventDestroy(c->peer_ev[s]);
    if (c->peer_pack_ev) (void)hipEventDestroy(c->peer_pack_ev);
    if (c->peer_pull_ev) (void)hipEventDestroy(c->peer_pull_ev);
    delete static_cast<PeerBarrier *>(c->peer_barrier);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int smc_synchronize(smc_ctx *c) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smc_device_info(smc_ctx *c, char *name, int name_len, char *arch, int arch_len, int *cu_count) {
    if (!c) return fail(nullptr, "NULL context");
    hipDeviceProp_t p;
    HIPC(c, hipGetDeviceProperties(&p, c->device));
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name);
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    return 0;
}

// ---- model + prior ---------------------------------------------------------------------------
int smc_set_model_mm(smc_ctx *c, const double *t, const double *P_obs, const double *S0, int n_ex, int n_t,
                     int est_sigma, double sigma_fixed, double rtol, double atol) {
    if (!c) return fail(nullptr, "NULL context");
    if (c->dim != 3) return fail(c, "Michaelis-Menten model needs dim == 3 (Vmax, Km, sigma)");
    if (n_ex < 1 || n_ex > kMaxEx || n_t < 1 || n_t > kMaxNt) return fail(c, "n_ex must be 1..16 and n_t 1..256");
    for (int e = 0; e < n_ex; ++e)
        for (int i = 1; i < n_t; ++i)
            if (!(t[e * n_t + i] > t[e * n_t + i - 1])) return fail(c, "t must be strictly increasing (ivp.py:606-609)");
    for (int i = 0; i < n_ex * n_t; ++i)   // the dense-output division relies on it (mm_rk45.h, lean_x)
        if (!(t[i] == 0.0 || (fabs(t[i]) >= 0x1p-400 && fabs(t[i]) <= 0x1p400)))
            return fail(c, "data times must be 0 or between 2^-400 and 2^400 in magnitude");
    HIPC(c, hipSetDevice(c->device));
    (void)hipFree(c->d_t);
    (void)hipFree(c->d_P);
    (void)hipFree(c->d_S0);
    c->d_t = c->d_P = c->d_S0 = nullptr;
    const size_t nb = (size_t)n_ex * n_t * sizeof(double);
    HIPC(c, hipMalloc(&c->d_t, nb));
    HIPC(c, hipMalloc(&c->d_P, nb));
    HIPC(c, hipMalloc(&c->d_S0, n_ex * sizeof(double)));
    HIPC(c, hipMemcpyAsync(c->d_t, t, nb, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_P, P_obs, nb, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_S0, S0, n_ex * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->mm.t = c->d_t;
    c->mm.P_obs = c->d_P;
    c->mm.S0 = c->d_S0;
    c->mm.n_ex = n_ex;
    c->mm.n_t = n_t;
    c->mm.est_sigma = est_sigma;
    c->mm.sigma_fixed = sigma_fixed;
    c->mm.rtol = rtol;
    c->mm.atol = atol;
    c->item_cap = 0;  // per-(experiment, particle) scratch is sized by n_ex
    c->model_kind = 1;
    c->have_model = true;
    return ensure_item_capacity(c, c->n_local);
}

int smc_set_model_methanation(smc_ctx *c, const double *cond, const double *guess, const double *obs, int n_data,
                              const double *base_params, const int *est_position, int est_sigma, double sigma_fixed,
                              double tf, double rtol, double atol) {
    if (!c) return fail(nullptr, "NULL context");
    if (n_data < 1 || n_data > 128) return fail(c, "n_data must be 1..128");
    for (int q = 0; q < c->dim; ++q)
        if (est_position[q] < 0 || est_position[q] > 8) return fail(c, "est_position entries must be 0..8");
    HIPC(c, hipSetDevice(c->device));
    (void)hipFree(c->d_mcond); (void)hipFree(c->d_mguess); (void)hipFree(c->d_mobs); (void)hipFree(c->d_mflows);
    (void)hipFree(c->d_mlk2); (void)hipFree(c->d_mstatus); (void)hipFree(c->d_mwork);
    c->d_mwork = nullptr;
    c->d_mcond = c->d_mguess = c->d_mobs = c->d_mflows = c->d_mlk2 = nullptr;
    c->d_mstatus = nullptr;
    HIPC(c, hipMalloc(&c->d_mcond, (size_t)n_data * 10 * 8));
    HIPC(c, hipMalloc(&c->d_mguess, (size_t)n_data * 357 * 8));
    HIPC(c, hipMalloc(&c->d_mobs, (size_t)n_data * 5 * 8));
    HIPC(c, hipMalloc(&c->d_mflows, (size_t)c->n_local * n_data * 5 * 8));
    HIPC(c, hipMalloc(&c->d_mlk2, (size_t)c->n_local * 8));
    HIPC(c, hipMalloc(&c->d_mstatus, (size_t)c->n_local * n_data * sizeof(int)));
    HIPC(c, hipMalloc(&c->d_mwork, (size_t)c->n_local * n_data * sizeof(int64_t)));
    (void)hipFree(c->d_mstat); (void)hipFree(c->d_morder);
    c->d_mstat = nullptr; c->d_morder = nullptr;
    HIPC(c, hipMalloc(&c->d_mstat, 2 * (size_t)n_data * sizeof(double)));
    HIPC(c, hipMemsetAsync(c->d_mstat, 0, 2 * (size_t)n_data * sizeof(double), c->stream));   // no statistics yet: index order
    HIPC(c, hipMalloc(&c->d_morder, (size_t)n_data * sizeof(int)));
    HIPC(c, hipMemcpyAsync(c->d_mcond, cond, (size_t)n_data * 10 * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_mguess, guess, (size_t)n_data * 357 * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_mobs, obs, (size_t)n_data * 5 * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemsetAsync(c->d_mflows, 0, (size_t)c->n_local * n_data * 5 * 8, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    smc::MethModel &m = c->meth;
    m.cond = c->d_mcond; m.guess = c->d_mguess; m.obs = c->d_mobs;
    m.n_data = n_data; m.dim = c->dim; m.est_sigma = est_sigma;
    for (int q = 0; q < 9; ++q) m.base[q] = base_params[q];
    for (int q = 0; q < SMC_MAX_DIM; ++q) m.est_pos[q] = (q < c->dim) ? est_position[q] : -1;
    m.sigma_fixed = sigma_fixed; m.tf = tf; m.rtol = rtol; m.atol = atol; m.h0 = 1e-5;
    m.S = 3.141592653589793 * (0.01 / 2) * (0.01 / 2);   // methanation_set_conditon.py:80-81
    m.P_stp = 1.013 * 100000;                            // :89
    c->model_kind = 2;
    c->have_model = true;
    return 0;
}

int smc_set_prior(smc_ctx *c, const int *kind, const double *a, const double *b, int dim) {
    if (!c) return fail(nullptr, "NULL context");
    if (dim != c->dim) return fail(c, "smc_set_prior: dim mismatch");
    for (int i = 0; i < dim; ++i) {
        if (kind[i] != SMC_PRIOR_UNIFORM && kind[i] != SMC_PRIOR_NORMAL && kind[i] != SMC_PRIOR_FLAT)
            return fail(c, "Unknown prior kind");
        c->prior.kind[i] = kind[i];
        c->prior.a[i] = a[i];
        c->prior.b[i] = b[i];
    }
    c->prior.d = dim;
    c->have_prior = true;
    return 0;
}

int smc_set_prior_mode(smc_ctx *c, int mode) {
    if (!c) return fail(nullptr, "NULL context");
    if (mode != SMC_PRIOR_MODE_MASK && mode != SMC_PRIOR_MODE_RATIO_MASK && mode != SMC_PRIOR_MODE_RATIO)
        return fail(c, "Unknown prior mode");
    HIPC(c, hipSetDevice(c->device));
    if (mode != SMC_PRIOR_MODE_MASK && !c->d_pratio) HIPC(c, hipMalloc(&c->d_pratio, (size_t)c->n_local * sizeof(double)));
    c->prior_mode = mode;
    return 0;
}

int smc_meth_sweep_counters(smc_ctx *c, int64_t out[4]) {
    if (!c) return fail(nullptr, "NULL context");
    out[0] = (int64_t)c->h_counters->rk_attempts;
    out[1] = (int64_t)c->h_counters->newton_iters;
    out[2] = (int64_t)c->h_counters->factorisations;
    out[3] = (int64_t)c->h_counters->failed_solves;
    return 0;
}

int smc_meth_sweep_check(smc_ctx *c, int64_t out[5]) {
    if (!c) return fail(nullptr, "NULL context");
    out[0] = (int64_t)c->h_counters->expected_solves;
    out[1] = (int64_t)c->h_counters->completed_solves;
    out[2] = (int64_t)c->h_counters->unsolved_items;
    out[3] = (int64_t)c->h_counters->wave_split;
    out[4] = (int64_t)c->h_counters->cancelled_solves;
    return 0;
}

int smc_meth_download_solves(smc_ctx *c, double *flows, int32_t *status, int64_t n) {
    if (!c) return fail(nullptr, "NULL context");
    if (c->model_kind != 2) return fail(c, "smc_set_model_methanation has not been called");
    if (n < 0 || n > c->n_local) return fail(c, "n exceeds the context capacity");
    HIPC(c, hipSetDevice(c->device));
    const size_t items = (size_t)n * c->meth.n_data;
    if (flows) HIPC(c, hipMemcpyAsync(flows, c->d_mflows, items * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) HIPC(c, hipMemcpyAsync(status, c->d_mstatus, items * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smc_set_early_reject(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->early_reject = enable != 0;
    return 0;
}

int smc_set_exact_pow(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->exact_pow = enable != 0;
    return 0;
}

int smc_set_in_phase(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->in_phase = enable != 0;
    return 0;
}

int smc_debug_set_order(smc_ctx *c, const int32_t *order, int64_t n, int patience) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    if (!order) {
        c->order_debug = 0;
        return 0;
    }
    if (n != c->n_local) return fail(c, "smc_debug_set_order: n must be the context's particle count");
    HIPC(c, hipMemcpyAsync(c->d_order, order, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->order_debug = 1;
    c->order_debug_patience = patience;
    return 0;
}
int smc_set_cost_order(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->cost_order = enable != 0;
    return 0;
}
int smc_set_fast_tail(smc_ctx *c, int enable) {
    if (!c) return smc_fail(nullptr, "NULL context");
    c->fast_tail = enable != 0;
    return 0;
}
int smc_set_stiff_first(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->stiff_first = enable != 0;
    return 0;
}

int smc_set_resampling(smc_ctx *c, int scheme) {
    if (!c) return fail(nullptr, "NULL context");
    if (scheme != SMC_RESAMPLE_RESIDUAL_SYSTEMATIC && scheme != SMC_RESAMPLE_SYSTEMATIC && scheme != SMC_RESAMPLE_MULTINOMIAL)
        return fail(c, "Unknown resampling scheme");
    if (scheme == SMC_RESAMPLE_MULTINOMIAL && !c->d_mn_thr) {
        HIPC(c, hipSetDevice(c->device));
        const int64_t m = c->n_global + 1, ntm = (m + kScanTile - 1) / kScanTile;
        HIPC(c, hipMalloc(&c->d_mn_thr, (size_t)m * sizeof(double)));
        HIPC(c, hipMalloc(&c->d_mn_blk, (size_t)(ntm + 1) * sizeof(double)));
    }
    c->resampling = scheme;
    return 0;
}

// ---- particle movement -------------------------------------------------------------------------
static int check_set(smc_ctx *c, int set, int64_t n) {
    if (!c) return fail(nullptr, "NULL context");
    if (set != SMC_SET_PRED && set != SMC_SET_FILT) return fail(c, "bad particle set id");
    if (n < 0 || n > c->n_local) return fail(c, "n exceeds the context capacity");
    return 0;
}

int smc_upload_particles(smc_ctx *c, int set, const double *aos, int64_t n) {
    if (check_set(c, set, n)) return 1;
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    HIPC(c, hipMemcpyAsync(c->d_stage, aos, (size_t)n * c->dim * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_aos_to_soa(c, c->d_stage, c->set[set].theta, n, c->dim, c->set[set].stride);
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_download_particles(smc_ctx *c, int set, double *aos, int64_t n) {
    if (check_set(c, set, n)) return 1;
    HIPC(c, hipSetDevice(c->device));
    launch_soa_to_aos(c, c->set[set].theta, c->d_stage, n, c->dim, c->set[set].stride);
    HIPC(c, hipMemcpyAsync(aos, c->d_stage, (size_t)n * c->dim * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_upload_lk(smc_ctx *c, int set, const double *lk, int64_t n) {
    if (check_set(c, set, n)) return 1;
    if (set == SMC_SET_PRED) c->ess_max_valid = false;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(c->set[set].lk, lk, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_download_lk(smc_ctx *c, int set, double *lk, int64_t n) {
    if (check_set(c, set, n)) return 1;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(lk, c->set[set].lk, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_download_accept_flags(smc_ctx *c, uint8_t *flags, int64_t n) {
    if (check_set(c, SMC_SET_FILT, n)) return 1;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(flags, c->r_ac, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_download_item_info(smc_ctx *c, int32_t *info, int64_t n) {
    if (!c) return fail(nullptr, "NULL context");
    if (c->model_kind != 1 || !c->d_info || n < 0 || n > c->item_cap) return fail(c, "smc_download_item_info: Michaelis-Menten sweeps only, n <= particles of the last sweep");
    HIPC(c, hipSetDevice(c->device));
    // the last sweep wrote item (e, p) at e * n_sweep + p; the caller passes the particle count of that sweep
    HIPC(c, hipMemcpyAsync(info, c->d_info, (size_t)n * c->mm.n_ex * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
int smc_reset_accept_flags(smc_ctx *c) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemsetAsync(c->r_ac, 0, (size_t)c->n_local, c->stream));
    return 0;
}
int smc_commit_filt_to_pred(smc_ctx *c) {
    if (!c) return fail(nullptr, "NULL context");
    c->ess_max_valid = false;
    HIPC(c, hipSetDevice(c->device));
    const size_t nb = (size_t)c->n_local * sizeof(double);
    HIPC(c, hipMemcpyAsync(c->set[SMC_SET_PRED].theta, c->set[SMC_SET_FILT].theta, nb * c->dim,
                           hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->set[SMC_SET_PRED].lk, c->set[SMC_SET_FILT].lk, nb, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int smc_sample_prior_device(smc_ctx *c, uint64_t seed, int64_t global_offset) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_prior) return fail(c, "smc_set_prior has not been called");
    HIPC(c, hipSetDevice(c->device));
    launch_sample_prior(c, seed, global_offset);
    HIPC(c, hipGetLastError());
    return 0;
}

// ---- sweeps --------------------------------------------------------------------------------------
static int ensure_item_capacity(smc_ctx *c, int64_t n) {
    if (n <= c->item_cap) return 0;
    HIPC(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_sum_r2);
    (void)hipFree(c->d_info);
    (void)hipFree(c->d_stiff_list);
    c->d_sum_r2 = nullptr;
    c->d_info = nullptr;
    c->d_stiff_list = nullptr;
    c->item_cap = 0;
    HIPC(c, hipMalloc(&c->d_sum_r2, (size_t)n * c->mm.n_ex * sizeof(double)));
    HIPC(c, hipMalloc(&c->d_info, (size_t)n * c->mm.n_ex * sizeof(int)));
    HIPC(c, hipMalloc(&c->d_stiff_list, (size_t)n * sizeof(int32_t)));   // every particle at most once per sweep
    HIPC(c, hipMemsetAsync(c->d_stiff_count, 0, 4 * sizeof(unsigned), c->stream));
    c->item_cap = n;
    return 0;
}
static int counters_begin(smc_ctx *c) {
    HIPC(c, hipMemsetAsync(c->d_counters, 0, sizeof(SweepCounters), c->stream));
    return 0;
}
static int counters_end(smc_ctx *c) {
    HIPC(c, hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(SweepCounters), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (c->model_kind == 1) {   // smc_work_totals: every sweep that ends here launched the solve kernel once
        c->w_solved_items += (int64_t)c->h_counters->solved_items;
        c->w_rk_attempts += (int64_t)c->h_counters->rk_attempts;
        c->w_solve_launches += 1;
    }
    if (c->model_kind == 1 && c->pending_sweep_items > 0) {   // what the next Metropolis sweep's in-phase decision looks at
        c->last_sweep_items = c->pending_sweep_items;
        c->last_sweep_long_items = (int64_t)c->h_counters->long_items;
        c->pending_sweep_items = 0;
    }
    if (c->model_kind == 2) {   // every (particle, experiment) item the sweep asked for must have been solved exactly once
        const SweepCounters &k = *c->h_counters;
        if (k.completed_solves + k.cancelled_solves != k.expected_solves || k.unsolved_items != 0 || k.wave_split != 0) {
            char buf[320];
            snprintf(buf, sizeof buf, "methanation sweep incomplete: %llu of %llu DAE solves finished (+ %llu cancelled by the exact "
                     "early rejection), %llu live items unsolved, %llu waves split at a dequeue", k.completed_solves,
                     k.expected_solves, k.cancelled_solves, k.unsolved_items, k.wave_split);
            return fail(c, buf);
        }
    }
    return 0;
}

int smc_loglik(smc_ctx *c, int set, int64_t *n_failed, int64_t *rk_attempts) {
    if (check_set(c, set, 0)) return 1;
    if (!c->have_model) return fail(c, "smc_set_model_mm has not been called");
    HIPC(c, hipSetDevice(c->device));
    if (set == SMC_SET_PRED) c->ess_max_valid = false;
    if (counters_begin(c)) return 1;
    {
        ScopedTimer tm(c, SMC_T_LOGLIK);
        if (c->model_kind == 2)
            launch_meth_loglik(c, c->set[set].theta, c->set[set].stride, c->n_local, c->set[set].lk);
        else if (c->model_kind == 3)
            launch_user_loglik(c, c->set[set].theta, c->set[set].stride, c->n_local, c->set[set].lk);
        else
            launch_mm_loglik(c, c->set[set].theta, c->set[set].stride, c->n_local, c->set[set].lk, nullptr);
    }
    HIPC(c, hipGetLastError());
    if (c->launch_failed) { c->launch_failed = false; return 1; }
    if (counters_end(c)) return 1;
    if (n_failed) *n_failed = (int64_t)c->h_counters->n_failed;
    if (rk_attempts) *rk_attempts = (int64_t)c->h_counters->rk_attempts;
    return 0;
}

int smc_mm_loglik_host(smc_ctx *c, const double *particle, int64_t n, double *lk, double *pred, int64_t *n_failed,
                       int64_t *rk_attempts) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_model || c->model_kind != 1) return fail(c, "smc_set_model_mm has not been called");
    if (n < 0) return fail(c, "n < 0");
    HIPC(c, hipSetDevice(c->device));
    if (n_failed) *n_failed = 0;
    if (rk_attempts) *rk_attempts = 0;
    if (n == 0) return 0;
    if (n > c->hb_cap) {
        (void)hipFree(c->d_hb_theta);
        (void)hipFree(c->d_hb_lk);
        c->d_hb_theta = c->d_hb_lk = nullptr;
        c->hb_cap = 0;
        HIPC(c, hipMalloc(&c->d_hb_theta, (size_t)n * 3 * sizeof(double) * 2));  // AoS staging + SoA
        HIPC(c, hipMalloc(&c->d_hb_lk, (size_t)n * sizeof(double)));
        c->hb_cap = n;
    }
    const size_t per = (size_t)c->mm.n_ex * c->mm.n_t;
    if (pred && n > c->hb_pred_cap) {
        (void)hipFree(c->d_hb_pred);
        c->d_hb_pred = nullptr;
        c->hb_pred_cap = 0;
        HIPC(c, hipMalloc(&c->d_hb_pred, (size_t)n * per * sizeof(double)));
        c->hb_pred_cap = n;
    }
    if (ensure_item_capacity(c, n)) return 1;
    double *aos = c->d_hb_theta, *soa = c->d_hb_theta + (size_t)c->hb_cap * 3;
    HIPC(c, hipMemcpyAsync(aos, particle, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_aos_to_soa(c, aos, soa, n, 3, n);
    if (counters_begin(c)) return 1;
    {
        ScopedTimer tm(c, SMC_T_LOGLIK);
        launch_mm_loglik(c, soa, n, n, c->d_hb_lk, pred ? c->d_hb_pred : nullptr);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(lk, c->d_hb_lk, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pred)
        HIPC(c, hipMemcpyAsync(pred, c->d_hb_pred, (size_t)n * per * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (counters_end(c)) return 1;
    if (n_failed) *n_failed = (int64_t)c->h_counters->n_failed;
    if (rk_attempts) *rk_attempts = (int64_t)c->h_counters->rk_attempts;
    return 0;
}

static int fetch_small(smc_ctx *c, int n) {
    HIPC(c, hipMemcpyAsync(c->h_small, c->d_small, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smc_max_lk_local(smc_ctx *c, double *max_lk) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_MAX);
        launch_max(c, c->set[SMC_SET_PRED].lk, c->n_local, c->d_small);
    }
    HIPC(c, hipGetLastError());
    if (fetch_small(c, 1)) return 1;
    *max_lk = c->h_small[0];
    return 0;
}

int smc_ess_partials(smc_ctx *c, double max_lk, const double *gm, int n_cand, double *sum_w, double *sum_w2) {
    if (!c) return fail(nullptr, "NULL context");
    if (n_cand < 1 || n_cand > SMC_MAX_ESS_CAND) return fail(c, "n_cand out of range");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_ESS);
        launch_ess(c, c->set[SMC_SET_PRED].lk, c->n_local, max_lk, gm, n_cand, c->d_small);
    }
    HIPC(c, hipGetLastError());
    if (fetch_small(c, 2 * ess_padded_k(n_cand))) return 1;
    for (int k = 0; k < n_cand; ++k) {
        sum_w[k] = c->h_small[2 * k];
        sum_w2[k] = c->h_small[2 * k + 1];
    }
    return 0;
}

// ---- reductions over ALL ranks: kernel -> ncclAllReduce on the device buffer -> ONE read-back ----------------
// (the *_local calls above return this rank's partial and leave the reduction to a host-side communicator: three PCIe
// hops per collective; these keep the partial on the device)
}  // extern "C"
static int dev_allreduce(smc_ctx *c, void *buf, size_t n, ncclDataType_t dt, ncclRedOp_t op) {
    if (!c->nccl_comm) {
        if (peers_on(c)) {
            if (dt != ncclDouble || (op != ncclSum && op != ncclMax)) return fail(c, "local-peer collective: only SUM / MAX of doubles");
            return peer_allreduce(c, static_cast<double *>(buf), n, op == ncclMax, false, nullptr);
        }
        if (c->world > 1) return fail(c, "world > 1 but smc_comm_init has not been called (the *_global entry points need RCCL)");
        return 0;
    }
    NCCLC(c, ncclAllReduce(buf, buf, n, dt, op, (ncclComm_t)c->nccl_comm, c->stream));
    return 0;
}
// all-gather of n 8-byte words per rank from d_small[0..n) into d_small[2048 ..), then to h_small[0 .. n*world)
static int dev_allgather_words(smc_ctx *c, int n) {
    if ((size_t)n * c->world > 2048) return fail(c, "collective payload too large");
    if (!c->nccl_comm && peers_on(c)) {
        if (peer_allreduce(c, c->d_small, (size_t)n, false, true, c->d_small + 2048)) return 1;
        HIPC(c, hipMemcpyAsync(c->h_small, c->d_small + 2048, (size_t)n * c->world * 8, hipMemcpyDeviceToHost, c->stream));
    } else if (!c->nccl_comm) {
        if (c->world > 1) return fail(c, "world > 1 but smc_comm_init has not been called (the *_global entry points need RCCL)");
        HIPC(c, hipMemcpyAsync(c->h_small, c->d_small, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    } else {
        NCCLC(c, ncclAllGather(c->d_small, c->d_small + 2048, (size_t)n, ncclUint64, (ncclComm_t)c->nccl_comm, c->stream));
        HIPC(c, hipMemcpyAsync(c->h_small, c->d_small + 2048, (size_t)n * c->world * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" {

int smc_max_lk_global(smc_ctx *c, double *max_lk) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_MAX);
        launch_max(c, c->set[SMC_SET_PRED].lk, c->n_local, c->d_small);
    }
    HIPC(c, hipGetLastError());
    if (dev_allreduce(c, c->d_small, 1, ncclDouble, ncclMax)) return 1;
    if (fetch_small(c, 1)) return 1;
    *max_lk = c->h_small[0];
    return 0;
}

int smc_ess_partials_global(smc_ctx *c, double max_lk, const double *gm, int n_cand, double *sum_w, double *sum_w2) {
    if (!c) return fail(nullptr, "NULL context");
    if (n_cand < 1 || n_cand > SMC_MAX_ESS_CAND) return fail(c, "n_cand out of range");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_ESS);
        launch_ess(c, c->set[SMC_SET_PRED].lk, c->n_local, max_lk, gm, n_cand, c->d_small);
    }
    HIPC(c, hipGetLastError());
    const int nv = 2 * ess_padded_k(n_cand);
    if (dev_allreduce(c, c->d_small, (size_t)nv, ncclDouble, ncclSum)) return 1;
    if (fetch_small(c, nv)) return 1;
    for (int k = 0; k < n_cand; ++k) {
        sum_w[k] = c->h_small[2 * k];
        sum_w2[k] = c->h_small[2 * k + 1];
    }
    return 0;
}

// Micmem_SMC_main.py:116-134 for a whole batch of back-off candidates with ONE read-back: max(lk) -> [all-reduce MAX] stays
// on the device, up to two 16-candidate ESS passes read it from there back to back -> [ONE all-reduce SUM] -> one copy of
// (max_lk, sums) to the host.  Round 2 needed one round trip for the maximum and one per 16 candidates (318 launches and as
// many synchronisations for the 3 044 search iterations of the benchmark's 20 runs; the first step alone needs 17-18
// candidates).  with_max == 0: max(lk) is still in place from the previous call of the same search (candidates 33 ...).
int smc_ess_search_global(smc_ctx *c, const double *gm, int n_cand, int with_max, double *max_lk, double *sum_w, double *sum_w2) {
    if (!c) return fail(nullptr, "NULL context");
    if (n_cand < 1 || n_cand > 2 * SMC_MAX_ESS_CAND) return fail(c, "smc_ess_search_global: n_cand must be 1 .. 32");
    HIPC(c, hipSetDevice(c->device));
    // The search's own buffer: with_max == 0 relies on max(lk) still being where the previous call left it, and
    // every other entry point with a small result writes d_small - a host all-reduce between two calls of one search must
    // not be able to change the maximum the second pass normalises with.  Nothing else writes d_ess.
    constexpr int kMaxAt = 0, kSumsAt = 8;           // d_ess layout: [kMaxAt] max(lk), [kSumsAt ..) 2 x 32 sums (padded)
    double *S = c->d_ess;
    if (with_max) {
        {
            ScopedTimer tm(c, SMC_T_MAX);
            launch_max(c, c->set[SMC_SET_PRED].lk, c->n_local, S + kMaxAt);
        }
        if (dev_allreduce(c, S + kMaxAt, 1, ncclDouble, ncclMax)) return 1;
        c->ess_max_valid = true;
    } else if (!c->ess_max_valid) {
        return fail(c, "smc_ess_search_global: with_max == 0 needs an earlier call with with_max != 0 on the same lk");
    }
    const int n_a = n_cand < SMC_MAX_ESS_CAND ? n_cand : SMC_MAX_ESS_CAND, n_b = n_cand - n_a;
    const int pad_a = ess_padded_k(n_a), pad_b = n_b > 0 ? ess_padded_k(n_b) : 0;
    {
        ScopedTimer tm(c, SMC_T_ESS);
        launch_ess(c, c->set[SMC_SET_PRED].lk, c->n_local, 0.0, gm, n_a, S + kSumsAt, S + kMaxAt);
        if (n_b > 0) launch_ess(c, c->set[SMC_SET_PRED].lk, c->n_local, 0.0, gm + n_a, n_b, S + kSumsAt + 2 * pad_a, S + kMaxAt);
    }
    HIPC(c, hipGetLastError());
    const int nv = 2 * (pad_a + pad_b);
    if (dev_allreduce(c, S + kSumsAt, (size_t)nv, ncclDouble, ncclSum)) return 1;
    HIPC(c, hipMemcpyAsync(c->h_ess + kMaxAt, S + kMaxAt, (size_t)(kSumsAt - kMaxAt + nv) * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (max_lk) *max_lk = c->h_ess[kMaxAt];
    for (int k = 0; k < n_cand; ++k) {
        const int at = kSumsAt + (k < n_a ? 2 * k : 2 * pad_a + 2 * (k - n_a));
        sum_w[k] = c->h_ess[at];
        sum_w2[k] = c->h_ess[at + 1];
    }
    return 0;
}

int smc_resample_global(smc_ctx *c, double max_lk, double gm, double sum_weight_global, double wrand, int first_step,
                        int64_t *n_offspring, int64_t *count_sum) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    const int W = c->world, R = c->rank;
    // 1. residual sums and integer copies of this rank; every rank learns all of them
    {
        ScopedTimer tm(c, SMC_T_RESAMPLE);
        launch_resample_phase1(c, max_lk, gm, sum_weight_global);
        HIPC(c, hipMemcpyAsync(c->d_small, c->d_blk_r + c->n_tiles, 8, hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->d_small + 1, c->d_blk_c + c->n_tiles, 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPC(c, hipGetLastError());
    if (dev_allgather_words(c, 2)) return 1;
    double prefix = 0.0;                     // running sum in rank order, as the sequential loop would form it (:165-167)
    int64_t csum = 0;
    for (int q = 0; q < W; ++q) {
        int64_t cq;
        memcpy(&cq, c->h_small + 2 * q + 1, sizeof cq);
        csum += cq;
        if (q < R) prefix = prefix + c->h_small[2 * q];
    }
    // 2. offspring of this rank given the residual mass below it; every rank learns all counts
    {
        ScopedTimer tm(c, SMC_T_RESAMPLE);
        launch_resample_phase2(c, max_lk, gm, sum_weight_global, prefix, wrand);
        HIPC(c, hipMemcpyAsync(c->d_small, c->d_blk_o + c->n_tiles, 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPC(c, hipGetLastError());
    if (dev_allgather_words(c, 1)) return 1;
    std::vector<int64_t> o_all(W), bases(W);
    int64_t tot = 0;
    for (int q = 0; q < W; ++q) {
        memcpy(&o_all[q], c->h_small + q, sizeof(int64_t));
        bases[q] = tot;
        tot += o_all[q];
    }
    // 3. gather + exchange
    if (smc_resample_phase3(c, bases.data(), o_all.data(), first_step)) return 1;
    if (n_offspring) *n_offspring = tot;
    if (count_sum) *count_sum = csum;
    return 0;
}

// Micmem_SMC_main.py:147-184 ENQUEUED: with one rank nothing of the resampling needs the host - the residual prefix of the lower
// ranks is zero, the offspring total stays on the device and the gather kernel reads it there - so the three phases go onto the
// stream without a synchronisation (smc_resample_global: two) and the Metropolis sweeps can be enqueued right behind them.  The
// two numbers the driver logs (n_tmp = N - sum trunc(w N), :176; the offspring total) are fetched by smc_resample_result, which
// the driver calls after its next synchronisation.  Several ranks: the exchange plan needs the counts on the host, so this IS
// smc_resample_global and the result call returns what it found.
int smc_resample_enqueue(smc_ctx *c, double max_lk, double gm, double sum_weight_global, double wrand, int first_step) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    if (c->world > 1) {
        int64_t o = 0, cs = 0;
        if (smc_resample_global(c, max_lk, gm, sum_weight_global, wrand, first_step, &o, &cs)) return 1;
        c->h_rs[0] = cs;
        c->h_rs[1] = o;
        c->rs_pending = 2;
        return 0;
    }
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    {
        ScopedTimer tm(c, SMC_T_RESAMPLE);
        launch_resample_phase1(c, max_lk, gm, sum_weight_global);
        launch_resample_phase2(c, max_lk, gm, sum_weight_global, 0.0, wrand);
        launch_resample_gather_all(c, first_step, c->d_rs);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->h_rs, c->d_rs, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    c->rs_pending = 1;
    return 0;
}
int smc_resample_result(smc_ctx *c, int64_t *n_offspring, int64_t *count_sum) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->rs_pending) return fail(c, "smc_resample_result: no smc_resample_enqueue before it");
    HIPC(c, hipSetDevice(c->device));
    if (c->rs_pending == 1) HIPC(c, hipStreamSynchronize(c->stream));   // as a rule the stream has drained long ago
    c->rs_pending = 0;
    if (count_sum) *count_sum = c->h_rs[0];
    if (n_offspring) *n_offspring = c->h_rs[1];
    if (c->h_rs[1] > c->n_global)
        return fail(c, "resampling produced more offspring than particles (the reference raises IndexError, "
                       "Micmem_SMC_main.py:180)");
    return 0;
}

// ---- pinned host buffers for results ------------------------------------------------------------------
// A download into pageable memory is staged through the runtime's own pinned buffers and a host-side copy (~3 ms for the 32 MB
// of 10^6 final particles); into memory from here the DMA engine writes directly.  The memory belongs to the process, not to a
// context: it stays valid after smc_destroy, until smc_pinned_free.
int smc_pinned_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return fail(nullptr, "smc_pinned_alloc: bad arguments");
    *out = nullptr;
    HIPC((smc_ctx *)nullptr, hipHostMalloc(out, bytes));
    return 0;
}
int smc_pinned_free(void *p) {
    if (p) HIPC((smc_ctx *)nullptr, hipHostFree(p));
    return 0;
}

// ---- resampling ------------------------------------------------------------------------------------
int smc_resample_phase1(smc_ctx *c, double max_lk, double gm, double sum_weight_global, double *residual_sum_local,
                        int64_t *count_sum_local) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_RESAMPLE);
        launch_resample_phase1(c, max_lk, gm, sum_weight_global);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->h_small, c->d_blk_r + c->n_tiles, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(c->h_small + 1, c->d_blk_c + c->n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    *residual_sum_local = c->h_small[0];
    int64_t cs;
    memcpy(&cs, c->h_small + 1, sizeof cs);
    *count_sum_local = cs;
    return 0;
}

int smc_resample_phase2(smc_ctx *c, double max_lk, double gm, double sum_weight_global, double residual_prefix,
                        double wrand, int64_t *offspring_local) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_RESAMPLE);
        launch_resample_phase2(c, max_lk, gm, sum_weight_global, residual_prefix, wrand);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->h_small, c->d_blk_o + c->n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    int64_t o;
    memcpy(&o, c->h_small, sizeof o);
    *offspring_local = o;
    return 0;
}

int smc_download_offspring(smc_ctx *c, int64_t *p_is, int64_t n) {
    if (check_set(c, SMC_SET_PRED, n)) return 1;
    HIPC(c, hipSetDevice(c->device));
    int64_t *d_tmp = nullptr;
    HIPC(c, hipMalloc(&d_tmp, (size_t)c->n_local * sizeof(int64_t)));
    launch_offspring_from_scan(c, d_tmp);
    hipError_t e = hipMemcpyAsync(p_is, d_tmp, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tmp);
    HIPC(c, e);
    return 0;
}

static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }

// Particle blocks between ranks: per peer q one ncclSend of send_cnt[q] particles ([component][cnt] layout, d+1
// components, at d_sendbuf + send_off[q]*(d+1)) and one ncclRecv of recv_cnt[q] particles into the receive staging,
// all in one group on the context's stream; the received blocks are then spread over the FILT SoA rows starting at
// recv_row[q].  Entries with a zero count are skipped (a rank may list itself: RCCL copies locally).
static int rccl_exchange_blocks(smc_ctx *c, const std::vector<int64_t> &send_off, const std::vector<int64_t> &send_cnt,
                                const std::vector<int64_t> &recv_off, const std::vector<int64_t> &recv_cnt,
                                const std::vector<int64_t> &recv_row, int64_t recv_total) {
    const int W = (int)send_cnt.size(), d = c->dim;
    ParticleSet &F = c->set[SMC_SET_FILT];
    ncclComm_t comm = (ncclComm_t)c->nccl_comm;
    if (!comm) return fail(c, "smc_comm_init has not been called");
    if (recv_total > c->recvbuf_cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_recvbuf);
        c->d_recvbuf = nullptr;
        c->recvbuf_cap = 0;
        HIPC(c, hipMalloc(&c->d_recvbuf, (size_t)recv_total * (d + 1) * sizeof(double)));
        c->recvbuf_cap = recv_total;
    }
    NCCLC(c, ncclGroupStart());
    for (int q = 0; q < W; ++q) {
        if (send_cnt[q] > 0)
            NCCLC(c, ncclSend(c->d_sendbuf + (size_t)send_off[q] * (d + 1), (size_t)send_cnt[q] * (d + 1), ncclDouble, q, comm,
                              c->stream));
        if (recv_cnt[q] > 0)
            NCCLC(c, ncclRecv(c->d_recvbuf + (size_t)recv_off[q] * (d + 1), (size_t)recv_cnt[q] * (d + 1), ncclDouble, q, comm,
                              c->stream));
    }
    NCCLC(c, ncclGroupEnd());
    for (int q = 0; q < W; ++q) {
        if (recv_cnt[q] == 0) continue;
        const double *blk = c->d_recvbuf + (size_t)recv_off[q] * (d + 1);
        const int64_t cnt = recv_cnt[q], off = recv_row[q];
        for (int k = 0; k < d; ++k)
            HIPC(c, hipMemcpyAsync(F.theta + (size_t)k * F.stride + off, blk + (size_t)k * cnt, (size_t)cnt * sizeof(double),
                                   hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(F.lk + off, blk + (size_t)d * cnt, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice,
                               c->stream));
    }
    return 0;
}

// Rehearsal of the exchange on ONE rank: particles [row, row+cnt) of the PRED set travel through d_sendbuf, an RCCL
// send/recv pair addressed to this rank itself and the receive staging into rows [dst_row, dst_row+cnt) of the FILT set
// - the code path of smc_resample_phase3's step 3 with real RCCL calls (needs a communicator: SMC_FORCE_RCCL=1).
int smc_debug_rccl_self_exchange(smc_ctx *c, int64_t row, int64_t cnt, int64_t dst_row) {
    if (!c) return fail(nullptr, "NULL context");
    if (row < 0 || cnt < 1 || row + cnt > c->n_local || dst_row < 0 || dst_row + cnt > c->n_local) return fail(c, "bad row range");
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    const int d = c->dim;
    ParticleSet &P = c->set[SMC_SET_PRED];
    if (cnt > c->sendbuf_cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_sendbuf);
        c->d_sendbuf = nullptr;
        c->sendbuf_cap = 0;
        HIPC(c, hipMalloc(&c->d_sendbuf, (size_t)cnt * (d + 1) * sizeof(double)));
        c->sendbuf_cap = cnt;
    }
    for (int k = 0; k < d; ++k)
        HIPC(c, hipMemcpyAsync(c->d_sendbuf + (size_t)k * cnt, P.theta + (size_t)k * P.stride + row, (size_t)cnt * sizeof(double),
                               hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(c->d_sendbuf + (size_t)d * cnt, P.lk + row, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice,
                           c->stream));
    std::vector<int64_t> so(c->world, 0), sc(c->world, 0), ro(c->world, 0), rc(c->world, 0), rr(c->world, 0);
    sc[c->rank] = cnt;
    rc[c->rank] = cnt;
    rr[c->rank] = dst_row;
    if (rccl_exchange_blocks(c, so, sc, ro, rc, rr, cnt)) return 1;
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smc_exchange_plan(int world, int rank, int64_t n_local, const int64_t *out_base_all, const int64_t *offspring_all,
                      int64_t *send_off, int64_t *send_cnt, int64_t *src_lo, int64_t *recv_off, int64_t *recv_cnt,
                      int64_t *recv_row, int64_t *own /* src_lo, cnt, row */, int64_t *stale_lo) {
    if (world < 1 || world > SMC_MAX_RANKS || rank < 0 || rank >= world || n_local < 1 || !out_base_all || !offspring_all)
        return 2;
    const ExchangePlan p = make_exchange_plan(world, rank, n_local, out_base_all, offspring_all);
    if (!p.prefix_ok) return 1;
    for (int q = 0; q < world; ++q) {
        if (send_off) send_off[q] = p.send_off[q];
        if (send_cnt) send_cnt[q] = p.send_cnt[q];
        if (src_lo) src_lo[q] = p.src_lo[q];
        if (recv_off) recv_off[q] = p.recv_off[q];
        if (recv_cnt) recv_cnt[q] = p.recv_cnt[q];
        if (recv_row) recv_row[q] = p.recv_row[q];
    }
    if (own) { own[0] = p.own_src_lo; own[1] = p.own_cnt; own[2] = p.own_row; }
    if (stale_lo) *stale_lo = p.stale_lo;
    return 0;
}

// the copies of smc_resample_phase3_pull, enqueued (peers' blocks -> this rank's FILT rows)
static int peer_pull_copies(smc_ctx *c) {
    const int W = c->world, R = c->rank, d = c->dim;
    const int64_t nl = c->n_local;
    ParticleSet &F = c->set[SMC_SET_FILT];
    for (int q = 0; q < W; ++q) {
        if (q == R) continue;
        smc_ctx *src = c->peers[q];
        const int64_t lo = imax(c->plan_base[q], R * nl), hi = imin(c->plan_base[q] + c->plan_cnt[q], (R + 1) * nl);
        if (hi <= lo) continue;
        const int64_t off = lo - R * nl, cnt = hi - lo;
        if (src->plan_send_cnt[R] != cnt) return fail(c, "loopback exchange: sender and receiver disagree on a count");
        const double *blk = src->d_sendbuf + (size_t)src->plan_send_off[R] * (d + 1);
        for (int k = 0; k < d; ++k)
            HIPC(c, hipMemcpyAsync(F.theta + (size_t)k * F.stride + off, blk + (size_t)k * cnt, (size_t)cnt * sizeof(double),
                                   hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(F.lk + off, blk + (size_t)d * cnt, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice,
                               c->stream));
    }
    return 0;
}
// step 3 of smc_resample_phase3 among local peers with the collectives inside the engine: packed | barrier | pull | barrier
static int peer_exchange(smc_ctx *c) {
    HIPC(c, hipEventRecord(c->peer_pack_ev, c->stream));
    if (peer_host_barrier(c)) return 1;            // every rank has packed (recorded) and published its plan
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank) HIPC(c, hipStreamWaitEvent(c->stream, c->peers[q]->peer_pack_ev, 0));
    if (peer_pull_copies(c)) return 1;
    HIPC(c, hipEventRecord(c->peer_pull_ev, c->stream));
    c->peer_pull_recorded = true;
    return peer_host_barrier(c);                   // nobody packs again before every pull is at least enqueued and recorded
}

int smc_resample_phase3(smc_ctx *c, const int64_t *out_base_all, const int64_t *offspring_all, int first_step) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    const int W = c->world, R = c->rank, d = c->dim;
    const int64_t nl = c->n_local;
    ParticleSet &F = c->set[SMC_SET_FILT];
    const ExchangePlan plan = make_exchange_plan(W, R, nl, out_base_all, offspring_all);   // exchange_plan.h
    if (!plan.prefix_ok) return fail(c, "out_base_all must be the exclusive prefix of offspring_all");
    if (plan.total > c->n_global)
        return fail(c, "resampling produced more offspring than particles (the reference raises IndexError, "
                       "Micmem_SMC_main.py:180)");
    ScopedTimer tm(c, SMC_T_RESAMPLE);

    if (W > 1 && peers_on(c))                 // the peers' pulls out of this rank's send staging at the previous exchange must be
        for (int q = 0; q < W; ++q)           // over before it is packed (or reallocated) again
            if (q != R && c->peers[q]->peer_pull_recorded) HIPC(c, hipStreamWaitEvent(c->stream, c->peers[q]->peer_pull_ev, 0));
    if (plan.send_total > c->sendbuf_cap) {   // how much leaves this rank
        HIPC(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_sendbuf);
        c->d_sendbuf = nullptr;
        c->sendbuf_cap = 0;
        HIPC(c, hipMalloc(&c->d_sendbuf, (size_t)plan.send_total * (d + 1) * sizeof(double)));
        c->sendbuf_cap = plan.send_total;
    }
    if (W > 1 && !c->nccl_comm && c->peers.empty()) return fail(c, "world > 1 but smc_comm_init has not been called");

    // 1. gather: own slots straight into p_filt / lk1, remote slots into the send staging
    if (plan.own_cnt > 0)
        launch_resample_gather(c, plan.own_src_lo, plan.own_src_lo + plan.own_cnt, F.theta, F.stride, F.lk, plan.own_row);
    for (int q = 0; q < W; ++q) {
        const int64_t cnt = plan.send_cnt[q];
        if (cnt <= 0) continue;
        double *blk = c->d_sendbuf + (size_t)plan.send_off[q] * (d + 1);  // [component][cnt], component d = lk
        launch_resample_gather(c, plan.src_lo[q], plan.src_lo[q] + cnt, blk, cnt, blk + (size_t)d * cnt, 0);
    }
    // 2. rows nobody writes (total < N): what the reference's persistent buffers hold
    if (plan.stale_lo < nl) launch_resample_stale(c, plan.stale_lo, nl, first_step);
    HIPC(c, hipGetLastError());
    c->plan_send_off = plan.send_off;
    c->plan_send_cnt = plan.send_cnt;
    c->plan_base.assign(out_base_all, out_base_all + W);
    c->plan_cnt.assign(offspring_all, offspring_all + W);
    if (W > 1 && peers_on(c) && !c->nccl_comm) return peer_exchange(c);   // rehearsal with the collectives inside the engine
    if (W > 1 && !c->peers.empty()) {  // loopback rehearsal: the pull happens after a barrier between the ranks
        HIPC(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    // 3. exchange: ONE contiguous send and ONE contiguous recv per peer ([component][cnt] blocks), then the
    //    received blocks are spread over the SoA rows by device-to-device row copies on the same stream
    if (W > 1)
        return rccl_exchange_blocks(c, plan.send_off, plan.send_cnt, plan.recv_off, plan.recv_cnt, plan.recv_row, plan.recv_total);
    return 0;
}

int smc_debug_set_local_peers(smc_ctx *c, smc_ctx **peers, int rank, int world) {
    if (!c) return fail(nullptr, "NULL context");
    if (world < 1 || world > SMC_MAX_RANKS || rank < 0 || rank >= world || peers[rank] != c) return fail(c, "bad peer set");
    if (c->n_global != c->n_local * world) return fail(c, "n_global must equal world * n_local");
    c->peers.assign(peers, peers + world);
    c->rank = rank;
    c->world = world;
    return 0;
}

int smc_debug_peer_collectives(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    if (c->peers.empty()) return fail(c, "smc_debug_set_local_peers has not been called");
    if (c->world > kPeerMax) return fail(c, "in-engine collectives among local peers: at most 8 ranks");
    HIPC(c, hipSetDevice(c->device));
    if (enable && !c->d_peer_contrib) {
        HIPC(c, hipMalloc(&c->d_peer_contrib, 2 * (size_t)kPeerWords * sizeof(double)));
        for (int s = 0; s < 2; ++s) HIPC(c, hipEventCreateWithFlags(&c->peer_ev[s], hipEventDisableTiming));
        HIPC(c, hipEventCreateWithFlags(&c->peer_pack_ev, hipEventDisableTiming));
        HIPC(c, hipEventCreateWithFlags(&c->peer_pull_ev, hipEventDisableTiming));
        c->peer_barrier = new (std::nothrow) PeerBarrier();
        if (!c->peer_barrier) return fail(c, "out of memory");
    }
    c->peer_collectives = enable != 0;
    return 0;
}

int smc_resample_phase3_pull(smc_ctx *c) {
    if (!c) return fail(nullptr, "NULL context");
    if (c->peers.empty()) return fail(c, "smc_debug_set_local_peers has not been called");
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    if (peer_pull_copies(c)) return 1;
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- moments -----------------------------------------------------------------------------------------
int smc_moment_sums_local(smc_ctx *c, double *sums) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_MOMENTS);
        launch_moment_sums(c, c->d_small);
    }
    HIPC(c, hipGetLastError());
    if (fetch_small(c, c->dim)) return 1;
    for (int i = 0; i < c->dim; ++i) sums[i] = c->h_small[i];
    return 0;
}
int smc_moment_centered_local(smc_ctx *c, const double *mean, double *centered) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    {
        ScopedTimer tm(c, SMC_T_MOMENTS);
        launch_moment_centered(c, mean, c->d_small);
    }
    HIPC(c, hipGetLastError());
    const int d = c->dim;
    if (fetch_small(c, d * (d + 1) / 2)) return 1;
    int k = 0;
    for (int a = 0; a < d; ++a)
        for (int b = a; b < d; ++b) {
            centered[a * d + b] = c->h_small[k];
            centered[b * d + a] = c->h_small[k];
            ++k;
        }
    return 0;
}

// ---- MH ------------------------------------------------------------------------------------------------
static int mh_finish(smc_ctx *c, int64_t *accepted_now, int64_t *accepted_ever, int64_t *n_failed, int64_t *rk_attempts) {
    HIPC(c, hipGetLastError());
    if (c->launch_failed) { c->launch_failed = false; return 1; }
    if (counters_end(c)) return 1;
    if (accepted_now) *accepted_now = (int64_t)c->h_counters->accepted_now;
    if (accepted_ever) *accepted_ever = (int64_t)c->h_counters->accepted_ever;
    if (n_failed) *n_failed = (int64_t)c->h_counters->n_failed;
    if (rk_attempts) *rk_attempts = (int64_t)c->h_counters->rk_attempts;
    return 0;
}

int smc_mh_step_host_rng(smc_ctx *c, double gamma, double mhstep_ratio, const double *noise, const double *rr, int64_t n,
                         int64_t *accepted_now, int64_t *accepted_ever, int64_t *n_failed, int64_t *rk_attempts) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_model || !c->have_prior) return fail(c, "model and prior must be set before an MH step");
    if (n != c->n_local) return fail(c, "smc_mh_step_host_rng: n must equal the context's n_local");
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    const int d = c->dim;
    HIPC(c, hipMemcpyAsync(c->d_stage, noise, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_aos_to_soa(c, c->d_stage, c->d_noise, n, d, n);
    HIPC(c, hipMemcpyAsync(c->d_rr, rr, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (counters_begin(c)) return 1;
    MHParams mh{};
    mh.gamma = gamma;
    mh.ratio = mhstep_ratio;
    mh.noise = c->d_noise;
    mh.rr = c->d_rr;
    mh.device_rng = 0;
    mh.prior_mode = c->prior_mode;
    mh.pratio = c->d_pratio;
    {
        ScopedTimer tm(c, SMC_T_MH);
        if (c->model_kind == 2) launch_meth_mh(c, n, mh); else if (c->model_kind == 3) launch_user_mh(c, n, mh); else launch_mm_mh(c, n, mh);
    }
    return mh_finish(c, accepted_now, accepted_ever, n_failed, rk_attempts);
}

int smc_mh_step_device_rng(smc_ctx *c, double gamma, double mhstep_ratio, const double *transform, uint64_t seed,
                           uint64_t stream, int64_t global_offset, int64_t *accepted_now, int64_t *accepted_ever,
                           int64_t *n_failed, int64_t *rk_attempts) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_model || !c->have_prior) return fail(c, "model and prior must be set before an MH step");
    HIPC(c, hipSetDevice(c->device));
    c->moments_valid = false;   // the FILT set changes: carried moments no longer describe it
    if (counters_begin(c)) return 1;
    MHParams mh{};
    mh.gamma = gamma;
    mh.ratio = mhstep_ratio;
    mh.device_rng = 1;
    mh.prior_mode = c->prior_mode;
    mh.pratio = c->d_pratio;
    mh.seed = seed;
    mh.stream = stream;
    mh.global_offset = global_offset;
    for (int i = 0; i < c->dim * c->dim; ++i) mh.transform[i] = transform[i];
    {
        ScopedTimer tm(c, SMC_T_MH);
        if (c->model_kind == 2) launch_meth_mh(c, c->n_local, mh); else if (c->model_kind == 3) launch_user_mh(c, c->n_local, mh); else launch_mm_mh(c, c->n_local, mh);
    }
    return mh_finish(c, accepted_now, accepted_ever, n_failed, rk_attempts);
}

// Layout of the fused iteration's own buffer d_fused / h_fused (doubles): what one read-back of kFusedWords words brings to
// the host.  NOT d_small: the moments an accept kernel leaves at kV are carried to the NEXT iteration (moments_valid), and
// every other entry point that returns small results (smc_max_lk_*, smc_ess_partials*, smc_moment_*_local, the generic
// all-reduce / all-gather helpers) writes d_small[0..] - a caller that interleaved one of them between two fused
// iterations used to get a proposal covariance built from whatever that call left there (now tested:
// test_fused_iterations_survive_interleaved_small_result_calls).
constexpr int kV = 0;        // [sum y (d) | sum y y^T (d(d+1)/2) | accepted_now, accepted_ever, n_failed]: ONE all-reduce
constexpr int kSums = 64;    // two-pass start of a step: column sums, then
constexpr int kCent = 72;    //   sums centred about the global mean
constexpr int kShift = 112;  // shift vector of the carried moments = mean of the previous iteration
constexpr int kCov = 128;    // cov_m
constexpr int kXf = 192;     // its multivariate_normal factor
constexpr int kFusedWords = 256;

static int two_pass_factor(smc_ctx *c, const double *w_cov) {
    const int d = c->dim, npair = d * (d + 1) / 2;
    double *S = c->d_fused;
    ScopedTimer tm(c, SMC_T_MOMENTS);
    launch_moment_sums(c, S + kSums);
    if (dev_allreduce(c, S + kSums, (size_t)d, ncclDouble, ncclSum)) return 1;
    launch_moment_centered_dev(c, S + kSums, S + kCent);
    if (dev_allreduce(c, S + kCent, (size_t)npair, ncclDouble, ncclSum)) return 1;
    launch_mh_transform(c, S + kCent, S + kSums, w_cov, S + kShift, S + kCov, S + kXf);
    return 0;
}

// One Metropolis iteration of the device-RNG mode with everything on the stream (Micmem_SMC_main.py:212-241):
//   first iteration after the FILT set changed: column sums -> [allreduce] -> sums centred about the global mean ->
//   [allreduce] -> cov_m = cov * w_cov and its multivariate_normal factor (one thread);
//   then propose -> solve -> accept/select -> [ONE allreduce of moments + accept counts] -> ONE read-back.
// Michaelis-Menten path: the accept kernel accumulates the moments of the particles it selects (about the previous mean),
// so the following iterations start at the factor kernel - no pass over the particles, no further all-reduce (SURVEY.md
// 8(e), rows A6/A10: "fused with the accept count").
int smc_mh_iteration_device_rng(smc_ctx *c, double gamma, double mhstep_ratio, const double *w_cov, uint64_t seed,
                                uint64_t stream, int64_t global_offset, int64_t *accepted_now, int64_t *accepted_ever,
                                int64_t *n_failed, int64_t *rk_attempts_local, double *cov_m) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_model || !c->have_prior) return fail(c, "model and prior must be set before an MH step");
    if (!w_cov) return fail(c, "smc_mh_iteration_device_rng: w_cov is NULL");
    HIPC(c, hipSetDevice(c->device));
    const int d = c->dim;
    const bool mm = c->model_kind == 1;
    double *S = c->d_fused;
    if (mm && c->moments_valid) {
        ScopedTimer tm(c, SMC_T_MOMENTS);
        launch_mh_transform(c, S + kV, nullptr, w_cov, S + kShift, S + kCov, S + kXf);
    } else if (two_pass_factor(c, w_cov)) {
        return 1;
    }
    c->moments_valid = false;
    MHParams mh{};
    mh.gamma = gamma;
    mh.ratio = mhstep_ratio;
    mh.device_rng = 1;
    mh.prior_mode = c->prior_mode;
    mh.pratio = c->d_pratio;
    mh.seed = seed;
    mh.stream = stream;
    mh.global_offset = global_offset;
    mh.transform_dev = S + kXf;
    if (mm) {      // the propose kernel clears counters and queue, the accept kernel leaves the moment rows in d_partials
        mh.zero_counters = c->d_counters;
        mh.zero_queue = c->d_queue;
        mh.moment_shift = S + kShift;
        mh.moment_rows = c->d_partials;
    } else if (counters_begin(c)) {
        return 1;
    }
    {
        ScopedTimer tm(c, SMC_T_MH);
        if (c->model_kind == 2) launch_meth_mh(c, c->n_local, mh); else if (c->model_kind == 3) launch_user_mh(c, c->n_local, mh); else launch_mm_mh(c, c->n_local, mh);
    }
    HIPC(c, hipGetLastError());
    if (c->launch_failed) { c->launch_failed = false; return 1; }
    const int nv = mm ? d + d * (d + 1) / 2 : 0;
    launch_moments_reduce(c, mm ? c->moment_rows_n : 0, nv, S + kV);
    if (dev_allreduce(c, S + kV, (size_t)nv + 3, ncclDouble, ncclSum)) return 1;
    HIPC(c, hipMemcpyAsync(c->h_fused, S, (size_t)kFusedWords * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (counters_end(c)) return 1;      // this rank's counters + the one synchronisation of the iteration
    c->moments_valid = mm;
    if (accepted_now) *accepted_now = (int64_t)c->h_fused[kV + nv];
    if (accepted_ever) *accepted_ever = (int64_t)c->h_fused[kV + nv + 1];
    if (n_failed) *n_failed = (int64_t)c->h_fused[kV + nv + 2];
    if (rk_attempts_local) *rk_attempts_local = (int64_t)c->h_counters->rk_attempts;
    if (cov_m)
        for (int i = 0; i < d * d; ++i) cov_m[i] = c->h_fused[kCov + i];
    return 0;
}

// A BATCH of Metropolis iterations with the loop control of Micmem_SMC_main.py:243-249 on the device: the
// host enqueues n_iter iterations back to back and synchronises ONCE.  Between two iterations mh_control_kernel (one block,
// stage_kernels.hip) does what the driver's Python did on the counts it had just read back: break when r_ac.sum() > r_th * N,
// halve mhstep_ratio when r_ac.sum() < r_threshold_min * N, and form the next iteration's cov_m and factor; every kernel of an
// iteration after the break returns at once (ctl->stop).  With RCCL the counts this kernel reads were all-reduced on the
// stream just before it, so every rank takes the same decision and the enqueued collectives stay matched.  Stream order per
// iteration (one rank):  control -> propose -> [cost_hist, cost_offsets, cost_scatter] -> solve -> accept  - the row reduction
// of the moments and the factor live in the control kernel; several ranks:  ... -> accept -> moments_reduce -> ncclAllReduce
// -> control.  What the host decides once per batch instead of once per sweep: in phase / cost order (from the last sweep
// before the batch); neither can change a result.
int smc_mh_sweeps_device_rng(smc_ctx *c, double gamma, double mhstep_ratio, const double *w_cov, uint64_t seed, uint64_t stream0,
                             int n_iter, double thr_stop, double thr_halve, int64_t global_offset, int *n_done, int *stopped,
                             double *ratio_next, int64_t *accepted_now, int64_t *accepted_ever, int64_t *n_failed,
                             int64_t *rk_attempts_local, double *ratio_used, double *cov_m, int64_t *sweep_counters) {
    if (!c) return fail(nullptr, "NULL context");
    if (!c->have_model || !c->have_prior) return fail(c, "model and prior must be set before an MH step");
    if (c->model_kind != 1 && c->model_kind != 2)
        return fail(c, "smc_mh_sweeps_device_rng: Michaelis-Menten and methanation models (a user model takes smc_mh_iteration_device_rng)");
    if (!w_cov) return fail(c, "smc_mh_sweeps_device_rng: w_cov is NULL");
    if (n_iter < 1 || n_iter > kMHBatchMax) return fail(c, "smc_mh_sweeps_device_rng: n_iter must be 1 .. 32");
    if (c->world > 1 && !can_reduce(c)) return fail(c, "world > 1 but smc_comm_init has not been called (the batch needs RCCL)");
    HIPC(c, hipSetDevice(c->device));
    const bool mm = c->model_kind == 1;
    const int d = c->dim, npair = d * (d + 1) / 2;
    const int nv = mm ? d + npair : 0;         // carried moments in front of the counts (Michaelis-Menten accept kernel only)
    const bool one_rank = !can_reduce(c);      // a one-rank RCCL communicator (tests) takes the several-ranks path as well
    double *S = c->d_fused;
    // where an iteration's [moments | accepted_now, accepted_ever, n_failed] vector lives: Michaelis-Menten S + kV (carried
    // moments); the other models right behind the centred sums of the two-pass covariance, so that ONE all-reduce takes both
    double *vec = mm ? S + kV : S + kCent + npair;
    MHControlArgs a{};
    a.ctl = c->d_mhctl;
    a.log = c->d_mhlog;
    a.ratio0 = mhstep_ratio;
    a.thr_stop = thr_stop;
    a.thr_halve = thr_halve;
    a.nv = nv;
    a.vec = vec;
    a.counters = c->d_counters;
    a.n_global = (double)c->n_global;
    a.shift_io = S + kShift;
    a.cov_out = S + kCov;
    a.xform_out = S + kXf;
    const int parity0 = c->stiff_parity;
    // a failure in the middle of the batch (a launch, a collective): what is already enqueued drains, and the list parity goes
    // back to where the batch found it - the counters of lists that were never consumed are cleared by the next builder anyway,
    // but a caller that goes on after the error must not find the parity toggled an unknown number of times
    auto bail = [&]() {
        (void)hipStreamSynchronize(c->stream);
        c->stiff_parity = parity0;
        c->moments_valid = false;
        return 1;
    };
    // np.cov's two-pass route over the FILT set as it is now (:212): column sums -> [all-reduce] -> sums centred about the global
    // mean [+ `extra` words behind them] -> [all-reduce]
    auto two_pass = [&](int extra) {
        launch_moment_sums(c, S + kSums);
        if (dev_allreduce(c, S + kSums, (size_t)d, ncclDouble, ncclSum)) return 1;
        launch_moment_centered_dev(c, S + kSums, S + kCent);
        return dev_allreduce(c, S + kCent, (size_t)(npair + extra), ncclDouble, ncclSum);
    };
    ScopedTimer tm_mh(c, SMC_T_MH);
    {   // the first iteration's factor: carried moments, or the two-pass route when the FILT set changed since
        ScopedTimer tm(c, SMC_T_MOMENTS);
        a.mode = kCtlInit | kCtlTransform;
        a.iteration = 0;
        if (mm && c->moments_valid) {
            a.mom = S + kV;
            a.sums = nullptr;
        } else {
            if (two_pass(0)) return bail();
            a.mom = S + kCent;
            a.sums = S + kSums;
        }
        launch_mh_control(c, a, w_cov);
    }
    c->moments_valid = false;
    for (int i = 0; i < n_iter; ++i) {
        MHParams mh{};
        mh.gamma = gamma;
        mh.ratio = mhstep_ratio;       // not read: the kernels take ctl->ratio
        mh.device_rng = 1;
        mh.prior_mode = c->prior_mode;
        mh.pratio = c->d_pratio;
        mh.seed = seed;
        mh.stream = stream0 + (uint64_t)i;
        mh.global_offset = global_offset;
        mh.transform_dev = S + kXf;
        mh.zero_counters = c->d_counters;
        mh.ctl = c->d_mhctl;
        if (mm) {
            mh.zero_queue = c->d_queue;
            mh.moment_shift = S + kShift;
            mh.moment_rows = c->d_partials;
            launch_mm_mh(c, c->n_local, mh);
        } else {
            launch_meth_mh(c, c->n_local, mh);
        }
        if (hipGetLastError() != hipSuccess) { fail(c, "smc_mh_sweeps_device_rng: a kernel launch of the batch failed"); return bail(); }
        if (c->launch_failed) { c->launch_failed = false; return bail(); }
        const bool transform_next = i + 1 < n_iter;
        ScopedTimer tm(c, SMC_T_MOMENTS);
        if (mm) {
            if (!one_rank) {
                launch_moments_reduce(c, c->moment_rows_n, nv, vec, c->d_mhctl);
                if (dev_allreduce(c, vec, (size_t)nv + 3, ncclDouble, ncclSum)) return bail();
            }
            a.rows = one_rank ? c->d_partials : nullptr;
            a.n_rows = c->moment_rows_n;
            a.counts_local = 0;
            a.mom = S + kV;
            a.sums = nullptr;
        } else {
            // no carried moments: the next iteration's covariance is the two-pass one of the FILT set this iteration's accept
            // kernel has just left - its second all-reduce also carries the counts.  (Computed even when the loop then turns
            // out to have ended: the collectives of all ranks stay matched; the values are not used.)
            a.rows = nullptr;
            a.n_rows = 0;
            a.counts_local = one_rank ? 1 : 0;
            if (!one_rank) launch_moments_reduce(c, 0, 0, vec, c->d_mhctl);
            if (transform_next) {
                if (two_pass(one_rank ? 0 : 3)) return bail();
            } else if (!one_rank && dev_allreduce(c, vec, 3, ncclDouble, ncclSum)) {
                return bail();
            }
            a.mom = S + kCent;
            a.sums = S + kSums;
        }
        a.mode = kCtlDecide | (transform_next ? kCtlTransform : 0);
        a.iteration = i + 1;
        launch_mh_control(c, a, w_cov);
    }
    if (hipGetLastError() != hipSuccess) { fail(c, "smc_mh_sweeps_device_rng: a kernel launch of the batch failed"); return bail(); }
    HIPC(c, hipMemcpyAsync(c->h_mhlog, c->d_mhlog, (size_t)(n_iter + 1) * sizeof(MHLogEntry), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(c->h_mhctl, c->d_mhctl, sizeof(MHControl), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(c->h_fused, S, (size_t)kFusedWords * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));          // the one synchronisation of the batch
    const int done = c->h_mhctl->n_done;
    if (done < 1 || done > n_iter) return fail(c, "smc_mh_sweeps_device_rng: the device reports an impossible iteration count");
    c->moments_valid = mm;             // S + kV: moments of the particles the last iteration that ran selected
    *c->h_counters = c->h_mhlog[done - 1].snap;        // what smc_meth_sweep_counters / smc_meth_sweep_check report: the last sweep that ran
    if (mm) {
        c->stiff_parity = parity0 ^ (done & 1);     // the lists of the iterations after the break were never built: their counters
                                                    // are as the last iteration that ran left them (cleared for its successor)
        c->pending_sweep_items = 0;
        c->last_sweep_items = c->n_local * c->mm.n_ex;
        c->last_sweep_long_items = (int64_t)c->h_mhlog[done - 1].long_items;
        c->w_solve_launches += done;
        c->w_noop_launches += n_iter - done;
    }
    if (n_done) *n_done = done;
    if (stopped) *stopped = c->h_mhctl->stop;
    if (ratio_next) *ratio_next = c->h_mhctl->ratio;
    static_assert(sizeof(SweepCounters) == SMC_SWEEP_COUNTER_WORDS * sizeof(unsigned long long), "include/smc_hip.h: SMC_SWEEP_COUNTER_WORDS");
    for (int i = 0; i < done; ++i) {
        const MHLogEntry &e = c->h_mhlog[i];
        if (mm) {
            c->w_solved_items += (int64_t)e.solved_items;
            c->w_rk_attempts += (int64_t)e.rk_attempts;
        } else {   // every (particle, experiment) item the sweep asked for must have been solved or cancelled exactly once
            const SweepCounters &k = e.snap;
            if (k.completed_solves + k.cancelled_solves != k.expected_solves || k.unsolved_items != 0 || k.wave_split != 0) {
                char buf[320];
                snprintf(buf, sizeof buf, "methanation sweep %d of the batch incomplete: %llu of %llu DAE solves finished (+ %llu cancelled "
                         "by the exact early rejection), %llu live items unsolved, %llu waves split at a dequeue", i, k.completed_solves,
                         k.expected_solves, k.cancelled_solves, k.unsolved_items, k.wave_split);
                return fail(c, buf);
            }
        }
        if (accepted_now) accepted_now[i] = (int64_t)e.accepted_now;
        if (accepted_ever) accepted_ever[i] = (int64_t)e.accepted_ever;
        if (n_failed) n_failed[i] = (int64_t)e.n_failed;
        if (rk_attempts_local) rk_attempts_local[i] = (int64_t)e.rk_attempts;
        if (ratio_used) ratio_used[i] = e.ratio;
        if (cov_m)
            for (int q = 0; q < d * d; ++q) cov_m[(size_t)i * d * d + q] = e.cov[q];
        if (sweep_counters) memcpy(sweep_counters + (size_t)i * SMC_SWEEP_COUNTER_WORDS, &e.snap, sizeof(SweepCounters));
    }
    return 0;
}

// Stage A6 + the factor of A7 alone, over all ranks: cov_m and its multivariate_normal factor from the FILT set (no model
// needed) - the two-pass start of smc_mh_iteration_device_rng with a read-back.
int smc_proposal_factor_device(smc_ctx *c, const double *w_cov, double *cov_m, double *transform) {
    if (!c) return fail(nullptr, "NULL context");
    if (!w_cov) return fail(c, "smc_proposal_factor_device: w_cov is NULL");
    HIPC(c, hipSetDevice(c->device));
    const int d = c->dim;
    c->moments_valid = false;
    if (two_pass_factor(c, w_cov)) return 1;
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->h_fused, c->d_fused, (size_t)kFusedWords * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < d * d; ++i) {
        if (cov_m) cov_m[i] = c->h_fused[kCov + i];
        if (transform) transform[i] = c->h_fused[kXf + i];
    }
    return 0;
}

// the d x d factor the last smc_mh_iteration_device_rng drew its proposals with (row-major; debugging / tests)
int smc_mh_iteration_last_transform(smc_ctx *c, double *transform) {
    if (!c) return fail(nullptr, "NULL context");
    for (int i = 0; i < c->dim * c->dim; ++i) transform[i] = c->h_fused[kXf + i];
    return 0;
}

int smc_set_debug_capture(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    if (enable && !c->dbg_lk2) {
        const size_t n = (size_t)c->n_local;
        HIPC(c, hipMalloc(&c->dbg_lk2, n * sizeof(double)));
        HIPC(c, hipMalloc(&c->dbg_r, n));
    }
    c->debug_capture = enable;
    return 0;
}
int smc_download_debug_proposals(smc_ctx *c, double *aos, double *lk2, uint8_t *p0, uint8_t *r, int64_t n) {
    if (check_set(c, SMC_SET_FILT, n)) return 1;
    if (!c->dbg_lk2) return fail(c, "debug capture is not enabled");
    HIPC(c, hipSetDevice(c->device));
    // the proposals of the last MH iteration are what SMC_SET_PRED holds (the reference's p_pred, :220,228)
    launch_soa_to_aos(c, c->set[SMC_SET_PRED].theta, c->d_stage, n, c->dim, c->set[SMC_SET_PRED].stride);
    HIPC(c, hipMemcpyAsync(aos, c->d_stage, (size_t)n * c->dim * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(lk2, c->dbg_lk2, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(p0, c->d_p0, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(r, c->dbg_r, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- collectives -----------------------------------------------------------------------------------------
int smc_comm_get_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCLC((smc_ctx *)nullptr, ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return 0;
}
int smc_comm_init(smc_ctx *c, const uint8_t id[128], int rank, int world) {
    if (!c) return fail(nullptr, "NULL context");
    if (world < 1 || world > SMC_MAX_RANKS || rank < 0 || rank >= world) return fail(c, "bad rank/world");
    if (c->n_global != c->n_local * world) return fail(c, "n_global must equal world * n_local");
    HIPC(c, hipSetDevice(c->device));
    c->rank = rank;
    c->world = world;
    // a single rank needs no communicator; SMC_FORCE_RCCL=1 creates one anyway so that the RCCL call
    // sequence can be exercised on a one-GPU machine
    if (world == 1 && !getenv("SMC_FORCE_RCCL")) return 0;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm;
    NCCLC(c, ncclCommInitRank(&comm, world, u, rank));
    c->nccl_comm = comm;
    return 0;
}
int smc_comm_info(smc_ctx *c, int *count, int *user_rank, int *device) {
    if (!c) return fail(nullptr, "NULL context");
    int n = 0, r = -1, dv = -1;
    if (c->nccl_comm) {
        NCCLC(c, ncclCommCount((ncclComm_t)c->nccl_comm, &n));
        NCCLC(c, ncclCommUserRank((ncclComm_t)c->nccl_comm, &r));
        NCCLC(c, ncclCommCuDevice((ncclComm_t)c->nccl_comm, &dv));
    }
    if (count) *count = n;
    if (user_rank) *user_rank = r;
    if (device) *device = dv;
    return 0;
}

}  // extern "C"
template <typename T>
static int allreduce_impl(smc_ctx *c, T *inout, int n, ncclDataType_t dt, ncclRedOp_t op) {
    if (!c) return fail(nullptr, "NULL context");
    if (n < 0 || (size_t)n * sizeof(T) > 4096 * sizeof(double)) return fail(c, "collective payload too large");
    if (!c->nccl_comm || n == 0) {
        if (c->world > 1) return fail(c, "smc_comm_init has not been called");
        return 0;
    }
    HIPC(c, hipSetDevice(c->device));
    memcpy(c->h_small, inout, (size_t)n * sizeof(T));
    HIPC(c, hipMemcpyAsync(c->d_small, c->h_small, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    NCCLC(c, ncclAllReduce(c->d_small, c->d_small, (size_t)n, dt, op, (ncclComm_t)c->nccl_comm, c->stream));
    HIPC(c, hipMemcpyAsync(c->h_small, c->d_small, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    memcpy(inout, c->h_small, (size_t)n * sizeof(T));
    return 0;
}
extern "C" {
int smc_comm_allreduce_sum_f64(smc_ctx *c, double *inout, int n) { return allreduce_impl(c, inout, n, ncclDouble, ncclSum); }
int smc_comm_allreduce_max_f64(smc_ctx *c, double *inout, int n) { return allreduce_impl(c, inout, n, ncclDouble, ncclMax); }
int smc_comm_allreduce_sum_i64(smc_ctx *c, int64_t *inout, int n) { return allreduce_impl(c, inout, n, ncclInt64, ncclSum); }

}  // extern "C"
template <typename T>
static int allgather_impl(smc_ctx *c, const T *in, int n, T *out, ncclDataType_t dt) {
    if (!c) return fail(nullptr, "NULL context");
    if (n < 0 || (size_t)n * c->world * sizeof(T) > 2048 * sizeof(double)) return fail(c, "collective payload too large");
    if (!c->nccl_comm) {
        if (c->world > 1) return fail(c, "smc_comm_init has not been called");
        memcpy(out, in, (size_t)n * sizeof(T));
        return 0;
    }
    HIPC(c, hipSetDevice(c->device));
    T *dsend = (T *)c->d_small, *drecv = (T *)(c->d_small + 2048);
    memcpy(c->h_small, in, (size_t)n * sizeof(T));
    HIPC(c, hipMemcpyAsync(dsend, c->h_small, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    NCCLC(c, ncclAllGather(dsend, drecv, (size_t)n, dt, (ncclComm_t)c->nccl_comm, c->stream));
    HIPC(c, hipMemcpyAsync(c->h_small, drecv, (size_t)n * c->world * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->h_small, (size_t)n * c->world * sizeof(T));
    return 0;
}
extern "C" {
int smc_comm_allgather_f64(smc_ctx *c, const double *in, int n, double *out) { return allgather_impl(c, in, n, out, ncclDouble); }
int smc_comm_allgather_i64(smc_ctx *c, const int64_t *in, int n, int64_t *out) { return allgather_impl(c, in, n, out, ncclInt64); }
int smc_comm_barrier(smc_ctx *c) {
    double x = 0.0;
    return smc_comm_allreduce_sum_f64(c, &x, 1);
}

// ---- timing ----------------------------------------------------------------------------------------------
int smc_timing_enable(smc_ctx *c, int enable) {
    if (!c) return fail(nullptr, "NULL context");
    c->timing = enable;
    return 0;
}
int smc_timing_reset(smc_ctx *c) {
    if (!c) return fail(nullptr, "NULL context");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    for (int i = 0; i < SMC_T_COUNT; ++i) {
        c->t_launches[i] = 0;
        c->t_ms[i] = 0.0;
    }
    c->w_solved_items = c->w_rk_attempts = c->w_solve_launches = c->w_noop_launches = 0;
    return 0;
}
int smc_timing_get(smc_ctx *c, int which, int64_t *launches, double *total_ms) {
    if (!c) return fail(nullptr, "NULL context");
    if (which < 0 || which >= SMC_T_COUNT) return fail(c, "bad timing class");
    HIPC(c, hipSetDevice(c->device));
    timing_collect(c);
    if (launches) *launches = c->t_launches[which];
    if (total_ms) *total_ms = c->t_ms[which];
    return 0;
}
int smc_work_totals(smc_ctx *c, int64_t out[4]) {
    if (!c) return fail(nullptr, "NULL context");
    if (!out) return fail(c, "smc_work_totals: out is NULL");
    out[0] = c->w_solved_items;
    out[1] = c->w_rk_attempts;
    out[2] = c->w_solve_launches;
    out[3] = c->w_noop_launches;
    return 0;
}

}  // extern "C"
