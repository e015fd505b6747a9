// stage_kernels.hip -- the HBM-bound stages around the sweep: AoS<->SoA transposes, max / tempered
// weight sums (ESS), residual-systematic resampling (scan + binary-search gather), moments, device
// prior draw.  gfx950 only.  All floating-point reductions use fixed shapes (per-block partials in a
// fixed order + a single-block final pass), so results are reproducible run to run.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstring>

#include "philox.h"
#include "smc_internal.h"
#include "stage_kernels.h"

namespace smc {

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave64: six shuffle steps, then one LDS hop across the waves)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(v, off);
        v = (o > v || o != o) ? o : v;  // NaN-propagating like np.max
    }
    return v;
}

// block-wide sum for blockDim.x == kScanBlock (4 waves); result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *lds /* >= 4 */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) lds[w] = v;
    __syncthreads();
    T r = lds[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += lds[i];
    return r;
}

// inclusive scan across the block of one value per thread (kScanBlock threads); returns inclusive
// value, *total = block total.  Wave scan by shuffles, wave totals through LDS.
template <typename T>
__device__ __forceinline__ T block_inclusive_scan(T v, T *lds /* >= 4 */, T *total) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(v, off);
        if (l >= off) v += o;
    }
    __syncthreads();
    if (l == 63) lds[w] = v;
    __syncthreads();
    T base = 0;
    T tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
        if (i < w) base += lds[i];
        tot += lds[i];
    }
    *total = tot;
    return v + base;
}

// ---------------------------------------------------------------------------------------------
// transposes (AoS (n,d) <-> SoA d x stride)
// ---------------------------------------------------------------------------------------------
__global__ void aos_to_soa_kernel(const double *__restrict__ aos, double *__restrict__ soa, int64_t n, int d,
                                  int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const int64_t p = i / d;
    const int c = (int)(i - p * d);
    soa[c * stride + p] = aos[i];
}
__global__ void soa_to_aos_kernel(const double *__restrict__ soa, double *__restrict__ aos, int64_t n, int d,
                                  int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const int64_t p = i / d;
    const int c = (int)(i - p * d);
    aos[i] = soa[c * stride + p];
}

// ---------------------------------------------------------------------------------------------
// device prior draw (replaces sample_prior in device-RNG mode)
// ---------------------------------------------------------------------------------------------
__global__ void sample_prior_kernel(Prior prior, double *__restrict__ theta, int64_t stride, int64_t n, uint64_t seed,
                                    int64_t goff) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t g = (uint64_t)(goff + p);
    for (int c = 0; c < prior.d; ++c) {
        const u32x4 r = philox_block(seed, g, 0xFFFFFFFF00000000ull | (uint64_t)c, 0);
        double v;
        if (prior.kind[c] == SMC_PRIOR_UNIFORM) {
            v = prior.a[c] + (prior.b[c] - prior.a[c]) * u01_from(r.x, r.y);
        } else {
            const double u1 = 1.0 - u01_from(r.x, r.y), u2 = u01_from(r.z, r.w);
            v = prior.a[c] + prior.b[c] * (sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
        }
        theta[c * stride + p] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// max(lk)  (Micmem_SMC_main.py:116)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kScanBlock) max_partial_kernel(const double *__restrict__ lk, int64_t n,
                                                                 double *__restrict__ partials) {
    __shared__ double lds[4];
    double m = -__longlong_as_double(0x7ff0000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = lk[i];
        m = (v > m || v != v) ? v : m;
    }
    m = wave_max(m);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) lds[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int i = 1; i < 4; ++i) r = (lds[i] > r || lds[i] != lds[i]) ? lds[i] : r;
        partials[blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(kScanBlock) max_final_kernel(const double *__restrict__ partials, int np,
                                                               double *__restrict__ out) {
    __shared__ double lds[4];
    double m = -__longlong_as_double(0x7ff0000000000000LL);
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        const double v = partials[i];
        m = (v > m || v != v) ? v : m;
    }
    m = wave_max(m);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) lds[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int i = 1; i < 4; ++i) r = (lds[i] > r || lds[i] != lds[i]) ? lds[i] : r;
        out[0] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// tempered weight sums for K candidate increments in ONE pass over lk (Micmem_SMC_main.py:118,
// 124-134): 8 bytes of HBM per particle for all K candidates instead of ~6 NumPy passes each.
// ---------------------------------------------------------------------------------------------
struct EssCand {
    double gm[SMC_MAX_ESS_CAND];
    int k;
};

// Sums NV values per lane over the 64 lanes of a wave by a transposing butterfly: at every step a lane hands half of its
// values to its partner (lane ^ offset) and adds the partner's copy of the half it keeps, so the number of live values
// halves while the offset halves; once one value is left the remaining steps are plain exchanges.  NV - 1 + log2(64 / NV)
// additions per lane instead of 6 NV; afterwards lane l holds the wave total of value l >> (6 - log2 NV) in v[0].
// (The epilogue of the ESS pass used to be 32 block-wide sums, each with its own shuffles, LDS hop and two barriers: about
// twice the instructions of the pass over the particles itself.)
template <int NV, int CNT, int OFF>
struct WaveSumStep {   // one step of the butterfly: CNT live values, partner lane ^ OFF (recursion = guaranteed unrolling)
    static __device__ __forceinline__ void run(double (&v)[NV], int lane) {
        if constexpr (CNT > 1) {
            constexpr int half = CNT / 2;
            const bool upper = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const double keep = upper ? v[i + half] : v[i];
                const double give = upper ? v[i] : v[i + half];
                v[i] = keep + __shfl_xor(give, OFF);
            }
        } else {
            v[0] += __shfl_xor(v[0], OFF);
        }
        if constexpr (OFF > 1) WaveSumStep<NV, (CNT > 1 ? CNT / 2 : 1), OFF / 2>::run(v, lane);
    }
};
template <int NV>
__device__ __forceinline__ void wave_sum_transposed(double (&v)[NV], int lane) {
    static_assert(NV >= 1 && NV <= 64 && (NV & (NV - 1)) == 0, "NV must be a power of two");
    WaveSumStep<NV, NV, 32>::run(v, lane);
}

template <int K>
__global__ void __launch_bounds__(kScanBlock) ess_partial_kernel(const double *__restrict__ lk, int64_t n, double max_lk_val,
                                                                 const double *__restrict__ max_lk_dev, EssCand cand,
                                                                 double *__restrict__ partials) {
    // max(lk) either by value (the caller has read it back) or from device memory (fused search: the max kernels and the
    // all-reduce that precede this pass on the stream left it there - no host round trip in between)
    const double max_lk = max_lk_dev ? *max_lk_dev : max_lk_val;
    constexpr int NV = 2 * K;                          // value 2k = sum of weights, 2k + 1 = sum of squared weights
    constexpr int kShift = NV == 2 ? 5 : NV == 8 ? 3 : NV == 16 ? 2 : 1;   // 6 - log2(NV)
    static_assert(NV == 2 || NV == 8 || NV == 16 || NV == 32, "K must be 1, 4, 8 or 16");
    __shared__ double s_part[kScanBlock / 64][NV];
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = lk[i] - max_lk;  // d_lk (:118)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double w = exp(d * cand.gm[k]);  // :124
            v[2 * k] += w;
            v[2 * k + 1] += w * w;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_sum_transposed<NV>(v, lane);
    if ((lane & ((1 << kShift) - 1)) == 0) s_part[wave][lane >> kShift] = v[0];
    __syncthreads();
    if (threadIdx.x < NV) {                            // the waves of the block in fixed order
        double r = s_part[0][threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += s_part[w][threadIdx.x];
        partials[(size_t)blockIdx.x * NV + threadIdx.x] = r;
    }
}
// sums nvals interleaved values over np partial rows: one block per value (round 1 had ONE block walk through all values,
// 18.7 us for 32 values x 2048 rows - a third of an ESS pass); the order of additions within a value is unchanged
__global__ void __launch_bounds__(kScanBlock) sum_rows_final_kernel(const double *__restrict__ partials, int np,
                                                                    int nvals, double *__restrict__ out) {
    __shared__ double lds[4];
    for (int v = blockIdx.x; v < nvals; v += gridDim.x) {
        double s = 0.0;
        for (int i = threadIdx.x; i < np; i += blockDim.x) s += partials[(size_t)i * nvals + v];
        s = block_sum(s, lds);
        if (threadIdx.x == 0) out[v] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// moments for np.cov(p_filt.T, bias=True)  (Micmem_SMC_main.py:212)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kScanBlock) moment_sum_kernel(const double *__restrict__ theta, int64_t stride,
                                                                int64_t n, int d, double *__restrict__ partials) {
    __shared__ double lds[4];
    for (int c = 0; c < d; ++c) {
        double s = 0.0;
        const double *x = theta + c * stride;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            s += x[i];
        s = block_sum(s, lds);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * d + c] = s;
    }
}
struct MeanArg {
    double m[SMC_MAX_DIM];
};
// dev_sums != nullptr: the mean is formed on the device from the (already all-reduced) column sums, X.mean(axis=1) =
// sum / N - the fused Metropolis iteration (smc_mh_iteration_device_rng) never takes the sums to the host
__global__ void __launch_bounds__(kScanBlock) moment_centered_kernel(const double *__restrict__ theta, int64_t stride,
                                                                     int64_t n, int d, MeanArg mean,
                                                                     const double *__restrict__ dev_sums, double n_div,
                                                                     double *__restrict__ partials) {
    __shared__ double lds[4];
    __shared__ double s_mu[SMC_MAX_DIM];   // (writing the device-side mean into the by-value argument would move it to scratch)
    double acc[SMC_MAX_DIM * (SMC_MAX_DIM + 1) / 2];
    const int npair = d * (d + 1) / 2;
    if ((int)threadIdx.x < d) s_mu[threadIdx.x] = dev_sums ? dev_sums[threadIdx.x] / n_div : mean.m[threadIdx.x];
    __syncthreads();
    for (int k = 0; k < npair; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x[SMC_MAX_DIM];
        for (int c = 0; c < d; ++c) x[c] = theta[c * stride + i] - s_mu[c];
        int k = 0;
        for (int a = 0; a < d; ++a)
            for (int b = a; b < d; ++b) acc[k++] += x[a] * x[b];
    }
    for (int k = 0; k < npair; ++k) {
        const double s = block_sum(acc[k], lds);
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * npair + k] = s;
    }
}

// cov_m = np.cov(p_filt.T, bias=True) * w_cov (Micmem_SMC_main.py:212-215) and the factor NumPy's legacy
// multivariate_normal multiplies standard normals with (:220): (u, s, v) = svd(cov_m); x = z @ (sqrt(s)[:, None] * v).
// cov_m is symmetric, so its SVD is its eigen-decomposition with s = |lambda| (sorted descending) and the rows of v the
// eigenvectors: a cyclic Jacobi iteration in ONE thread (d <= 8: a few hundred flops, against a sweep of >= 1 ms).
// Row signs are fixed by making the largest component of every row positive (LAPACK's are arbitrary; the distribution of
// z @ A does not depend on them).  Pinned against driver.mvn_transform in tests/test_gpu_parity.py.
struct WCov {
    double w[SMC_MAX_DIM * SMC_MAX_DIM];
};
// Two sources of the second moments:
//   sums != nullptr  two-pass (np.cov's own algorithm): `mom` = the d(d+1)/2 sums centred about the mean sums / N; the mean is
//                    stored as the shift vector of the iterations that follow;
//   sums == nullptr  carried: `mom` = [sum y (d) | sum y y^T (upper)] with y = x - shift, accumulated by the accept kernel of the
//                    previous iteration about the mean of the iteration before (so |E y| << spread: no cancellation to speak
//                    of): cov = E[y y^T] - E[y] E[y]^T, and the shift moves on to the new mean.
template <int D>
__device__ __forceinline__ void mh_transform_body(const double *__restrict__ mom, const double *__restrict__ sums, double n_global,
                                                  const WCov &wcov, double *__restrict__ shift_io, double *__restrict__ cov_out,
                                                  double *__restrict__ xform_out) {
    // D is a compile-time constant and every loop below is unrolled, so A and V live in registers: the first version, with
    // run-time d and SMC_MAX_DIM arrays in scratch, took 23 us per call - on the critical path of every iteration
    constexpr int d = D;
    double A[D][D], V[D][D];
    const double inv_n = 1.0 / n_global;     // np.true_divide(1, fact), then c *= that (np.cov)
    double ey[D];
#pragma unroll
    for (int a = 0; a < d; ++a) {
        if (sums) {
            ey[a] = 0.0;
            shift_io[a] = sums[a] / n_global;
        } else {
            ey[a] = mom[a] * inv_n;
            shift_io[a] = shift_io[a] + ey[a];
        }
    }
    const double *cent = sums ? mom : mom + d;
    {
        int k = 0;
#pragma unroll
        for (int a = 0; a < d; ++a)
#pragma unroll
            for (int b = a; b < d; ++b) {
                const double v = cent[k++] * inv_n - ey[a] * ey[b];
                A[a][b] = v * wcov.w[a * d + b];
                A[b][a] = v * wcov.w[b * d + a];
            }
    }
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
        for (int b = 0; b < d; ++b) {
            cov_out[a * d + b] = A[a][b];
            V[a][b] = (a == b) ? 1.0 : 0.0;
        }
    // w_cov is symmetric in the reference (Micmem_settings.py:94-97); should a caller pass an asymmetric one, the
    // decomposition below is that of the symmetric part
#pragma unroll
    for (int a = 0; a < d; ++a)
#pragma unroll
        for (int b = a + 1; b < d; ++b) A[a][b] = A[b][a] = 0.5 * (A[a][b] + A[b][a]);
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < d; ++p) {
            diag += A[p][p] * A[p][p];
#pragma unroll
            for (int q = p + 1; q < d; ++q) off += A[p][q] * A[p][q];
        }
        if (!(off > 1e-34 * diag)) break;   // also leaves on NaN
#pragma unroll
        for (int p = 0; p < d; ++p)
#pragma unroll
            for (int q = p + 1; q < d; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                    for (int r = 0; r < d; ++r) {   // A <- A J
                        const double arp = A[r][p], arq = A[r][q];
                        A[r][p] = cs * arp - sn * arq;
                        A[r][q] = sn * arp + cs * arq;
                    }
#pragma unroll
                    for (int r = 0; r < d; ++r) {   // A <- J^T A
                        const double apr = A[p][r], aqr = A[q][r];
                        A[p][r] = cs * apr - sn * aqr;
                        A[q][r] = sn * apr + cs * aqr;
                    }
#pragma unroll
                    for (int r = 0; r < d; ++r) {
                        const double vrp = V[r][p], vrq = V[r][q];
                        V[r][p] = cs * vrp - sn * vrq;
                        V[r][q] = sn * vrp + cs * vrq;
                    }
                }
            }
    }
    // rows of the factor by |lambda| descending (svd order): rank of column e = number of columns that come before it
#pragma unroll
    for (int e = 0; e < d; ++e) {
        const double le = fabs(A[e][e]);
        int rank = 0;
#pragma unroll
        for (int f = 0; f < d; ++f) {
            const double lf = fabs(A[f][f]);
            rank += (lf > le || (lf == le && f < e)) ? 1 : 0;
        }
        const double sv = sqrt(le);
        double big = V[0][e];
#pragma unroll
        for (int c = 1; c < d; ++c)
            if (fabs(V[c][e]) > fabs(big)) big = V[c][e];
        const double sg = (big < 0.0) ? -1.0 : 1.0;
#pragma unroll
        for (int c = 0; c < d; ++c) xform_out[rank * d + c] = sv * (sg * V[c][e]);
    }
}
template <int D>
__global__ void mh_transform_kernel(const double *__restrict__ mom, const double *__restrict__ sums, double n_global, WCov wcov,
                                    double *__restrict__ shift_io, double *__restrict__ cov_out,
                                    double *__restrict__ xform_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    mh_transform_body<D>(mom, sums, n_global, wcov, shift_io, cov_out, xform_out);
}

// ---------------------------------------------------------------------------------------------
// Loop control of the Metropolis iterations ON THE DEVICE (Micmem_SMC_main.py:243-249)
// ---------------------------------------------------------------------------------------------
// The host enqueues a batch of iterations back to back; between two of them this kernel (one block) does what the driver's
// Python did after a synchronisation:
//   DECIDE     the iteration that has just ended: [W == 1: per-block moment rows of its accept kernel -> S+kV, in
//              moments_reduce_kernel's summation order]; its totals go into the batch log; `break` if r_ac.sum() > r_th * N
//              (:243-246) - ctl->stop, which every kernel of the later iterations of the batch tests first thing - else
//              mhstep_ratio *= 0.5 if r_ac.sum() < r_threshold_min * N (:247-249); a failed solve also ends the loop (the
//              driver raises, as the reference does);
//   TRANSFORM  the iteration about to start: cov_m * w_cov and its multivariate_normal factor (mh_transform_body), logged.
// The thresholds are doubles computed by the host exactly as Python evaluates `r_th * n_particle`; the counts are exact in a
// double, so `count > thr` is Python's int-vs-float comparison.  With several ranks the counts in S+kV are already summed by
// the all-reduce that precedes this kernel on the stream: every rank takes the same decision.
template <int D>
__global__ void __launch_bounds__(kScanBlock) mh_control_kernel(MHControlArgs a, WCov wcov) {
    MHControl *ctl = a.ctl;
    const int t = threadIdx.x;
    if (a.mode & kCtlInit) {
        if (t == 0) {
            ctl->stop = 0;
            ctl->n_done = 0;
            ctl->ratio = a.ratio0;
            ctl->thr_stop = a.thr_stop;
            ctl->thr_halve = a.thr_halve;
        }
    } else if (ctl->stop) {
        return;
    }
    constexpr int nv = D + D * (D + 1) / 2;      // == a.nv (the Michaelis-Menten accept kernel writes rows of d + d(d+1)/2 values)
    if (a.mode & kCtlDecide) {
        if (a.rows) {      // one rank: no all-reduce between the accept kernel and this one, so the row reduction happens here
            // moments_reduce_kernel's summation order per value - thread t adds rows t, t + 256, ... in turn, the wave sums by
            // shuffles, the four waves in order - with all nv values of a row (contiguous) taken in one pass over the rows
            double acc[nv];
#pragma unroll
            for (int v = 0; v < nv; ++v) acc[v] = 0.0;
            for (int i = t; i < a.n_rows; i += blockDim.x) {
                const double *row = a.rows + (size_t)i * nv;
#pragma unroll
                for (int v = 0; v < nv; ++v) acc[v] += row[v];
            }
            __shared__ double wpart[kScanBlock / 64][nv];
#pragma unroll
            for (int v = 0; v < nv; ++v) {
                const double ws = wave_sum(acc[v]);
                if ((t & 63) == 0) wpart[t >> 6][v] = ws;
            }
            __syncthreads();
            if (t < nv) {
                double r = wpart[0][t];
                for (int q = 1; q < kScanBlock / 64; ++q) r += wpart[q][t];
                a.vec[t] = r;
            }
            __syncthreads();      // thread 0 reads the vector back below
            if (t == 0) {
                a.vec[a.nv] = (double)a.counters->accepted_now;
                a.vec[a.nv + 1] = (double)a.counters->accepted_ever;
                a.vec[a.nv + 2] = (double)a.counters->n_failed;
            }
        } else if (a.counts_local && t == 0) {
            a.vec[a.nv] = (double)a.counters->accepted_now;
            a.vec[a.nv + 1] = (double)a.counters->accepted_ever;
            a.vec[a.nv + 2] = (double)a.counters->n_failed;
        }
    }
    if (t != 0) return;
    if (a.mode & kCtlDecide) {
        const double acc_now = a.vec[a.nv], acc_ever = a.vec[a.nv + 1], n_failed = a.vec[a.nv + 2];
        MHLogEntry &e = a.log[a.iteration - 1];
        e.accepted_now = acc_now;
        e.accepted_ever = acc_ever;
        e.n_failed = n_failed;
        e.rk_attempts = a.counters->rk_attempts;       // this rank's
        e.long_items = a.counters->long_items;
        e.solved_items = a.counters->solved_items;
        e.snap = *a.counters;
        ctl->n_done = a.iteration;
        if (acc_ever > ctl->thr_stop || n_failed != 0.0) {
            ctl->stop = 1;
            return;
        }
        if (acc_ever < ctl->thr_halve) ctl->ratio = ctl->ratio * 0.5;
    }
    if (a.mode & kCtlTransform) {
        static_assert(nv <= SMC_MAX_DIM + SMC_MAX_DIM * (SMC_MAX_DIM + 1) / 2, "");
        mh_transform_body<D>(a.mom, a.sums, a.n_global, wcov, a.shift_io, a.cov_out, a.xform_out);
        MHLogEntry &e = a.log[a.iteration];
        e.ratio = ctl->ratio;
        for (int i = 0; i < D * D; ++i) e.cov[i] = a.cov_out[i];
    }
}

// ---------------------------------------------------------------------------------------------
// residual-systematic resampling (Micmem_SMC_main.py:147-184)
// ---------------------------------------------------------------------------------------------
// Tiles of kScanTile consecutive particles; a thread owns kScanItems consecutive ones.
struct ResampleArgs {
    double max_lk, gm, sum_w;  // w_i = exp((lk_i-max_lk)*gm)/sum_w          (:124,130)
    double n_global, inv_np;   // p_is = trunc(w*N); residual = w - p_is*inv_Np (:147,150)
    double wrand, base;        // wrand (:156); base = residual sum of all lower ranks
    int first_rank;            // this rank holds global particle 0
    int scheme;                // SMC_RESAMPLE_*
    const double *thr;         // multinomial: the N sorted thresholds (order statistics of N uniforms)
    int64_t n_thr;
};

__device__ __forceinline__ void resample_item(const ResampleArgs &a, double lk, double &resid, int64_t &cnt) {
    const double w = exp((lk - a.max_lk) * a.gm) / a.sum_w;
    if (a.scheme != SMC_RESAMPLE_RESIDUAL_SYSTEMATIC) {   // no deterministic copies: the running sum is the cumulative weight
        cnt = 0;
        resid = w;
        return;
    }
    const double c = trunc(w * a.n_global);
    cnt = (int64_t)c;
    resid = w - c * a.inv_np;
}

// phase 1: per tile sums of residuals and integer parts
__global__ void __launch_bounds__(kScanBlock) resample_tile_sums_kernel(const double *__restrict__ lk, int64_t n,
                                                                        ResampleArgs a, double *__restrict__ blk_r,
                                                                        int64_t *__restrict__ blk_c) {
    __shared__ double lds_d[4];
    __shared__ int64_t lds_i[4];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    double rs = 0.0;
    int64_t cs = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < n) {
            double r;
            int64_t c;
            resample_item(a, lk[i], r, c);
            rs += r;
            cs += c;
        }
    }
    rs = block_sum(rs, lds_d);
    cs = block_sum(cs, lds_i);
    if (threadIdx.x == 0) {
        blk_r[blockIdx.x] = rs;
        blk_c[blockIdx.x] = cs;
    }
}

// exclusive scan over the tile sums by ONE block (tiles <= ~10^4): every thread takes a contiguous
// chunk.  In place: v[i] <- sum_{j<i} v[j]; v[nt] <- total (arrays hold nt+1 entries).
template <typename T>
__global__ void __launch_bounds__(kScanBlock) tile_exclusive_scan_kernel(T *__restrict__ v, int64_t nt) {
    __shared__ T lds[4];
    const int64_t chunk = (nt + kScanBlock - 1) / kScanBlock;
    const int64_t lo = (int64_t)threadIdx.x * chunk, hi = (lo + chunk < nt) ? lo + chunk : nt;
    T s = 0;
    for (int64_t i = lo; i < hi; ++i) s += v[i];
    T total;
    const T incl = block_inclusive_scan(s, lds, &total);
    T run = incl - s;
    for (int64_t i = lo; i < hi; ++i) {
        const T x = v[i];
        v[i] = run;
        run += x;
    }
    if (threadIdx.x == 0) v[nt] = total;
}

// number of systematic thresholds wrand + k/N (k >= 0) that are <= S   (:168-174)
__device__ __forceinline__ int64_t thresholds_below(double S, const ResampleArgs &a) {
    if (a.scheme == SMC_RESAMPLE_MULTINOMIAL) {   // upper_bound over the sorted thresholds
        int64_t lo = 0, hi = a.n_thr;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.thr[mid] <= S) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
    return (S >= a.wrand) ? (int64_t)floor((S - a.wrand) * a.n_global) + 1 : 0;
}

// Multinomial thresholds: the order statistics of N iid uniforms are the normalised partial sums of N+1 iid
// exponentials, U_(k) = (E_1 + .. + E_k) / (E_1 + .. + E_{N+1})  -  a scan instead of a sort, and every rank can
// produce the same array from the seed.  thr[k] = E_{k+1} on exit of the first kernel, the partial sums divided by
// the total after the second (thr[N] == 1 is not a threshold).
__global__ void __launch_bounds__(kScanBlock) mn_spacings_kernel(uint64_t seed, uint64_t stream, int64_t m,
                                                                 double *__restrict__ thr, double *__restrict__ blk) {
    __shared__ double lds[4];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < m) {
            const u32x4 r = philox_block(seed, (uint64_t)i, stream, 7);
            const double e = -log(1.0 - u01_from(r.x, r.y));   // u in [0,1) -> e in [0, inf)
            thr[i] = e;
            s += e;
        }
    }
    s = block_sum(s, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = s;
}
__global__ void __launch_bounds__(kScanBlock) mn_thresholds_kernel(int64_t m, double *__restrict__ thr,
                                                                   const double *__restrict__ blk_excl, int64_t nt) {
    __shared__ double lds[4];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    double e[kScanItems], s = 0.0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        e[k] = (i < m) ? thr[i] : 0.0;
        s += e[k];
    }
    double tile_total;
    const double incl = block_inclusive_scan(s, lds, &tile_total);
    double run = blk_excl[blockIdx.x] + (incl - s);
    const double inv_total = 1.0 / blk_excl[nt];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        run += e[k];
        if (i < m) thr[i] = run * inv_total;
    }
}

// phase 2: systematic offspring from the running residual sum, per-tile inclusive offspring scan.
// Boundary values of the running sum are defined ONCE (tile start = base + tile prefix, thread start =
// tile start + thread prefix) and used by both neighbours, so the extra offspring of a tile/rank add
// up to exactly thresholds_below(end) - thresholds_below(start).
__global__ void __launch_bounds__(kScanBlock)
resample_offspring_kernel(const double *__restrict__ lk, int64_t n, ResampleArgs a, const double *__restrict__ blk_r_excl,
                          int32_t *__restrict__ oscan, int64_t *__restrict__ blk_o) {
    __shared__ double lds_d[4];
    __shared__ int64_t lds_i[4];
    __shared__ double thread_start[kScanBlock + 1];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    double r[kScanItems];
    int64_t c[kScanItems];
    double rs = 0.0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        r[k] = 0.0;
        c[k] = 0;
        if (i < n) resample_item(a, lk[i], r[k], c[k]);
        rs += r[k];
    }
    double tile_total;
    const double incl = block_inclusive_scan(rs, lds_d, &tile_total);
    const double tile_start = a.base + blk_r_excl[blockIdx.x];
    const double tile_end = a.base + blk_r_excl[blockIdx.x + 1];
    thread_start[threadIdx.x] = tile_start + (incl - rs);  // thread 0: incl == rs, i.e. exactly tile_start
    if (threadIdx.x == 0) thread_start[kScanBlock] = tile_end;
    __syncthreads();
    double S = thread_start[threadIdx.x];
    const double S_end = thread_start[threadIdx.x + 1];
    int64_t m_prev = thresholds_below(S, a);
    if (a.first_rank && blockIdx.x == 0 && threadIdx.x == 0) m_prev = 0;  // nothing precedes particle 0
    int64_t o[kScanItems];
    int64_t os = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        o[k] = 0;
        if (i < n) {
            // the last valid item of a thread ends exactly on the next thread's start value
            // and the last particle of the rank ends on base + (this rank's residual total), the value
            // the next rank starts from
            const bool last = (k == kScanItems - 1);
            S = (i + 1 >= n) ? tile_end : (last ? S_end : S + r[k]);
            const int64_t m = thresholds_below(S, a);
            int64_t extra = m - m_prev;
            if (extra < 0) extra = 0;
            m_prev = (m > m_prev) ? m : m_prev;
            o[k] = c[k] + extra;
            os += o[k];
        }
    }
    int64_t tile_o;
    const int64_t incl_o = block_inclusive_scan(os, lds_i, &tile_o);
    int64_t run = incl_o - os;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < n) {
            run += o[k];
            oscan[i] = (int32_t)run;
        }
    }
    if (threadIdx.x == 0) blk_o[blockIdx.x] = tile_o;
}

// add the tile prefixes: oscan becomes the inclusive offspring scan over this rank's block
__global__ void __launch_bounds__(kScanBlock) resample_apply_prefix_kernel(int32_t *__restrict__ oscan, int64_t n,
                                                                           const int64_t *__restrict__ blk_o_excl) {
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    const int32_t off = (int32_t)blk_o_excl[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = base + k;
        if (i < n) oscan[i] += off;
    }
}

__global__ void offspring_from_scan_kernel(const int32_t *__restrict__ oscan, int64_t n, int64_t *__restrict__ p_is) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p_is[i] = (int64_t)oscan[i] - (i ? (int64_t)oscan[i - 1] : 0);
}

// phase 3 (:178-184): output slot -> ancestor by binary search in the inclusive scan, then gather.
// Local output slots [m_lo, m_hi) of this rank are written to dst (component c at dst + c*dst_stride
// + dst_off + (m - m_lo)); component d is lk.
__global__ void __launch_bounds__(256)
resample_gather_kernel(const int32_t *__restrict__ oscan, int64_t n, const double *__restrict__ src_theta,
                       int64_t src_stride, const double *__restrict__ src_lk, int d, int64_t m_lo, int64_t m_hi,
                       double *__restrict__ dst_theta, int64_t dst_stride, double *__restrict__ dst_lk, int64_t dst_off) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t m = m_lo + k;
    if (m >= m_hi) return;
    // first j with oscan[j] > m
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)oscan[mid] > m)
            hi = mid;
        else
            lo = mid + 1;
    }
    const int64_t j = lo;
    for (int c = 0; c < d; ++c) dst_theta[c * dst_stride + dst_off + k] = src_theta[c * src_stride + j];
    dst_lk[dst_off + k] = src_lk[j];
}

// ONE rank, no host in the loop: every output slot of the rank in one launch, the number of offspring read from device memory
// (the tile scan's total) - slots below it gather their ancestor, the others keep what the reference's persistent buffers hold
// there (:178-184; zeros in the first step).  Thread 0 also leaves (trunc(w N) total, offspring total) where ONE later read-back
// finds them: the driver looks at them after the Metropolis sweeps that follow, not before.
__global__ void __launch_bounds__(256)
resample_gather_all_kernel(const int32_t *__restrict__ oscan, int64_t n, const double *__restrict__ src_theta, int64_t src_stride,
                           const double *__restrict__ src_lk, int d, const int64_t *__restrict__ count_total,
                           const int64_t *__restrict__ offspring_total, int first_step, double *__restrict__ dst_theta,
                           int64_t dst_stride, double *__restrict__ dst_lk, int64_t *__restrict__ result /* [2] */) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = *offspring_total;
    if (m == 0) {
        result[0] = *count_total;
        result[1] = total;
    }
    if (m >= n) return;
    if (m < total) {          // first j with oscan[j] > m
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)oscan[mid] > m)
                hi = mid;
            else
                lo = mid + 1;
        }
        const int64_t j = lo < n ? lo : n - 1;      // total > n (the reference's IndexError case) is reported by the host afterwards
        for (int c = 0; c < d; ++c) dst_theta[c * dst_stride + m] = src_theta[c * src_stride + j];
        dst_lk[m] = src_lk[j];
    } else {
        for (int c = 0; c < d; ++c) dst_theta[c * dst_stride + m] = first_step ? 0.0 : src_theta[c * src_stride + m];
        dst_lk[m] = first_step ? 0.0 : src_lk[m];
    }
}

// rows the resampler did not write: what the reference's persistent p_filt / lk1 hold there
__global__ void resample_stale_rows_kernel(const double *__restrict__ src_theta, int64_t src_stride,
                                           const double *__restrict__ src_lk, int d, int64_t lo, int64_t hi,
                                           int first_step, double *__restrict__ dst_theta, int64_t dst_stride,
                                           double *__restrict__ dst_lk) {
    const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    for (int c = 0; c < d; ++c) dst_theta[c * dst_stride + i] = first_step ? 0.0 : src_theta[c * src_stride + i];
    dst_lk[i] = first_step ? 0.0 : src_lk[i];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// blocks of a grid-stride reduction: 2048 (8 per CU) for the passes that only stream, 512 (2 per CU) for the ESS pass (measured
// 128 .. 4096 with the transposing wave reduction in place: profiles/r02_ess_bench.log; more blocks mean more rows for the
// final sum, fewer leave SIMDs idle)
static inline int reduce_grid(int64_t n, int cap = 2048) {
    // A/B knob, clamped: the ESS pass writes 2 K <= 32 partial rows per block into d_partials (2048 * 64 doubles), so
    // anything above 4096 blocks would run over its end
    static const int ess_cap = [] {
        const char *e = getenv("SMC_REDUCE_BLOCKS");
        const int v = e ? atoi(e) : 512;
        return v < 1 ? 1 : (v > 4096 ? 4096 : v);
    }();
    if (cap == 512) cap = ess_cap;
    int64_t g = (n + kScanBlock - 1) / kScanBlock;
    if (g > cap) g = cap;    // grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

void launch_aos_to_soa(smc_ctx *c, const double *aos, double *soa, int64_t n, int d, int64_t stride) {
    const int64_t tot = n * d;
    if (tot <= 0) return;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, aos, soa, n, d,
                       stride);
}
void launch_soa_to_aos(smc_ctx *c, const double *soa, double *aos, int64_t n, int d, int64_t stride) {
    const int64_t tot = n * d;
    if (tot <= 0) return;
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, soa, aos, n, d,
                       stride);
}
void launch_sample_prior(smc_ctx *c, uint64_t seed, int64_t goff) {
    ParticleSet &P = c->set[SMC_SET_PRED];
    hipLaunchKernelGGL(sample_prior_kernel, dim3((unsigned)((c->n_local + 255) / 256)), dim3(256), 0, c->stream,
                       c->prior, P.theta, P.stride, c->n_local, seed, goff);
}
void launch_max(smc_ctx *c, const double *lk, int64_t n, double *d_out) {
    const int g = reduce_grid(n);
    hipLaunchKernelGGL(max_partial_kernel, dim3(g), dim3(kScanBlock), 0, c->stream, lk, n, c->d_partials);
    hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(kScanBlock), 0, c->stream, c->d_partials, g, d_out);
}
void launch_ess(smc_ctx *c, const double *lk, int64_t n, double max_lk, const double *gm, int k, double *d_out,
                const double *max_lk_dev) {
    EssCand cand{};
    cand.k = k;
    for (int i = 0; i < SMC_MAX_ESS_CAND; ++i) cand.gm[i] = (i < k) ? gm[i] : 0.0;
    const int g = reduce_grid(n, 512);
    int K;
    if (k <= 1) {
        K = 1;
        hipLaunchKernelGGL((ess_partial_kernel<1>), dim3(g), dim3(kScanBlock), 0, c->stream, lk, n, max_lk, max_lk_dev, cand,
                           c->d_partials);
    } else if (k <= 4) {
        K = 4;
        hipLaunchKernelGGL((ess_partial_kernel<4>), dim3(g), dim3(kScanBlock), 0, c->stream, lk, n, max_lk, max_lk_dev, cand,
                           c->d_partials);
    } else if (k <= 8) {
        K = 8;
        hipLaunchKernelGGL((ess_partial_kernel<8>), dim3(g), dim3(kScanBlock), 0, c->stream, lk, n, max_lk, max_lk_dev, cand,
                           c->d_partials);
    } else {
        K = 16;
        hipLaunchKernelGGL((ess_partial_kernel<16>), dim3(g), dim3(kScanBlock), 0, c->stream, lk, n, max_lk, max_lk_dev, cand,
                           c->d_partials);
    }
    // a second launch for the 2K sums over the blocks' rows: letting the last block to finish do it (ticket + __threadfence)
    // was tried and is 30 us SLOWER per pass - the fence writes back and invalidates the XCD's L2 in every block
    hipLaunchKernelGGL(sum_rows_final_kernel, dim3(2 * K), dim3(kScanBlock), 0, c->stream, c->d_partials, g, 2 * K, d_out);
}
int ess_padded_k(int k) { return k <= 1 ? 1 : k <= 4 ? 4 : k <= 8 ? 8 : 16; }

void launch_moment_sums(smc_ctx *c, double *d_out) {
    ParticleSet &F = c->set[SMC_SET_FILT];
    const int g = reduce_grid(c->n_local);
    hipLaunchKernelGGL(moment_sum_kernel, dim3(g), dim3(kScanBlock), 0, c->stream, F.theta, F.stride, c->n_local,
                       c->dim, c->d_partials);
    hipLaunchKernelGGL(sum_rows_final_kernel, dim3(c->dim), dim3(kScanBlock), 0, c->stream, c->d_partials, g, c->dim, d_out);
}
void launch_moment_centered(smc_ctx *c, const double *mean, double *d_out) {
    ParticleSet &F = c->set[SMC_SET_FILT];
    MeanArg m{};
    for (int i = 0; i < c->dim; ++i) m.m[i] = mean[i];
    const int g = reduce_grid(c->n_local);
    const int npair = c->dim * (c->dim + 1) / 2;
    hipLaunchKernelGGL(moment_centered_kernel, dim3(g), dim3(kScanBlock), 0, c->stream, F.theta, F.stride, c->n_local,
                       c->dim, m, (const double *)nullptr, 1.0, c->d_partials);
    hipLaunchKernelGGL(sum_rows_final_kernel, dim3(npair), dim3(kScanBlock), 0, c->stream, c->d_partials, g, npair, d_out);
}
void launch_moment_centered_dev(smc_ctx *c, const double *d_sums, double *d_out) {
    ParticleSet &F = c->set[SMC_SET_FILT];
    MeanArg m{};
    const int g = reduce_grid(c->n_local);
    const int npair = c->dim * (c->dim + 1) / 2;
    hipLaunchKernelGGL(moment_centered_kernel, dim3(g), dim3(kScanBlock), 0, c->stream, F.theta, F.stride, c->n_local,
                       c->dim, m, d_sums, (double)c->n_global, c->d_partials);
    hipLaunchKernelGGL(sum_rows_final_kernel, dim3(npair), dim3(kScanBlock), 0, c->stream, c->d_partials, g, npair, d_out);
}
void launch_mh_transform(smc_ctx *c, const double *d_mom, const double *d_sums, const double *w_cov, double *d_shift,
                         double *d_cov, double *d_xform) {
    WCov w{};
    for (int i = 0; i < c->dim * c->dim; ++i) w.w[i] = w_cov[i];
    const double ng = (double)c->n_global;
#define SMC_XF(D) case D: hipLaunchKernelGGL((mh_transform_kernel<D>), dim3(1), dim3(64), 0, c->stream, d_mom, d_sums, ng, w, \
                                             d_shift, d_cov, d_xform); break
    switch (c->dim) {
        SMC_XF(1); SMC_XF(2); SMC_XF(3); SMC_XF(4); SMC_XF(5); SMC_XF(6); SMC_XF(7); SMC_XF(8);
    }
#undef SMC_XF
    static_assert(SMC_MAX_DIM == 8, "one instantiation of mh_transform_kernel per dimension");
}
void launch_mh_control(smc_ctx *c, const MHControlArgs &a, const double *w_cov) {
    WCov w{};
    for (int i = 0; i < c->dim * c->dim; ++i) w.w[i] = w_cov[i];
#define SMC_CT(D) case D: hipLaunchKernelGGL((mh_control_kernel<D>), dim3(1), dim3(kScanBlock), 0, c->stream, a, w); break
    switch (c->dim) {
        SMC_CT(1); SMC_CT(2); SMC_CT(3); SMC_CT(4); SMC_CT(5); SMC_CT(6); SMC_CT(7); SMC_CT(8);
    }
#undef SMC_CT
}
// per-block rows of nv doubles (accept kernel) -> out[0..nv), then the sweep's accept counters as doubles (exact below 2^53):
// out[nv] = accepted_now, out[nv+1] = accepted_ever, out[nv+2] = n_failed - ONE vector for ONE all-reduce per iteration
// ctl (batch of iterations under device control): after the loop has ended nothing new was accumulated - but the all-reduce
// that follows on the stream still runs on every rank, so rank 0 keeps the vector of the last iteration that ran and the
// other ranks contribute zeros: the sum leaves it (and with it the carried moments) unchanged.
__global__ void __launch_bounds__(kScanBlock) moments_reduce_kernel(const double *__restrict__ rows, int n_rows, int nv,
                                                                    const SweepCounters *__restrict__ counters,
                                                                    double *__restrict__ out, const MHControl *__restrict__ ctl,
                                                                    int rank) {
    __shared__ double lds[4];
    const int v = blockIdx.x;          // one block per value, the last block carries the counters
    if (ctl && ctl->stop) {
        if (rank != 0 && threadIdx.x == 0) {
            if (v < nv) out[v] = 0.0;
            else out[nv] = out[nv + 1] = out[nv + 2] = 0.0;
        }
        return;
    }
    if (v < nv) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_rows; i += blockDim.x) s += rows[(size_t)i * nv + v];
        s = block_sum(s, lds);
        if (threadIdx.x == 0) out[v] = s;
    } else if (threadIdx.x == 0) {
        out[nv] = (double)counters->accepted_now;
        out[nv + 1] = (double)counters->accepted_ever;
        out[nv + 2] = (double)counters->n_failed;
    }
}
void launch_moments_reduce(smc_ctx *c, int n_rows, int nv, double *d_out, const MHControl *ctl) {
    hipLaunchKernelGGL(moments_reduce_kernel, dim3(nv + 1), dim3(kScanBlock), 0, c->stream, c->d_partials, n_rows, nv, c->d_counters,
                       d_out, ctl, c->rank);
}

static ResampleArgs make_args(smc_ctx *c, double max_lk, double gm, double sum_w, double wrand, double base) {
    ResampleArgs a{};
    a.max_lk = max_lk;
    a.gm = gm;
    a.sum_w = sum_w;
    a.n_global = (double)c->n_global;
    a.inv_np = 1.0 / (double)c->n_global;  // inv_Np = 1 / n_particle (Micmem_settings.py:17)
    a.wrand = wrand;
    a.base = base;
    a.first_rank = (c->rank == 0);
    a.scheme = c->resampling;
    a.thr = c->d_mn_thr;
    a.n_thr = c->n_global;
    return a;
}
void launch_resample_phase1(smc_ctx *c, double max_lk, double gm, double sum_w) {
    ParticleSet &P = c->set[SMC_SET_PRED];
    const ResampleArgs a = make_args(c, max_lk, gm, sum_w, 0.0, 0.0);
    const int64_t nt = c->n_tiles;
    hipLaunchKernelGGL(resample_tile_sums_kernel, dim3((unsigned)nt), dim3(kScanBlock), 0, c->stream, P.lk, c->n_local, a,
                       c->d_blk_r, c->d_blk_c);
    hipLaunchKernelGGL((tile_exclusive_scan_kernel<double>), dim3(1), dim3(kScanBlock), 0, c->stream, c->d_blk_r, nt);
    hipLaunchKernelGGL((tile_exclusive_scan_kernel<int64_t>), dim3(1), dim3(kScanBlock), 0, c->stream, c->d_blk_c, nt);
}
void launch_resample_phase2(smc_ctx *c, double max_lk, double gm, double sum_w, double base, double wrand) {
    ParticleSet &P = c->set[SMC_SET_PRED];
    const ResampleArgs a = make_args(c, max_lk, gm, sum_w, wrand, base);
    const int64_t nt = c->n_tiles;
    if (c->resampling == SMC_RESAMPLE_MULTINOMIAL) {   // thresholds of this resampling: seeded by the bits of wrand
        const int64_t m = c->n_global + 1, ntm = (m + kScanTile - 1) / kScanTile;
        uint64_t seed;
        memcpy(&seed, &wrand, sizeof seed);
        hipLaunchKernelGGL(mn_spacings_kernel, dim3((unsigned)ntm), dim3(kScanBlock), 0, c->stream, seed,
                           0x5EEDull << 32, m, c->d_mn_thr, c->d_mn_blk);
        hipLaunchKernelGGL((tile_exclusive_scan_kernel<double>), dim3(1), dim3(kScanBlock), 0, c->stream, c->d_mn_blk, ntm);
        hipLaunchKernelGGL(mn_thresholds_kernel, dim3((unsigned)ntm), dim3(kScanBlock), 0, c->stream, m, c->d_mn_thr,
                           c->d_mn_blk, ntm);
    }
    // the per-tile offspring totals have their own array: d_blk_c[n_tiles] keeps the sum of trunc(w N) of phase 1
    hipLaunchKernelGGL(resample_offspring_kernel, dim3((unsigned)nt), dim3(kScanBlock), 0, c->stream, P.lk, c->n_local,
                       a, c->d_blk_r, c->d_oscan, c->d_blk_o);
    hipLaunchKernelGGL((tile_exclusive_scan_kernel<int64_t>), dim3(1), dim3(kScanBlock), 0, c->stream, c->d_blk_o, nt);
    hipLaunchKernelGGL(resample_apply_prefix_kernel, dim3((unsigned)nt), dim3(kScanBlock), 0, c->stream, c->d_oscan,
                       c->n_local, c->d_blk_o);
}
void launch_resample_gather_all(smc_ctx *c, int first_step, int64_t *d_result) {
    ParticleSet &P = c->set[SMC_SET_PRED];
    ParticleSet &F = c->set[SMC_SET_FILT];
    hipLaunchKernelGGL(resample_gather_all_kernel, dim3((unsigned)((c->n_local + 255) / 256)), dim3(256), 0, c->stream, c->d_oscan,
                       c->n_local, P.theta, P.stride, P.lk, c->dim, c->d_blk_c + c->n_tiles, c->d_blk_o + c->n_tiles, first_step,
                       F.theta, F.stride, F.lk, d_result);
}
void launch_offspring_from_scan(smc_ctx *c, int64_t *d_out) {
    hipLaunchKernelGGL(offspring_from_scan_kernel, dim3((unsigned)((c->n_local + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_oscan, c->n_local, d_out);
}
void launch_resample_gather(smc_ctx *c, int64_t m_lo, int64_t m_hi, double *dst_theta, int64_t dst_stride,
                            double *dst_lk, int64_t dst_off) {
    const int64_t cnt = m_hi - m_lo;
    if (cnt <= 0) return;
    ParticleSet &P = c->set[SMC_SET_PRED];
    hipLaunchKernelGGL(resample_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->d_oscan,
                       c->n_local, P.theta, P.stride, P.lk, c->dim, m_lo, m_hi, dst_theta, dst_stride, dst_lk, dst_off);
}
void launch_resample_stale(smc_ctx *c, int64_t lo, int64_t hi, int first_step) {
    if (hi <= lo) return;
    ParticleSet &P = c->set[SMC_SET_PRED];
    ParticleSet &F = c->set[SMC_SET_FILT];
    hipLaunchKernelGGL(resample_stale_rows_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, c->stream,
                       P.theta, P.stride, P.lk, c->dim, lo, hi, first_step, F.theta, F.stride, F.lk);
}

}  // namespace smc
