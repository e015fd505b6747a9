// stage_kernels.h -- launchers of stage_kernels.hip (internal)
#pragma once
#include "smc_internal.h"

namespace smc {
void launch_aos_to_soa(smc_ctx *c, const double *aos, double *soa, int64_t n, int d, int64_t stride);
void launch_soa_to_aos(smc_ctx *c, const double *soa, double *aos, int64_t n, int d, int64_t stride);
void launch_sample_prior(smc_ctx *c, uint64_t seed, int64_t goff);
void launch_max(smc_ctx *c, const double *lk, int64_t n, double *d_out);
// max_lk_dev != nullptr: max(lk) is read from device memory (left there by launch_max / an all-reduce earlier on the stream)
void launch_ess(smc_ctx *c, const double *lk, int64_t n, double max_lk, const double *gm, int k, double *d_out,
                const double *max_lk_dev = nullptr);
int ess_padded_k(int k);
void launch_moment_sums(smc_ctx *c, double *d_out);
void launch_moment_centered(smc_ctx *c, const double *mean, double *d_out);
void launch_moment_centered_dev(smc_ctx *c, const double *d_sums, double *d_out);   // mean = d_sums / n_global on the device
void launch_mh_transform(smc_ctx *c, const double *d_mom, const double *d_sums, const double *w_cov, double *d_shift,
                         double *d_cov, double *d_xform);
void launch_moments_reduce(smc_ctx *c, int n_rows, int nv, double *d_out, const MHControl *ctl = nullptr);
void launch_mh_control(smc_ctx *c, const MHControlArgs &a, const double *w_cov);   // one block: decide / transform (stage_kernels.hip)
void launch_resample_phase1(smc_ctx *c, double max_lk, double gm, double sum_w);
void launch_resample_phase2(smc_ctx *c, double max_lk, double gm, double sum_w, double base, double wrand);
void launch_offspring_from_scan(smc_ctx *c, int64_t *d_out);
void launch_resample_gather(smc_ctx *c, int64_t m_lo, int64_t m_hi, double *dst_theta, int64_t dst_stride,
                            double *dst_lk, int64_t dst_off);
void launch_resample_stale(smc_ctx *c, int64_t lo, int64_t hi, int first_step);
void launch_resample_gather_all(smc_ctx *c, int first_step, int64_t *d_result);   // one rank: all slots, totals read on the device
}  // namespace smc
