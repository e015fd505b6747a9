// exchange_plan.h -- who sends which offspring to whom after a sharded resampling (host arithmetic only, no device call).
//
// After phase 2 every rank knows offspring_all[q], the number of offspring rank q's particles produce, and out_base_all[q],
// its exclusive prefix: rank q's offspring occupy the GLOBAL output slots [out_base_all[q], out_base_all[q] +
// offspring_all[q]) (ancestor order, Micmem_SMC_main.py:178-184), and rank r owns the slots [r * n_local, (r + 1) * n_local).
// The plan of rank R is the intersection of those two partitions seen from R:
//   own      slots of its own offspring that fall into its own range: gathered straight into p_filt / lk1;
//   send[q]  slots of its offspring that fall into rank q's range: gathered into the send staging at send_off[q], count
//            send_cnt[q], as ONE contiguous [component][count] block per peer;
//   recv[q]  slots of rank q's offspring that fall into R's range: received into the receive staging at recv_off[q] and
//            spread over the rows [recv_row[q], recv_row[q] + recv_cnt[q]) of p_filt;
//   stale    rows nobody writes because fewer than N offspring were produced: [stale_lo, n_local).
// smc_resample_phase3 executes this plan (rccl_exchange_blocks); smc_exchange_plan exposes it so that sender and receiver
// plans can be checked against each other for any world size on a machine without GPUs (tests/test_exchange_plan.py) - the
// part of the exchange that only several real ranks would otherwise exercise.
#pragma once
#include <stdint.h>

#include <vector>

namespace smc {

struct ExchangePlan {
    std::vector<int64_t> send_off, send_cnt;   // per peer; offsets in particles into the send staging; [R] stays 0
    std::vector<int64_t> src_lo;               // per peer (incl. R): first local offspring index (slot - out_base[R]) of the block
    std::vector<int64_t> recv_off, recv_cnt, recv_row;
    int64_t own_src_lo = 0, own_cnt = 0, own_row = 0;
    int64_t send_total = 0, recv_total = 0, stale_lo = 0, total = 0;
    bool prefix_ok = true;
};

inline ExchangePlan make_exchange_plan(int W, int R, int64_t nl, const int64_t *out_base_all, const int64_t *offspring_all) {
    auto imax = [](int64_t a, int64_t b) { return a > b ? a : b; };
    auto imin = [](int64_t a, int64_t b) { return a < b ? a : b; };
    ExchangePlan p;
    p.send_off.assign(W, 0); p.send_cnt.assign(W, 0); p.src_lo.assign(W, 0);
    p.recv_off.assign(W, 0); p.recv_cnt.assign(W, 0); p.recv_row.assign(W, 0);
    for (int q = 0; q < W; ++q) {
        if (out_base_all[q] != p.total || offspring_all[q] < 0) p.prefix_ok = false;
        p.total += offspring_all[q];
    }
    const int64_t my_base = out_base_all[R], my_cnt = offspring_all[R];
    for (int q = 0; q < W; ++q) {
        // my offspring that land in rank q's slot range
        int64_t lo = imax(my_base, q * nl), hi = imin(my_base + my_cnt, (q + 1) * nl);
        if (hi > lo) {
            p.src_lo[q] = lo - my_base;
            if (q == R) {
                p.own_src_lo = lo - my_base;
                p.own_cnt = hi - lo;
                p.own_row = lo - R * nl;
            } else {
                p.send_off[q] = p.send_total;
                p.send_cnt[q] = hi - lo;
                p.send_total += hi - lo;
            }
        }
        if (q == R) continue;
        // rank q's offspring that land in my slot range
        lo = imax(out_base_all[q], R * nl);
        hi = imin(out_base_all[q] + offspring_all[q], (R + 1) * nl);
        if (hi > lo) {
            p.recv_off[q] = p.recv_total;
            p.recv_cnt[q] = hi - lo;
            p.recv_row[q] = lo - R * nl;
            p.recv_total += hi - lo;
        }
    }
    p.stale_lo = imin(imax(p.total - R * nl, 0), nl);
    return p;
}

}  // namespace smc
