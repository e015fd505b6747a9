// solve_sched.h -- the persistent, lane-level work scheduler of the likelihood sweeps, shared by the built-in
// Michaelis-Menten kernel (mm_kernels.hip: MMOps) and the run-time compiled user-model kernel (user_model.hip: UserOps; this
// file is handed to hiprtc as an in-memory header).  gfx950 (MI355X) only.
//
// What it schedules: the (particle, experiment) solves of one sweep (sim_particle, Micmem_likelihood.py:79-92) - n particles
// x n_ex experiments, each an adaptive RK45 integration whose number of step attempts varies 100-fold between particles
// over the prior.  The reference fans them out as one Ray task per particle (:83-87); here
//   * every wave is persistent and takes CHUNKS of item indices from one global counter (one atomic per kChunk items);
//   * items are STARTED (initial state, first derivative, select_initial_step) many at a time into a per-wave POOL in LDS,
//     by lanes 0 .. take-1 whatever those lanes are running, and a lane that finishes takes the oldest started item - so the
//     expensive start-up code runs at >= 75 % of the lanes and no lane waits for a hand-out;
//   * between hand-outs the live lanes run attempt after attempt in a tight loop (one ballot per attempt);
//   * optionally a LIST of predictably long items is handed out before the index-ordered ones (Ops::listed);
//   * when the queue is empty the TAIL begins: a wave left with exactly one live item broadcasts its state through v_readlane
//     and runs it on wave-uniform operands (scalar branches, no per-lane selects); during the tail a Metropolis sweep asks
//     every kRejectCheckEvery attempts whether the item's proposal is already certain to be rejected (Ops::certainly_rejected)
//     and stops it if so.
// All loop control is wave-uniform BY CONSTRUCTION (chunk bounds through v_readfirstlane, lane counts through ballots, the
// rejection verdict of the uniform tail through v_readfirstlane): tests/test_k8_uniform_control.py asks the compiler's own
// uniformity analysis that no loop with a divergent exit contains a cross-lane operation.  Every lane leaves when the counter
// is exhausted and its item is done (Ops::attempt must terminate every item after a bounded number of attempts), so the grid
// always drains.
//
// Ops (all members __device__ __forceinline__; the scheduler never looks inside an item):
//   typename Item                        per-lane state of a started item, including where its result goes
//   static constexpr int kPoolWords      8-byte words pack() writes
//   long long n; int n_ex;               particles, experiments; item order ((P * n_blk64 + blk) * kExPerChunk + j) * 64 + lane with experiment
//                                        e = P * kExPerChunk + j: a chunk is ONE block of 64 particles x kExPerChunk experiments (see kExPerChunk)
//   const int *list; unsigned n_list;    particles handed out first, or nullptr / 0
//   const int *solo; unsigned n_solo;    particles solo[0], solo[-1], ... whose solves run one per wave on uniform operands
//                                        from their first attempt on (the stiffest of the list), or nullptr / 0
//   int start(long long p, int e, bool from_list, Item &nb)
//                                        kStartStarted: nb needs attempts; kStartDone: the item is finished and published
//                                        (nothing to integrate, masked proposal, ...); kStartSkipped: index-ordered pass met a
//                                        particle of the list (from_list == false) - it has been handed out already
//   void pack(const Item &, double *slot)   / void unpack(Item &, const double *slot)     slot[w * 64], w < kPoolWords
//   int  attempt(Item &)                 one step attempt; 0 = still running, anything else = finished / failed
//   int  uniform_attempts(Item &, int n) up to n attempts of an item whose state is wave-uniform, stopping at the first that
//                                        does not return 0; returns the last status (uniform_attempts_plain() below, or better)
//   void finish(Item &, int status)      publish the result
//   Item broadcast(const Item &, int src)   the item of lane src in every lane (v_readlane on every field)
//   int patience                         attempts a wave waits for ALL its lanes before a hand-out (0: none; see kRefillAt below)
//   bool long_running(const Item &)      the item has already run long enough to count as a straggler: a wave that holds one does
//                                        not wait (per-lane)
//   long long positions()                positions of the index-ordered pass: n, or fewer when the sweep's order leaves out
//                                        particles that need no solve (wave-uniform)
//   int start_at(long long pos, int e, Item &nb)
//                                        start() for position pos of the index-ordered pass: particle pos, or what the sweep's
//                                        cost order puts there (like particles share a wave and stay in phase)
//   bool reject_enabled()                wave-uniform
//   bool certainly_rejected(const Item &)   exact bound; may read results other waves have published
//   void cancel(Item &)                  publish "stopped: its proposal is rejected"
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#ifndef SMC_CHUNK
#define SMC_CHUNK 128
#endif
#ifndef SMC_REFILL_AT
#define SMC_REFILL_AT 24
#endif
#ifndef SMC_POOL_FREE
#define SMC_POOL_FREE 48
#endif
#ifndef SMC_STIFF_PER_CHUNK
#define SMC_STIFF_PER_CHUNK 16
#endif
#ifdef SMC_ISA_MARKS   // analysis builds only (tools/isa_blocks.py): names the loops in the `hipcc -S` listing
#define SMC_ISA_MARK(name) asm volatile("; MARK " name)
#else
#define SMC_ISA_MARK(name)
#endif

namespace smc {

constexpr int kSchedWave = 64;                // gfx950 wavefront
constexpr int kChunk = SMC_CHUNK;             // item indices per global dequeue (2 per lane)
static_assert(kChunk % 64 == 0, "a chunk is a whole number of 64-item groups");
// Experiments a chunk covers.  Rounds 1-4 ran the index-ordered pass EXPERIMENT-MAJOR (item = (e * n_blk + blk) * 64 + lane): the
// kChunk / 64 groups of a chunk were consecutive blocks of one experiment, and a particle's parameters (24 B, or the 32-byte
// record of a cost-ordered sweep) were fetched from HBM once per experiment - six passes over the population by waves of
// different XCDs, far apart in time, no cache holds them: 135 MB fetched per 10^6-particle launch where 25 MB are algorithmic
// (profiles/r05_pmc_fetch_write_summary.json).  Now the groups of a chunk are the SAME block under consecutive experiments: the
// wave that starts experiment e of a block starts e + 1 of it a few microseconds later and finds the parameters in its CU's /
// XCD's cache.  A group is still one experiment (its 64 items run in phase); which items share a chunk cannot change a result.
// SMC_EX_PER_CHUNK=1 builds the old order (A/B).
#ifndef SMC_EX_PER_CHUNK
#define SMC_EX_PER_CHUNK (SMC_CHUNK / 64)
#endif
constexpr int kExPerChunk = SMC_EX_PER_CHUNK;
static_assert(kExPerChunk == 1 || kExPerChunk == kChunk / 64, "a chunk is one block x kExPerChunk experiments, or blocks of one experiment");
constexpr int kRefillAt = SMC_REFILL_AT;      // idle lanes that make a wave look at its pool (and refill it when it is empty)
constexpr int kPoolRefillFree = SMC_POOL_FREE;   // free pool slots that trigger the start of that many new items
// Entries of the list a wave takes with one dequeue.  Fewer than a full wave on purpose: the long solves spread over many
// waves (12 000 list items of a 10^6-particle prior sweep -> 750 waves), each of which fills its other lanes with ordinary
// items, so that at the end of the sweep a wave rarely holds two survivors and can run its last one on wave-uniform operands.
constexpr int kStiffPerChunk = SMC_STIFF_PER_CHUNK;
static_assert(kStiffPerChunk >= 1 && kStiffPerChunk <= 64, "a list chunk is started by one wave at once");
// Staying IN PHASE (Ops::patience).  In the posterior phase the 64 items a wave starts together (consecutive particles, one
// experiment) take the same number of attempts to within a few.  If the wave hands new items to the first kRefillAt
// finishers at once, those lanes run out of phase with the rest from then on - and the dense-output loop of EVERY attempt
// then runs as long as the lane that happens to be in the late, large-step part of its trajectory needs (5 outputs per step)
// although the average lane needs 2: 3.1 wave-iterations per attempt where the lanes' mean is 2.2 (SQ counters, DESIGN.md
// 4.1 item 8).  Letting the whole wave restart together costs the early finishers a few attempts of waiting and buys that
// back: steady-state sweep 1.33 -> 1.21 ms with a patience of 4 attempts, 1.17 ms with 16 (profiles/r03_ab_patience.log).
// Waiting idles the wave behind every straggler of a prior-like population, though (whole run 93 -> 95 / 102 ms when it is
// unconditional), and the finishers of a posterior-like wave do not arrive sharply enough for the wave to tell the two
// situations apart by itself (tried: "wait while >= 8 more lanes finished in the last attempt", "wait if 24 lanes finished
// within 4 attempts of the first": no gain).  So the HOST decides per sweep, from the device-counted number of long items
// of the previous sweep, and passes the patience in: 0 = hand out as soon as kRefillAt lanes are idle.
constexpr int kRejectCheckEvery = 512;        // attempts between two looks at the rejection bound (a look costs about five)
constexpr int kStartStarted = 0, kStartDone = 1, kStartSkipped = 2;

// value of lane `src` (wave-uniform index) in every lane, as a scalar
__device__ __forceinline__ double lane_value(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long lane_value_ll(long long v, int src) {
    return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), src) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src));
}

// Ops::uniform_attempts for a model without a hand-written lone-chain loop
template <class Ops>
__device__ __forceinline__ int uniform_attempts_plain(const Ops &ops, typename Ops::Item &u, int budget) {
    int st;
    do {
        SMC_ISA_MARK("uniform_tail_attempt");
        st = ops.attempt(u);
    } while (st == 0 && --budget > 0);
    return st;
}

// One item whose state is the same in every lane (wave-uniform), to the end: attempts in a tight loop of scalar branches,
// and - in a Metropolis sweep - a look at the exact rejection bound before the first attempt and every kRejectCheckEvery.
template <class Ops>
__device__ __forceinline__ void run_item_uniform(Ops &ops, typename Ops::Item &u, bool reject) {
    // The look at the bound sits in an outer loop so that the attempt loop itself stays the bare serial chain: with the
    // check inside it the compiler kept the check's operands live across every attempt and reloaded spilled SGPRs in the
    // loop (0.51 instead of 0.41 us per attempt, tools/tail_latency.py).
    int st = 0;
    bool cancelled = false;
    for (;;) {
        // Every lane evaluates the same bound on the same operands, but its atomic loads are a source of divergence for
        // the compiler: without the v_readfirstlane it wraps the WHOLE attempt loop below in exec-mask control flow
        // (s_and_saveexec per branch, per-lane selects) instead of scalar branches - that is how the uniform tail lost a
        // tenth of a microsecond per attempt when early rejection went in (tools/isa_blocks.py on the listing: 11
        // saveexec / 0 s_cbranch_vcc with the bare call, 0 / 8 with the broadcast).
        if (reject && __builtin_amdgcn_readfirstlane((int)ops.certainly_rejected(u))) {
            cancelled = true;
            break;
        }
        st = ops.uniform_attempts(u, kRejectCheckEvery);
        if (st != 0) break;
    }
    // Every lane holds the same result and stores it to the same address: one wave-wide store of 64 identical values
    // instead of an `if (lane == src)` - a divergent branch whose join would be the enclosing loop's exit block, which is
    // exactly what makes the compiler's uniformity analysis call that whole loop, with its ballots and v_readlanes, a cycle
    // with a divergent exit.
    if (cancelled)
        ops.cancel(u);
    else
        ops.finish(u, st);
}

// s_pool: this wave's kPoolWords * 64 doubles of LDS.  queue: the global item counter, zero at launch.
template <class Ops>
__device__ __forceinline__ void solve_persistent(Ops &ops, unsigned long long *queue, double *s_pool) {
    using Item = typename Ops::Item;
    constexpr int kWave = kSchedWave;
    const int n_ex = ops.n_ex;
    // Queue space: first the list, n_ex passes over it in chunks of kChunk indices of which the first kStiffPerChunk are
    // list entries (so that the chunk arithmetic stays in units of kChunk), then the index-ordered items in groups of 64
    // particles x n_ex experiments; the last group may be partial.
    const long long n_pos = ops.positions();
    const unsigned long long n_blk = (unsigned long long)((n_pos + kWave - 1) / kWave);
    // groups of 64 items in queue order: kExPerChunk == 1: (e, blk), experiment-major; otherwise (P, blk, j) with e = P * kExPerChunk + j
    // (n_ex need not be a multiple of kExPerChunk: the groups of the experiments beyond n_ex - 1 start nothing)
    const unsigned n_ex_padded = (unsigned)((n_ex + kExPerChunk - 1) / kExPerChunk) * kExPerChunk;
    const unsigned n_list = ops.n_list;
    const unsigned list_cpe = (n_list + kStiffPerChunk - 1) / kStiffPerChunk;          // chunks per experiment
    const unsigned long long q_list_end = (unsigned long long)list_cpe * n_ex * kChunk;
    const unsigned long long n_items = q_list_end + n_blk * kWave * n_ex_padded;
    const int lane = threadIdx.x & (kWave - 1);

    // ---- solo phase: the stiffest solves of the sweep, one per wave, on wave-uniform operands from their first attempt.
    // Static round-robin over the waves of the grid (no atomic): item s = e * n_solo + j is experiment e of solo[-j].
    {
        const unsigned n_solo = ops.n_solo;
        const unsigned n_solo_items = n_solo * (unsigned)n_ex;
        const unsigned wave_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
        const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
        const bool reject = ops.reject_enabled();
        for (unsigned s = wave_id; s < n_solo_items; s += n_waves) {   // at most one per wave when the caller caps n_solo at waves / n_ex
            const int e = (int)(s / n_solo);
            const long long p = ops.solo[-(long long)(s - (unsigned)e * n_solo)];
            Item u;
            if (ops.start(p, e, true, u) == kStartStarted) run_item_uniform(ops, u, reject);   // every lane: the same item
        }
    }

    Item it;
    bool live = false;              // this lane holds a running item
    // Loop control lives in SGPRs (the chunk bounds come back from the atomic through v_readfirstlane, n_idle from a
    // ballot): every branch of the scheduling logic is a scalar branch.  In round 1 q_lo / q_hi travelled through a
    // __shfl, the compiler had to treat `q_lo == q_hi` and `drained` as divergent and wrapped the whole loop in exec-mask
    // bookkeeping: 0.19 us per iteration on top of the 0.41 us of an attempt for a wave that runs alone (the stragglers of
    // the early tempering steps), measured with tools/tail_latency.py.
    unsigned long long q_lo = 0, q_hi = 0;  // the wave's current chunk
    unsigned long long q_blk = 0;           // regular chunk: block of 64 particles of the 64-item group of its first item ...
    unsigned long long q_grp = 0;           // ... that group
    int q_e = 0;                            // ... and experiment (one division per chunk, none per hand-out)
    unsigned q_list = 0;                    // list chunk: list index of its first entry
    bool q_is_list = false;
    bool drained = false;                   // the global queue is empty

    int pool_head = 0, pool_count = 0;      // the ring of started items: slots [head, head + count) mod 64 (wave-uniform)

    for (;;) {
        // ---- start new items into the pool ------------------------------------------------------------------
        if (!drained && kWave - pool_count >= kPoolRefillFree) {
            if (q_lo == q_hi) {  // next chunk: one atomic per wave and kChunk indices
                // The first lane adds kChunk, every other lane adds 0, and the first lane's return value is the start of
                // the chunk: correct whether the compiler's atomic optimiser folds the 64 lane atomics into one (it does)
                // or not - round 2's form (every lane adds kChunk / 64) was correct only with it.  And no `if (lane == 0)`
                // around the atomic: the compiler may split such a branch from the v_readfirstlane that follows it and
                // let the other lanes run ahead with b = 0 (profiles/r02_k8_dequeue_hang_isa.md).
                const unsigned long long b = atomicAdd(queue, lane == 0 ? (unsigned long long)kChunk : 0ull);
                const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
                const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
                q_lo = ((unsigned long long)b_hi << 32) | b_lo;
                if (q_lo >= n_items) {
                    drained = true;
                    q_lo = q_hi = 0;
                } else if (q_lo < q_list_end) {
                    q_is_list = true;
                    const unsigned c = (unsigned)(q_lo / kChunk);
                    q_e = (int)(c / list_cpe);
                    q_list = (c - (unsigned)q_e * list_cpe) * kStiffPerChunk;
                    const unsigned left = n_list - q_list;
                    q_hi = q_lo + (left < (unsigned)kStiffPerChunk ? left : (unsigned)kStiffPerChunk);
                } else {
                    q_is_list = false;
                    q_hi = (q_lo + kChunk < n_items) ? q_lo + kChunk : n_items;
                    q_grp = (q_lo - q_list_end) >> 6;
                    if (kExPerChunk == 1) {
                        q_e = (int)(q_grp / n_blk);          // experiment-major
                        q_blk = q_grp - (unsigned long long)q_e * n_blk;
                    } else {                                 // chunks are kChunk-aligned: q_grp is a multiple of kExPerChunk
                        const unsigned long long P = q_grp / (n_blk * kExPerChunk);
                        q_e = (int)P * kExPerChunk;          // first experiment of the chunk's block
                        q_blk = (q_grp - P * n_blk * kExPerChunk) / kExPerChunk;
                    }
                }
            }
            if (!drained) {
                const unsigned long long avail = q_hi - q_lo;
                const int free_slots = kWave - pool_count;
                const int take = (unsigned long long)free_slots < avail ? free_slots : (int)avail;
                bool started = false;               // this lane's new item needs attempts: it goes into the pool
                Item nb;
                if (lane < take) {
                    int e = q_e;
                    if (q_is_list) {
                        started = ops.start(ops.list[q_list + (unsigned)(q_lo % kChunk) + (unsigned)lane], e, true, nb) == kStartStarted;
                    } else {
                        const unsigned long long item = q_lo - q_list_end + lane;
                        unsigned long long blk;
                        if (kExPerChunk == 1) {
                            // 64-item group = (experiment, block of 64 particles); a chunk spans kChunk / 64 + 1 groups at most
                            blk = q_blk + ((item >> 6) - q_grp);
#pragma unroll
                            for (int w = 0; w < kChunk / 64 + 1; ++w)
                                if (blk >= n_blk) { blk -= n_blk; ++e; }
                        } else {                             // the chunk's block under experiment q_e + (group within the chunk)
                            blk = q_blk;
                            e += (int)((item >> 6) - q_grp);
                        }
                        const long long pos = (long long)blk * kWave + (long long)(item & 63);
                        if (pos < n_pos && e < n_ex) started = ops.start_at(pos, e, nb) == kStartStarted;
                    }
                }
                const unsigned long long started_mask = __ballot(started);
                if (started) {
                    const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(started_mask >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)started_mask, 0u));
                    ops.pack(nb, s_pool + ((pool_head + pool_count + r) & (kWave - 1)));
                }
                pool_count += __popcll(started_mask);
                q_lo += take;
            }
        }
        // ---- idle lanes take started items from the pool (LDS operations of one wave execute in order) ------------
        __builtin_amdgcn_wave_barrier();
        {
            const unsigned long long idle_mask = ~__ballot(live);
            const int n_idle = __popcll(idle_mask);
            if (pool_count > 0 && n_idle > 0) {
                const int k = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle_mask >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned)idle_mask, 0u));
                const int n_take = n_idle < pool_count ? n_idle : pool_count;
                if (!live && k < n_take) {
                    ops.unpack(it, s_pool + ((pool_head + k) & (kWave - 1)));
                    live = true;
                }
                pool_head = (pool_head + n_take) & (kWave - 1);
                pool_count -= n_take;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!drained && pool_count == 0 && __ballot(live) == 0ull) continue;   // every started item was done at once: start more
        if (drained && pool_count == 0) {
            // tail: no item is left to hand out or to take from the pool.  What remains are the long serial chains of stiff items.
            // A wave that holds exactly ONE of them (the usual case: they are 1 in 10^3 .. 10^4 items) runs it with the
            // item's state broadcast to the whole wave through v_readlane: every operand is then wave-uniform, the
            // compiler turns the accept / reject / output branches into scalar branches and drops the per-lane selects
            // (tools/attempt_probe.hip: the same solve with uniform and with per-lane operands; tools/tail_latency.py on
            // the product kernel).  The arithmetic is the same function on the same operands: bit-identical results.
            const bool reject = ops.reject_enabled();
            for (;;) {
                const unsigned long long tail_mask = __ballot(live);
                const int n_live = __popcll(tail_mask);
                if (n_live == 0) break;
                if (n_live == 1) {
                    const int src = __ffsll((unsigned long long)tail_mask) - 1;
                    Item u = ops.broadcast(it, src);
                    run_item_uniform(ops, u, reject);
                    live = false;
                    break;
                }
                // several stiff items in this wave: per-lane attempts until one of them is done, then look again
                int n_now, since_check = kRejectCheckEvery;     // first look at the bound at once
                do {
                    if (live) {
                        if (reject && ++since_check > kRejectCheckEvery) {
                            since_check = 0;
                            if (ops.certainly_rejected(it)) {
                                ops.cancel(it);
                                live = false;
                            }
                        }
                    }
                    if (live) {
                        SMC_ISA_MARK("lane_tail_attempt");
                        const int st = ops.attempt(it);
                        if (st != 0) {
                            ops.finish(it, st);
                            live = false;
                        }
                    }
                    n_now = __popcll(__ballot(live));
                } while (n_now == n_live);
            }
            break;
        }
        // attempts of the live lanes until kRefillAt lanes are idle (they then take items from the pool, which is refilled
        // above when it runs low): a tight inner loop (item state stays in its registers, one ballot and one scalar branch
        // per attempt) - the scheduling logic above runs once per hand-out, not once per attempt
        const int patience = ops.patience;   // attempts to go on waiting for ALL lanes once kRefillAt are idle (see above)
        int idle_now, waited = 0;
        do {
            SMC_ISA_MARK("bulk_attempt");
            if (live) {
                const int st = ops.attempt(it);
                if (st != 0) {
                    ops.finish(it, st);
                    live = false;
                }
            }
            idle_now = kWave - __popcll(__ballot(live));
            waited += (idle_now >= kRefillAt);                      // scalar
            // waiting for ALL lanes makes sense only while none of them is a straggler (looked at only once the wave could hand out)
            if (patience > 0 && idle_now >= kRefillAt && __ballot(live && ops.long_running(it)) != 0ull) waited = patience + 1;
        } while (idle_now < kRefillAt || (idle_now < kWave && waited <= patience));
    }
}

}  // namespace smc
