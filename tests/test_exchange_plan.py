"""The send / receive plan of the sharded resampling exchange (include/smc_hip.h: smc_exchange_plan, csrc/exchange_plan.h),
checked on the CPU for world sizes 2..8 - the arithmetic of smc_resample_phase3 / rccl_exchange_blocks that only several real
ranks would otherwise exercise (the driver's multi-GPU run is the first execution with real ranks).

For every rank the library computes: what stays (own), one contiguous block per peer in the send staging (send_off, send_cnt,
src_lo), one block per peer in the receive staging (recv_off, recv_cnt) and the rows of p_filt it is spread over (recv_row),
and the first stale row.  The test plays the exchange with NumPy exactly as rccl_exchange_blocks does - send staging ->
receive staging -> rows - and demands
  * sender and receiver agree:   send_cnt[s -> r] == recv_cnt[r <- s]  for every pair,
  * blocks do not overlap in either staging buffer and are laid out in peer order,
  * every row of every rank is written exactly once (own + received + stale tile [0, n_local)),
  * the result equals the reference's sequential emission (Micmem_SMC_main.py:178-184): np.repeat of the ancestors in
    order, then the untouched rows,
for adversarial offspring splits: all mass on the last rank, on the first, empty ranks, one particle taking everything,
total = N - 1 (stale row), total = N exactly, random splits.  No GPU and no context are needed (host arithmetic only).
"""
import ctypes

import numpy as np
import pytest


def _plan(L, W, R, nl, base, cnt):
    i64 = ctypes.c_int64
    arr = lambda: (i64 * W)()                                            # noqa: E731
    so, sc, sl, ro, rc, rr = arr(), arr(), arr(), arr(), arr(), arr()
    own = (i64 * 3)()
    stale = i64(0)
    b = (i64 * W)(*[int(x) for x in base])
    c = (i64 * W)(*[int(x) for x in cnt])
    st = L.smc_exchange_plan(W, R, nl, b, c, so, sc, sl, ro, rc, rr, own, ctypes.byref(stale))
    assert st == 0, st
    return dict(send_off=list(so), send_cnt=list(sc), src_lo=list(sl), recv_off=list(ro), recv_cnt=list(rc), recv_row=list(rr),
                own=list(own), stale_lo=stale.value)


def _splits(W, nl, rs):
    N = W * nl
    yield "all on the last rank", [0] * (W - 1) + [N]
    yield "all on the first rank", [N] + [0] * (W - 1)
    yield "one short (stale row)", [nl] * (W - 1) + [nl - 1]
    yield "one short, all on the last rank", [0] * (W - 1) + [N - 1]
    yield "nothing at all", [0] * W
    yield "balanced", [nl] * W
    yield "alternating empty ranks", [2 * nl if q % 2 == 0 and q + 1 < W else (0 if q % 2 else nl) for q in range(W)]
    yield "middle rank takes everything", [N if q == W // 2 else 0 for q in range(W)]
    for k in range(6):
        w = rs.dirichlet(np.full(W, 0.3))
        c = np.floor(w * N).astype(np.int64)
        c[rs.randint(W)] += N - c.sum() - (k % 2)                         # total N or N - 1
        yield f"random {k}", [int(x) for x in c]


@pytest.mark.parametrize("W", [2, 3, 4, 5, 6, 7, 8])
def test_sender_and_receiver_plans_agree_and_tile_every_rank(pkg, W):
    L = pkg.lib()
    rs = np.random.RandomState(W)
    for nl in (1, 7, 64):
        N = W * nl
        for name, cnt in _splits(W, nl, rs):
            cnt = np.asarray(cnt, dtype=np.int64)
            assert cnt.sum() <= N and cnt.min() >= 0, (name, cnt)
            base = np.concatenate([[0], np.cumsum(cnt)[:-1]])
            plans = [_plan(L, W, R, nl, base, cnt) for R in range(W)]
            # the reference's emission: offspring of rank q, local index j, carries the tag (q, j); old rows carry (-1, row)
            expect = np.full((N, 2), -1, dtype=np.int64)
            expect[:, 1] = np.arange(N)
            pos = 0
            for q in range(W):
                for j in range(cnt[q]):
                    expect[pos] = (q, j)
                    pos += 1
            # ---- play the exchange ----
            sendbuf = []
            for s in range(W):
                p = plans[s]
                tot = sum(p["send_cnt"])
                buf = np.full((tot, 2), -7, dtype=np.int64)
                off = 0
                for q in range(W):                                     # blocks in peer order, back to back
                    if p["send_cnt"][q] == 0:
                        continue
                    assert q != s and p["send_off"][q] == off, (name, s, q)
                    j0 = p["src_lo"][q]
                    assert 0 <= j0 and j0 + p["send_cnt"][q] <= cnt[s]
                    buf[off:off + p["send_cnt"][q], 0] = s
                    buf[off:off + p["send_cnt"][q], 1] = np.arange(j0, j0 + p["send_cnt"][q])
                    off += p["send_cnt"][q]
                assert off == tot
                sendbuf.append(buf)
            got = np.full((N, 2), -9, dtype=np.int64)
            for r in range(W):
                p = plans[r]
                written = np.zeros(nl, dtype=np.int64)
                rows = got[r * nl:(r + 1) * nl]
                o_src, o_cnt, o_row = p["own"]
                if o_cnt:
                    rows[o_row:o_row + o_cnt, 0] = r
                    rows[o_row:o_row + o_cnt, 1] = np.arange(o_src, o_src + o_cnt)
                    written[o_row:o_row + o_cnt] += 1
                roff = 0
                for s in range(W):
                    c_rs = p["recv_cnt"][s]
                    assert c_rs == plans[s]["send_cnt"][r] if s != r else c_rs == 0, (name, W, nl, s, r)   # the two sides agree
                    if c_rs == 0:
                        continue
                    assert p["recv_off"][s] == roff
                    blk = sendbuf[s][plans[s]["send_off"][r]:plans[s]["send_off"][r] + c_rs]
                    row = p["recv_row"][s]
                    assert 0 <= row and row + c_rs <= nl
                    rows[row:row + c_rs] = blk
                    written[row:row + c_rs] += 1
                    roff += c_rs
                st = p["stale_lo"]
                assert 0 <= st <= nl
                rows[st:, 0] = -1
                rows[st:, 1] = np.arange(r * nl + st, (r + 1) * nl)
                written[st:] += 1
                assert np.all(written == 1), (name, W, nl, r, written)     # every row exactly once
            assert np.array_equal(got, expect), (name, W, nl)


def test_plan_rejects_a_wrong_prefix(pkg):
    L = pkg.lib()
    i64 = ctypes.c_int64
    b = (i64 * 2)(0, 5)
    c = (i64 * 2)(4, 4)                                                   # prefix of (4, 4) is (0, 4), not (0, 5)
    assert L.smc_exchange_plan(2, 0, 4, b, c, None, None, None, None, None, None, None, None) == 1
    assert L.smc_exchange_plan(0, 0, 4, b, c, None, None, None, None, None, None, None, None) == 2
