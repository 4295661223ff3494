"""CPU test of the stream-K work plan (host arithmetic in librf_flux, reached through rf_debug_sk_plan -- no GPU).

The kernel's correctness and deadlock-freedom rest on a few invariants of the plan; this test re-enacts the device
side's decode (gemm_bf16_sk_kernel) in Python over random grouped shapes and checks them:
  * every (tile, K-tile) iteration is processed exactly once;
  * a block stores at most one partial tile (it has exactly one scratch slot and one flag);
  * the block that finishes a split tile adds exactly the blocks holding that tile's earlier pieces, all of them on
    its own XCD with a LOWER worker index (=> lower hardware block index: dispatched earlier, no circular wait),
    and each of those produced its partial as the FIRST piece it processes;
  * no worker in a stream-K region has an empty range.
"""
import ctypes as C
import random

import pytest

from reflectionflow_amd import _lib as L


def make_desc(rows, N, ksegs):
    d = L.rf_gemm_desc()
    d.N, d.epilogue, d.num_groups = N, 0, len(rows)
    for gi, M in enumerate(rows):
        g = d.g[gi]
        g.M = M
        for si, K in enumerate(ksegs):
            g.seg[si].A, g.seg[si].lda, g.seg[si].W, g.seg[si].ldw, g.seg[si].K = 0x100000, K, 0x200000, K, K
        g.out, g.ldo = 0x300000, N
    return d


def plan(d, cus):
    out = (C.c_int32 * 64)()
    rc = L.load().rf_debug_sk_plan(C.byref(d), cus, out)
    assert rc in (0, 1), L.load().rf_last_error()
    if rc == 0:
        return None
    o = list(out)
    return dict(iter_start=o[0:5], nk=o[5:9], chunk_tile=o[9:17], dp_rounds=o[17:25], sk_begin=o[25:33], chunk_end=o[33:41],
                tiles_n=o[41], tiles_m=o[42:46], tile_start=o[46:50], T=o[50])


def simulate(pl, ngroups, cus):
    P, PL = cus // 8 * 8, cus // 8
    done = {}             # (tile, kt) -> block
    partial_of = {}       # block -> (tile, first piece?)
    owners = []           # (block, tile, ts, ub)
    rb = lambda x, w: pl["sk_begin"][x] + (w * (pl["chunk_end"][x] - pl["sk_begin"][x])) // PL

    def group_of_tile(t):
        g = 0
        while g + 1 < ngroups and t >= pl["tile_start"][g + 1]:
            g += 1
        return g

    def group_of_iter(it):
        g = 0
        while g + 1 < ngroups and it >= pl["iter_start"][g + 1]:
            g += 1
        return g

    for b in range(P):
        x, cl = b % 8, b // 8
        order = 0
        for r in range(pl["dp_rounds"][x]):
            tile = pl["chunk_tile"][x] + r * PL + cl
            g = group_of_tile(tile)
            for kt in range(pl["nk"][g]):
                assert (tile, kt) not in done, "iteration processed twice"
                done[(tile, kt)] = b
            order += 1
        it_begin, cur_end = rb(x, cl), rb(x, cl + 1)
        assert cur_end > it_begin, f"empty stream-K range for block {b}"
        while cur_end > it_begin:
            last = cur_end - 1
            g = group_of_iter(last)
            nk = pl["nk"][g]
            lt = (last - pl["iter_start"][g]) // nk
            ts = pl["iter_start"][g] + lt * nk
            ub = max(it_begin, ts)
            tile = pl["tile_start"][g] + lt
            for it in range(ub, cur_end):
                assert (tile, it - ts) not in done, "iteration processed twice"
                done[(tile, it - ts)] = b
            if cur_end != ts + nk:                      # not the tile's final piece -> partial
                assert b not in partial_of, f"block {b} stores two partials"
                partial_of[b] = (tile, order == pl["dp_rounds"][x])
            elif ub > ts:
                owners.append((b, tile, ts, ub, x, cl))
            cur_end = ub
            order += 1
    # coverage
    expect = 0
    for g in range(ngroups):
        expect += pl["tiles_m"][g] * pl["tiles_n"] * pl["nk"][g]
    assert len(done) == expect == pl["iter_start"][ngroups], "not every iteration is covered"
    # fix-up dependencies
    consumed = set()
    for b, tile, ts, ub, x, cl in owners:
        c2 = cl - 1
        while True:
            assert c2 >= 0, "owner scans past worker 0"
            b2 = x + 8 * c2
            assert b2 < b and b2 % 8 == x
            assert partial_of.get(b2, (None,))[0] == tile, f"block {b} would add a slot that holds another tile"
            assert b2 not in consumed, "a partial is consumed twice"
            consumed.add(b2)
            if rb(x, c2) <= ts:
                break
            c2 -= 1
    assert consumed == set(partial_of), "a stored partial is never consumed (its flag would stay set)"
    for b2, (tile, first) in partial_of.items():
        assert first, f"block {b2} produces its partial after other stream-K pieces: its consumer could spin"
    return len(owners), len(partial_of)


@pytest.mark.parametrize("cus", [256, 304, 64])
def test_stream_k_plan_invariants(cus):
    rnd = random.Random(cus)
    made = 0
    cases = [((512, 4096, 1024), 3072, (3072,)), ((4608, 1024), 3072, (3072, 12288, 128)), ((4608,), 3072, (3072,)),
             ((512, 4096, 1024), 9216, (3072, 128)), ((5632,), 12288, (3072,))]
    for _ in range(60):
        ng = rnd.randint(1, 3)
        rows = tuple(rnd.choice([rnd.randint(1, 700), rnd.randint(700, 6000)]) for _ in range(ng))
        N = rnd.choice([256, 1000, 3072, 4104, 9216, 12288])
        ks = tuple(64 * rnd.randint(1, 60) for _ in range(rnd.randint(1, 3)))
        cases.append((rows, N, ks))
    for rows, N, ks in cases:
        pl = plan(make_desc(rows, N, ks), cus)
        if pl is None:
            continue
        made += 1
        simulate(pl, len(rows), cus)
    assert made >= 20, f"only {made} launches qualified for stream-K: the test lost its coverage"
