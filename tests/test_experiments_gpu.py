"""GPU tests of the EXPERIMENTS library (librf_flux_exp.so = the product sources + csrc/experiments/*.inc, built with
`make -C reflectionflow_amd/csrc EXPERIMENTS=1`): the A/B kernels kept for the studies in profiles/ stay bit-identical to
the shipped ones.  Skipped when that library has not been built (it is NOT part of the product and `build()` does not
build it); nothing here runs through librf_flux.so's kernel selection.
"""
import os

import pytest
import torch

from tests.test_kernels_gpu import BF, assert_close, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture
def exp(dev):
    from reflectionflow_amd import _lib
    if not os.path.exists(_lib.EXP_LIB_PATH):
        pytest.skip("librf_flux_exp.so not built (make -C reflectionflow_amd/csrc EXPERIMENTS=1)")
    try:
        lib = _lib.load_experiments()
    except _lib.RFError as e:                      # built from older sources than the binding: not the product's problem
        if "stale" in str(e):
            pytest.skip(str(e))
        raise
    yield lib
    _lib.unload_experiments()


@pytest.mark.parametrize("rows,N,K", [((512, 4096), 3072, 3072), ((300, 5000, 77), 1536, 320), ((4608,), 3072, 64)])
def test_gemm_mfma_shapes_and_experimental_loops_agree(dev, exp, rows, N, K):
    """The shipped 256x256 kernel multiplies with v_mfma_f32_16x16x32_bf16 in evenly loaded phases; rf_debug_gemm_even(0)
    selects the 8/4/8/4 phases (which the fp8 and stream-K kernels still use), rf_debug_gemm_mi16(0) the 32x32x16 loop.  Both walk the K-tiles in the same order with fp32 accumulation per output element, so the results
    are bit-identical -- as are the experimental main loops kept in the library (rf_debug_force_gemm_tile 258: one wave
    per SIMD over an LDS ring; 259 + variant 5 / 6: the balanced and the evenly loaded ping-pong phases on 32x32x16),
    with grouped rows, a ragged last tile in M and N, 1 / 5 / 48 K-tiles and the gate-residual epilogue."""
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, Group, Seg
    lib = exp
    xs = [rnd(m, K, dev=dev, seed=150 + i) for i, m in enumerate(rows)]
    Ws = [rnd(N, K, dev=dev, scale=0.05, seed=160 + i) for i in range(len(rows))]
    b, gate = rnd(N, dev=dev), rnd(N, dev=dev)
    res = [rnd(m, N, dev=dev, seed=170 + i) for i, m in enumerate(rows)]
    outs = {}
    try:
        for name, tile, var, mi16, even in (("16x16x32", 256, 0, 1, 1), ("16x16x32 8/4/8/4 phases", 256, 0, 1, 0), ("32x32x16", 256, 0, 0, 1),
                                            ("one wave per SIMD", 258, 0, 1, 1), ("balanced 32x32", 259, 5, 1, 1), ("even 32x32", 259, 6, 1, 1),
                                            ("16x16 harness, 8/4/8/4", 259, 7, 1, 1), ("16x16 harness, even", 259, 11, 1, 1)):
            lib.rf_debug_force_gemm_sk(0)
            lib.rf_debug_force_gemm_tile(tile)
            lib.rf_debug_gemm_w4_knock(var)
            lib.rf_debug_gemm_mi16(mi16)
            lib.rf_debug_gemm_even(even)
            o = [r_.clone() for r_ in res]
            ops.gemm([Group([Seg(xs[i], Ws[i])], bias=b, out=o[i], residual=o[i], gate=gate) for i in range(len(rows))], N,
                     RF_EPI_GATE_RES, splitk_ws=False)
            outs[name] = o
    finally:
        lib.rf_debug_force_gemm_sk(-1)
        lib.rf_debug_force_gemm_tile(0)
        lib.rf_debug_gemm_w4_knock(0)
        lib.rf_debug_gemm_mi16(1)
        lib.rf_debug_gemm_even(1)
    for i in range(len(rows)):
        assert_close(outs["16x16x32"][i], res[i].float() + gate.float() * (xs[i].float() @ Ws[i].float().t() + b.float()), f"group {i}")
        for name in outs:
            assert torch.equal(outs[name][i], outs["16x16x32"][i]), f"group {i}: '{name}' differs from the shipped kernel"


def test_gemm_stream_k_on_both_mfma_shapes(dev, exp):
    """Stream-K (partial accumulators travel through scratch quad by quad) on the 16x16x32 kernel and on the 32x32x16 one
    (S = 5632-like rows, 264 tiles): for a given schedule the two MFMA shapes agree bit for bit; stream-K splits the K sum
    of a tile between workers, so against one tile per block it is compared with the usual tolerance."""
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import Group, Seg
    lib = exp
    rows, N, K = (512, 4096, 1024), 3072, 3072
    xs = [rnd(m, K, dev=dev, seed=250 + i) for i, m in enumerate(rows)]
    Ws = [rnd(N, K, dev=dev, scale=0.05, seed=260 + i) for i in range(len(rows))]
    outs = {}
    try:
        for mi16 in (1, 0):
            for sk in (0, 1):
                lib.rf_debug_gemm_mi16(mi16)
                lib.rf_debug_force_gemm_sk(sk)
                o = [torch.empty(m, N, dtype=BF, device=dev) for m in rows]
                ops.gemm([Group([Seg(xs[i], Ws[i])], out=o[i]) for i in range(len(rows))], N)
                assert lib.rf_debug_last_gemm_path() == (2 if sk else 0)
                outs[(mi16, sk)] = o
    finally:
        lib.rf_debug_force_gemm_sk(-1)
        lib.rf_debug_gemm_mi16(1)
    for i in range(len(rows)):
        for sk in (0, 1):
            assert torch.equal(outs[(0, sk)][i], outs[(1, sk)][i]), f"group {i}, stream-K={sk}: the MFMA shapes disagree"
        assert_close(outs[(1, 1)][i], xs[i].float() @ Ws[i].float().t(), f"stream-K group {i}")
        assert_close(outs[(1, 1)][i], outs[(1, 0)][i].float(), f"stream-K vs tile-per-block, group {i}")   # (a bf16 ulp apart at most)


@pytest.mark.parametrize("M,N,K,K2", [(1024, 64, 3072, 0), (1024, 64, 3072, 12288), (1000, 32, 2048, 0), (517, 128, 1024, 512),
                                      (16384, 64, 3072, 0), (5, 16, 64, 0)])
def test_gemm_skinny_lora_down(dev, exp, M, N, K, K2):
    """x . lora_A^T as the engine issues it (no bias, plain store, N = r_pad <= 128, one or two activation segments, output
    rows 256 wide): the experimental skinny-N kernel [path 3, off by default: slower] vs fp32 and vs the split-K route
    [path 1 or 0], bit-stable, and nothing written outside its N columns or M rows."""
    from reflectionflow_amd import ops
    lib = exp
    x, A = rnd(M, K, dev=dev), rnd(N, K + K2, dev=dev, scale=0.05)
    segs = [ops.Seg(x, A[:, :K])]
    ref = x.float() @ A[:, :K].float().t()
    if K2:
        x2 = rnd(M, K2, dev=dev, seed=7)
        segs.append(ops.Seg(x2, A[:, K:]))
        ref = ref + x2.float() @ A[:, K:].float().t()
    outs = []
    try:
        for skinny in (1, 1, 0):
            lib.rf_debug_gemm_skinny(skinny)
            y = torch.full((M + 3, 256), 7.0, dtype=BF, device=dev)
            ops.gemm([ops.Group(segs, out=y[:M, :N])], N, ops.RF_EPI_STORE)
            assert (lib.rf_debug_last_gemm_path() == 3) == bool(skinny)
            outs.append(y)
    finally:
        lib.rf_debug_gemm_skinny(0)
    assert_close(outs[0][:M, :N], ref, "skinny vs fp32")
    assert torch.equal(outs[0], outs[1]), "the skinny kernel is not deterministic"
    assert (outs[0][:, N:] == 7.0).all() and (outs[0][M:] == 7.0).all(), "wrote outside its block"
    assert_close(outs[0][:M, :N], outs[2][:M, :N].float(), "skinny vs tiled route")


@pytest.mark.parametrize("S,H", [(2048, 8), (4608, 8)])
def test_attention_one_wave_per_simd_is_bit_identical(dev, exp, S, H):
    """attn_fwd_kernel_v6 (one wave per SIMD, 64 queries per wave) walks the keys in v5's order: bit-identical outputs."""
    from reflectionflow_amd import ops
    from tests.test_round2_gpu import _prescaled_case
    q, k, vt, ref, bound = _prescaled_case(H, S, dev, seed=S + H)
    lib = exp
    try:
        lib.rf_debug_attn_v5(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_mix(0)     # the plain one-size grid of both kernels
        lib.rf_debug_attn_knock(256)     # v5 with its row sums formed by v_add_f32 as in v6 (the shipped v5 forms them on the matrix pipe)
        o5 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound)
        assert lib.rf_debug_last_attn_path() == 5
        lib.rf_debug_attn_knock(0)
        lib.rf_debug_attn_v6(1)
        o6 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound)
        assert lib.rf_debug_last_attn_path() == 7
    finally:
        lib.rf_debug_attn_v5(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v6(0); lib.rf_debug_attn_mix(-1); lib.rf_debug_attn_knock(0)
    assert torch.equal(o5, o6)
    assert_close(o6, ref, f"attention v6 S={S}", atol=2e-3)
