"""GPU parity of the fp8 (W8A8) path of BASELINE cfg5: quantisation kernels, rf_gemm_w8a8 with every epilogue, and
the block / model level (tests further down) -- all through the C ABI.

The reference has no fp8 semantics; they are defined in include/rf_flux.h (rf_gemm_w8a8).  Two kinds of check:
  * IMPLEMENTATION parity (tight): the kernel against fp32 math on the SAME quantised operands -- e4m3 x e4m3
    products are exact in fp32, so the only differences are fp32 summation order and the bf16 rounding of the
    output: the bf16 kernels' tolerance applies (rel-L2 < 4e-3, element-wise 1.2e-2).
  * QUANTISATION cost (stated, loose): against the un-quantised fp32 result; e4m3 has 3 mantissa bits, so a linear
    layer with both operands quantised row-wise carries ~3-4 % relative error by construction.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import BF, assert_close, rnd, vt_unpermute

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def torch_quant_rows(x):
    """The kernels' arithmetic restated in torch fp32: scale = amax * (1/448), q = e4m3(clamp(x * (1/scale)))."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32, device=x.device), torch.ones_like(amax))
    inv = 1.0 / scale
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


@pytest.mark.parametrize("M,K0,K1", [(7, 256, 0), (130, 3072, 0), (64, 3072, 12288), (33, 1024, 8), (5, 16384, 0)])
def test_quant_rows_fp8_bit_exact(dev, M, K0, K1):
    from reflectionflow_amd import ops
    x0 = rnd(M, K0, dev=dev, scale=3.0)
    x1 = rnd(M, K1, dev=dev, scale=0.3) if K1 else None
    x0[0].zero_()                                   # an all-zero row: scale 1, codes 0
    if x1 is not None:
        x1[0].zero_()
    q, sc = ops.quant_rows_fp8(x0, x1)
    full = x0 if x1 is None else torch.cat([x0, x1], 1)
    qr, scr = torch_quant_rows(full)
    assert torch.equal(sc, scr), f"row scales differ: max {float((sc - scr).abs().max())}"
    assert float(sc[0]) == 1.0 and int(q[0].max()) == 0
    assert torch.equal(q, qr), f"{int((q != qr).sum())} of {q.numel()} fp8 codes differ"
    # strided input view (a column slice of a wider buffer)
    if K0 > 128:
        big = rnd(M, K0 + 64, dev=dev)
        q2, sc2 = ops.quant_rows_fp8(big[:, 64:K0])
        qr2, scr2 = torch_quant_rows(big[:, 64:K0])
        assert torch.equal(q2, qr2) and torch.equal(sc2, scr2)


@pytest.mark.parametrize("rows,D", [(5, 256), (130, 3072), (64, 1024)])
def test_layernorm_modulate_fp8(dev, rows, D):
    from reflectionflow_amd import ops
    x, sc, sh = rnd(rows, D, dev=dev, scale=2.0), rnd(D, dev=dev, scale=0.3), rnd(D, dev=dev, scale=0.3)
    q, rs = ops.layernorm_modulate_fp8(x, sc, sh)
    y = F.layer_norm(x.float(), (D,), eps=1e-6) * (1 + sc.float()) + sh.float()
    amax = y.abs().amax(1)
    assert torch.allclose(rs, amax / 448.0, rtol=1e-5), "row scale != amax/448"
    deq = ops.dequantize_fp8(q, rs)
    qr, scr = torch_quant_rows(y)
    deq_ref = ops.dequantize_fp8(qr, scr)
    # identical up to a few codes flipped by the different fp32 summation order of the LayerNorm statistics
    assert rel_l2(deq, deq_ref) < 3e-3, rel_l2(deq, deq_ref)
    assert (q != qr).float().mean() < 2e-3
    e = rel_l2(deq, y)
    print(f"  LN+mod -> fp8 rows [{rows}x{D}]: quantisation rel-L2 {e:.3e}")
    assert 1e-2 < e < 4e-2


def q_act(x):
    from reflectionflow_amd import ops
    return ops.quant_rows_fp8(x)


def ref_w8(A8, sa, W8, sw, bias=None):
    from reflectionflow_amd import ops
    acc = A8.view(torch.float8_e4m3fn).float() @ W8.view(torch.float8_e4m3fn).float().t()
    y = acc * sa[:, None] * sw[None, :]
    return y + bias.float() if bias is not None else y


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 256), (1000, 768, 512), (1, 512, 256), (1024, 1536, 1024),
                                   (130, 104, 128), (513, 3080, 384), (4608, 3072, 3072)])
def test_gemm_w8a8_store(dev, M, N, K):
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import Group, Seg
    x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
    A8, sa = q_act(x)
    W8, sw = ops.quantize_weight_fp8(W)
    out = torch.empty(M, N, device=dev, dtype=BF)
    ops.gemm_w8a8([Group([Seg(A8, W8)], bias=b, out=out, a_scale=sa, w_scale=sw)], N)
    assert_close(out, ref_w8(A8, sa, W8, sw, b), f"w8a8 store {M}x{N}x{K}")
    e = rel_l2(out, x.float() @ W.float().t() + b.float())
    print(f"  w8a8 {M}x{N}x{K}: rel-L2 vs un-quantised fp32 {e:.3e}")
    assert e < 6e-2


def test_gemm_w8a8_transpose_detecting(dev):
    """A = I (exact in e4m3) against an asymmetric W of small integers (exact in e4m3): catches any row/column or
    k-order mix-up in the fp8 fragment path (guide G9)."""
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import Group, Seg
    K, N = 256, 192
    x = torch.eye(K, device=dev, dtype=BF)
    W = ((torch.arange(K * N, device=dev).reshape(N, K) * 7 + torch.arange(N, device=dev)[:, None]) % 15 - 7).to(BF)
    A8, sa = q_act(x)
    W8, sw = ops.quantize_weight_fp8(W)
    out = torch.empty(K, N, device=dev, dtype=BF)
    ops.gemm_w8a8([Group([Seg(A8, W8)], out=out, a_scale=sa, w_scale=sw)], N)
    deq = ops.dequantize_fp8(W8, sw)
    assert torch.equal(out.float(), deq.t().to(BF).float())


def test_gemm_w8a8_epilogues_groups_segments(dev):
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, Group, Seg
    M, N, K = 520, 384, 256
    x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
    A8, sa = q_act(x)
    W8, sw = ops.quantize_weight_fp8(W)
    acc = ref_w8(A8, sa, W8, sw, b)
    out = torch.empty(M, N, device=dev, dtype=BF)
    ops.gemm_w8a8([Group([Seg(A8, W8)], bias=b, out=out, a_scale=sa, w_scale=sw)], N, RF_EPI_GELU)
    assert_close(out, F.gelu(acc, approximate="tanh"), "w8 gelu")
    res, gate = rnd(M, N, dev=dev), rnd(N, dev=dev)
    res2 = res.clone()
    ops.gemm_w8a8([Group([Seg(A8, W8)], bias=b, out=res2, residual=res2, gate=gate, a_scale=sa, w_scale=sw)], N, RF_EPI_GATE_RES)
    assert_close(res2, res.float() + gate.float() * acc, "w8 gate_res in place")
    # two K segments with ONE common row scale (single block proj_out = [attn | mlp])
    x2 = rnd(M, 384, dev=dev, scale=0.2)
    Wc = rnd(N, K + 384, dev=dev, scale=0.05)
    Ac, sac = ops.quant_rows_fp8(x, x2)
    Wc8, swc = ops.quantize_weight_fp8(Wc)
    ops.gemm_w8a8([Group([Seg(Ac[:, :K], Wc8[:, :K]), Seg(Ac[:, K:], Wc8[:, K:])], bias=b, out=out, a_scale=sac, w_scale=swc)], N)
    assert_close(out, ref_w8(Ac, sac, Wc8, swc, b), "w8 two segments")
    # grouped: token groups with their own weights and scales
    Ms = [96, 700, 130]
    xs = [rnd(m, K, dev=dev, seed=10 + i) for i, m in enumerate(Ms)]
    Ws = [rnd(N, K, dev=dev, scale=0.05, seed=20 + i) for i in range(3)]
    outs = [torch.empty(m, N, device=dev, dtype=BF) for m in Ms]
    qa = [q_act(t) for t in xs]
    qw = [ops.quantize_weight_fp8(t) for t in Ws]
    ops.gemm_w8a8([Group([Seg(qa[i][0], qw[i][0])], bias=b, out=outs[i], a_scale=qa[i][1], w_scale=qw[i][1]) for i in range(3)], N)
    for i in range(3):
        assert_close(outs[i], ref_w8(qa[i][0], qa[i][1], qw[i][0], qw[i][1], b), f"w8 group {i}")


@pytest.mark.parametrize("St,Si", [(32, 96), (512, 320)])
def test_gemm_w8a8_qkv_fused_norm_rope_and_gelu(dev, St, Si):
    """The double-block QKV launch (two token groups) and the single-block fused [q|k|v|mlp] launch in fp8: head-major
    q,k after per-head RMSNorm + RoPE (+ folded softmax scale), V^T tiles, GELU half."""
    from oracle import flux_oracle as O
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import RF_EPI_QKV, RF_EPI_QKV_GELU, Group, Seg
    H, D, MLP = 2, 256, 512
    S = St + Si
    xt, xi = rnd(St, D, dev=dev, seed=1), rnd(Si, D, dev=dev, seed=2)
    Wt, Wi = rnd(3 * D, D, dev=dev, scale=0.05, seed=3), rnd(3 * D + MLP, D, dev=dev, scale=0.05, seed=4)
    bt, bi = rnd(3 * D, dev=dev, seed=5), rnd(3 * D + MLP, dev=dev, seed=6)
    nw = [(1 + 0.1 * torch.randn(128, generator=torch.Generator().manual_seed(70 + i))).to(dev).to(BF) for i in range(4)]
    ids = torch.zeros(S, 3)
    ids[:, 1] = torch.arange(S) % 23
    ids[:, 2] = torch.arange(S) // 7
    cos, sin = O.FluxPosEmbed(10000, (16, 56, 56))(ids)
    cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()

    def nr(x, w, rows):                      # x [n, H*128] fp32 rows `rows` of the joint sequence -> [H, n, 128]
        n = x.shape[0]
        x = x.reshape(n, H, 128).permute(1, 0, 2)[None]
        xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
        return O.apply_rotary_emb(xn, (cos[rows], sin[rows]))[0]

    (At, sat), (Ai, sai) = q_act(xt), q_act(xi)
    # --- double-block flavour: text + image groups
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    Wi3, bi3 = Wi[:3 * D].contiguous(), bi[:3 * D].contiguous()
    (Wt8, swt), (Wi8, swi) = ops.quantize_weight_fp8(Wt), ops.quantize_weight_fp8(Wi3)
    groups = [Group([Seg(At, Wt8)], bias=bt, tok_offset=0, norm_q=nw[2], norm_k=nw[3], a_scale=sat, w_scale=swt),
              Group([Seg(Ai, Wi8)], bias=bi3, tok_offset=St, norm_q=nw[0], norm_k=nw[1], a_scale=sai, w_scale=swi)]
    ops.gemm_w8a8(groups, 3 * D, RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE)
    for acc, rows, wq, wk in ((ref_w8(At, sat, Wt8, swt, bt), slice(0, St), nw[2], nw[3]),
                              (ref_w8(Ai, sai, Wi8, swi, bi3), slice(St, S), nw[0], nw[1])):
        assert_close(q[:, rows], nr(acc[:, :D], wq, rows) * ops.QK_PRESCALE, "w8 qkv: q")
        assert_close(k[:, rows], nr(acc[:, D:2 * D], wk, rows), "w8 qkv: k")
        assert_close(vt_unpermute(vt, S)[:, rows], acc[:, 2 * D:].reshape(-1, H, 128).permute(1, 0, 2), "w8 qkv: v")
    # --- single-block flavour: fused [q|k|v|mlp], one group
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    Wf8, swf = ops.quantize_weight_fp8(Wi)
    hid = torch.empty(Si, MLP, device=dev, dtype=BF)
    ops.gemm_w8a8([Group([Seg(Ai, Wf8)], bias=bi, out=hid, tok_offset=St, norm_q=nw[0], norm_k=nw[1], a_scale=sai, w_scale=swf)],
                  3 * D + MLP, RF_EPI_QKV_GELU, n_split=3 * D, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin),
                  q_scale=ops.QK_PRESCALE)
    acc = ref_w8(Ai, sai, Wf8, swf, bi)
    rows = slice(St, S)
    assert_close(hid, F.gelu(acc[:, 3 * D:], approximate="tanh"), "w8 fused: gelu half")
    assert_close(q[:, rows], nr(acc[:, :D], nw[0], rows) * ops.QK_PRESCALE, "w8 fused: q")
    assert_close(k[:, rows], nr(acc[:, D:2 * D], nw[1], rows), "w8 fused: k")
    assert_close(vt_unpermute(vt, S)[:, rows], acc[:, 2 * D:3 * D].reshape(-1, H, 128).permute(1, 0, 2), "w8 fused: v")


def test_gemm_w8a8_mixed_precision_groups(dev):
    """One launch, three token groups: two fp8 (own weights / scales) and one bf16 with a LoRA-style second K-segment
    (a_scale = None) -- cfg5's condition rows riding with the fp8 text / image rows."""
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, Group, Seg
    N, K = 384, 256
    Ms = [96, 700, 130]
    xs = [rnd(m, K, dev=dev, seed=10 + i) for i, m in enumerate(Ms)]
    Ws = [rnd(N, K, dev=dev, scale=0.05, seed=20 + i) for i in range(3)]
    b, gate = rnd(N, dev=dev), rnd(N, dev=dev)
    res = [rnd(m, N, dev=dev, seed=40 + i) for i, m in enumerate(Ms)]
    outs = [r_.clone() for r_ in res]
    tl, Bl = rnd(Ms[2], 64, dev=dev), rnd(N, 64, dev=dev, scale=0.05)
    qa = [q_act(xs[i]) for i in range(2)]
    qw = [ops.quantize_weight_fp8(Ws[i]) for i in range(2)]
    groups = [Group([Seg(qa[i][0], qw[i][0])], bias=b, out=outs[i], residual=outs[i], gate=gate, a_scale=qa[i][1], w_scale=qw[i][1])
              for i in range(2)]
    groups.append(Group([Seg(xs[2], Ws[2]), Seg(tl, Bl)], bias=b, out=outs[2], residual=outs[2], gate=gate))
    ops.gemm_w8a8(groups, N, RF_EPI_GATE_RES)
    for i in range(2):
        assert_close(outs[i], res[i].float() + gate.float() * ref_w8(qa[i][0], qa[i][1], qw[i][0], qw[i][1], b), f"mixed: fp8 group {i}")
    ref = xs[2].float() @ Ws[2].float().t() + tl.float() @ Bl.float().t() + b.float()
    assert_close(outs[2], res[2].float() + gate.float() * ref, "mixed: bf16 group with a second K-segment")
    # the bf16 group of a mixed launch computes exactly what rf_gemm_bf16 computes
    o2 = res[2].clone()
    ops.gemm([Group([Seg(xs[2], Ws[2]), Seg(tl, Bl)], bias=b, out=o2, residual=o2, gate=gate)], N, RF_EPI_GATE_RES)
    assert torch.equal(o2, outs[2])


def test_gemm_w8a8_stream_k_matches_tile_per_block(dev):
    """cfg5-like tile counts (a last round that is mostly empty) take the stream-K schedule; results must equal the
    one-tile-per-block launch bit for bit, run to run."""
    from reflectionflow_amd import _lib, ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, Group, Seg
    lib = _lib.load()
    M, N, K = 5632, 3072, 3072          # 264 tiles on 256 CUs
    x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
    res, gate = rnd(M, N, dev=dev), rnd(N, dev=dev)
    A8, sa = q_act(x)
    W8, sw = ops.quantize_weight_fp8(W)
    outs = []
    for mode in (0, 1, 1):
        with ops.gemm_schedule(_lib.RF_SCHED_STREAMK if mode else _lib.RF_SCHED_TILE256):
            o = res.clone()
            ops.gemm_w8a8([Group([Seg(A8, W8)], bias=b, out=o, residual=o, gate=gate, a_scale=sa, w_scale=sw)], N, RF_EPI_GATE_RES)
            assert lib.rf_debug_last_gemm_path() == (2 if mode else 0)
            outs.append(o)
    assert_close(outs[0], res.float() + gate.float() * ref_w8(A8, sa, W8, sw, b), "w8 gate_res 264 tiles")
    assert torch.equal(outs[1], outs[2]), "stream-K fp8 launch is not bit-stable"
    assert rel_l2(outs[1], outs[0]) < 1e-3    # different fp32 summation order of split tiles only


def test_w8_errors_are_loud(dev):
    from reflectionflow_amd import ops
    from reflectionflow_amd.ops import Group, Seg
    x, W = rnd(64, 192, dev=dev), rnd(64, 192, dev=dev)
    A8, sa = q_act(x)
    W8, sw = ops.quantize_weight_fp8(W)
    out = torch.empty(64, 64, device=dev, dtype=BF)
    with pytest.raises(ops.RFError, match="multiple of 128"):
        ops.gemm_w8a8([Group([Seg(A8, W8)], out=out, a_scale=sa, w_scale=sw)], 64)
    with pytest.raises(ops.RFError, match="fp8 groups need rf_gemm_w8a8"):
        ops.gemm([Group([Seg(x[:, :128].contiguous(), W[:, :128].contiguous())], out=out, a_scale=sa, w_scale=sw)], 64)
    with pytest.raises(ops.RFError):
        ops.quant_rows_fp8(x.cpu())


# =====================================================================================================================
# block / model level: the product with dims.fp8 vs the fp32 oracle under the fp8 emulation of tests/w8_emulation.py.
#
# What can be asserted here.  The kernels' arithmetic is pinned exactly above (same quantised operands -> bf16-level
# agreement, bit-exact quantisers).  At block level the product and the emulation see inputs that differ by bf16
# rounding noise (~4e-3 relative), and e4m3's code spacing is 6-12 %: a few percent of the codes land on the other side
# of a rounding boundary, each moving its element by a full code step.  That "flip noise" is of the same order as the
# quantisation noise itself (measured: 0.6-0.8 x), so product-vs-emulation cannot be tighter than that for ANY correct
# implementation.  The tests therefore check the plumbing (right weights, scales, streams, epilogues) statistically:
#     cost_emu = rel-L2(emulated oracle, fp32 oracle)                 what the fp8 semantics cost in exact arithmetic
#     rel-L2(product, emulated oracle) <= 1.0 x cost_emu + 2 x e_bf16   flip noise does not exceed the quantisation noise
#     rel-L2(product, fp32 oracle)     <= 1.5 x cost_emu + 2 x e_bf16   the product is not noisier than its semantics
# with e_bf16 = rel-L2(eager bf16 oracle, fp32 oracle), the yard-stick of tests/test_model_gpu.py.  A wrong scale, a
# swapped weight or a stream quantised that should not be shows up as errors of order 1, far outside these bounds.
def check8(hip, emu, ref32, tb, what):
    assert torch.isfinite(hip.float()).all(), f"{what}: non-finite"
    hip, emu, ref32, tb = hip.cpu(), emu.cpu(), ref32.cpu(), tb.cpu()
    e_impl, e_t, e_cost, cost_emu = rel_l2(hip, emu), rel_l2(tb, ref32), rel_l2(hip, ref32), rel_l2(emu, ref32)
    print(f"  {what}: product vs emulated-fp8 oracle {e_impl:.3e}, product vs fp32 {e_cost:.3e}, emulation vs fp32 {cost_emu:.3e}, "
          f"eager-bf16 vs fp32 {e_t:.3e}")
    assert e_impl <= 1.0 * cost_emu + 2.0 * e_t, f"{what}: product vs emulated fp8 oracle {e_impl:.3e} (cost_emu {cost_emu:.3e})"
    assert e_cost <= 1.5 * cost_emu + 2.0 * e_t, f"{what}: product deviates {e_cost:.3e} from fp32 (cost_emu {cost_emu:.3e})"
    return e_impl, e_cost


@torch.no_grad()
@pytest.mark.parametrize("use_c", [False, True])
def test_blocks_fp8_vs_emulated_oracle(dev, use_c):
    import copy
    from oracle import flux_oracle as O
    from reflectionflow_amd.flux.block import block_forward, single_block_forward
    from tests import w8_emulation as EM
    from tests.golden_util import T, build, load
    from tests.test_model_gpu import bf16_oracle, g, to_product
    geom = "hd128"
    z = load(f"blocks_{geom}")
    om = build(geom, lora=True)
    ob = bf16_oracle(om)
    oe = EM.emulate_fp8(copy.deepcopy(om), 32, 64)
    pipe = to_product(om, dev)
    pm = pipe.transformer
    x, e, c, temb, ctemb = (T(z[k]) for k in ("x", "e", "c", "temb", "ctemb"))
    assert (e.shape[1], x.shape[1], c.shape[1]) == (32, 64, 16)
    rope = om.pos_embed(torch.cat([T(z["txt_ids"]), T(z["img_ids"])]))
    crope = om.pos_embed(T(z["cond_ids"]))
    cfg = {"union_cond_attn": True, "latent_lora": False}
    cfg8 = dict(cfg, fp8_weights=True)
    kw = lambda f: dict(condition_latents=f(c) if use_c else None, cond_rotary_emb=crope if use_c else None)  # noqa: E731
    f32, fbf, fg = (lambda a: a), (lambda a: a.to(BF)), (lambda a: g(a, dev))
    args = lambda m, f: dict(hidden_states=f(x), encoder_hidden_states=f(e), temb=f(temb),  # noqa: E731
                             cond_temb=f(ctemb) if use_c else None, image_rotary_emb=rope, **kw(f))
    ref = O.block_forward(om.transformer_blocks[0], model_config=cfg, **args(om, f32))
    emu = O.block_forward(oe.transformer_blocks[0], model_config=cfg, **args(oe, f32))
    tb = O.block_forward(ob.transformer_blocks[0], model_config=cfg, **args(ob, fbf))
    hp = block_forward(pm.transformer_blocks[0], model_config=cfg8, **args(pm, fg))
    hp16 = block_forward(pm.transformer_blocks[0], model_config=cfg, **args(pm, fg))
    for i, name in enumerate(("txt", "img", "cond")):
        if hp[i] is None:
            continue
        check8(hp[i], emu[i], ref[i], tb[i], f"double/{name} cond={use_c}")
    assert not torch.equal(hp[1], hp16[1]), "fp8_weights must change the numbers (is the fp8 path running?)"
    xs = torch.cat([e, x], 1)
    skw = lambda f: dict(condition_latents=f(c), cond_temb=f(ctemb), cond_rotary_emb=crope) if use_c else {}  # noqa: E731
    ref = O.single_block_forward(om.single_transformer_blocks[0], f32(xs), f32(temb), image_rotary_emb=rope, model_config=cfg, **skw(f32))
    emu = O.single_block_forward(oe.single_transformer_blocks[0], f32(xs), f32(temb), image_rotary_emb=rope, model_config=cfg, **skw(f32))
    tb = O.single_block_forward(ob.single_transformer_blocks[0], fbf(xs), fbf(temb), image_rotary_emb=rope, model_config=cfg, **skw(fbf))
    hp = single_block_forward(pm.single_transformer_blocks[0], fg(xs), fg(temb), image_rotary_emb=rope, model_config=cfg8, **skw(fg))
    tup = lambda v: v if isinstance(v, tuple) else (v,)  # noqa: E731
    for name, a, b_, c_, d_ in zip(("main", "cond"), tup(hp), tup(emu), tup(ref), tup(tb)):
        check8(a, b_, c_, d_, f"single/{name} cond={use_c}")


@torch.no_grad()
def test_full_width_blocks_fp8_cfg4_tokens(dev):
    """FLUX.1-dev width (D=3072, 24 heads, mlp 12288) at 512 + 4096 + 1024 tokens, r=16 LoRA on the condition rows:
    fp8 text/image streams + bf16 condition stream in the same block, vs the emulated oracle evaluated on the GPU."""
    import copy
    from oracle import flux_oracle as O
    from reflectionflow_amd.flux import modules as M
    from reflectionflow_amd.flux.block import block_forward, single_block_forward
    from reflectionflow_amd.flux.pipeline import FluxPipeline
    from tests import w8_emulation as EM
    D, H, St, Si, Sc = 3072, 24, 512, 4096, 1024
    cfgm = dict(num_layers=1, num_single_layers=1)
    torch.manual_seed(0)
    with torch.device(dev):
        om = O.FluxTransformer2DModel(**cfgm).float().eval()
    O.inject_lora(om, r=16, alpha=16.0)
    om = om.to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    for name, p in om.named_parameters():
        std = (1.0 / 16 if ".lora_A." in name else 0.02)
        p.copy_((1.0 if (name.endswith("weight") and p.ndim == 1) else 0.0) + std * torch.randn(p.shape, generator=gen, device=dev))
    pm = M.FluxTransformer2DModel(**cfgm)
    sd = om.state_dict()
    pm.load_state_dict({k.replace(".base_layer", ""): v.cpu() for k, v in sd.items() if ".lora_" not in k})
    pipe = FluxPipeline(pm.to(dev).to(BF))
    lora = {}
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            name, which = k.split(".lora_")
            lora[f"transformer.{name}.lora_{which[0]}.weight"] = v
    pipe.load_lora_weights(lora, alpha=16.0)
    pm = pipe.transformer
    oe = EM.emulate_fp8(copy.deepcopy(om), St, Si)
    ob = copy.deepcopy(om).to(BF)
    r = lambda *s: torch.randn(*s, generator=gen, device=dev).to(BF)  # noqa: E731
    x, e, c, temb, ctemb = r(1, Si, D), r(1, St, D), r(1, Sc, D), r(1, D), r(1, D)
    ids = torch.cat([torch.zeros(St, 3), O.prepare_latent_image_ids(64, 64)])
    pe = O.FluxPosEmbed(10000, (16, 56, 56))
    rope = tuple(t.to(dev) for t in pe(ids))
    crope = tuple(t.to(dev) for t in pe(O.condition_ids_for(512)))
    cfg = {"union_cond_attn": True, "latent_lora": False}
    cfg8 = dict(cfg, fp8_weights=True)
    f = lambda t: t.float()  # noqa: E731
    a32 = dict(hidden_states=f(x), encoder_hidden_states=f(e), condition_latents=f(c), temb=f(temb), cond_temb=f(ctemb))
    abf = dict(hidden_states=x, encoder_hidden_states=e, condition_latents=c, temb=temb, cond_temb=ctemb)
    rk = dict(cond_rotary_emb=crope, image_rotary_emb=rope)
    ref = O.block_forward(om.transformer_blocks[0], model_config=cfg, **a32, **rk)
    emu = O.block_forward(oe.transformer_blocks[0], model_config=cfg, **a32, **rk)
    tb = O.block_forward(ob.transformer_blocks[0], model_config=cfg, **abf, **rk)
    hp = block_forward(pm.transformer_blocks[0], model_config=cfg8, **abf, **rk)
    for i, name in enumerate(("txt", "img", "cond")):
        check8(hp[i], emu[i], ref[i], tb[i], f"full-width double/{name}")
    xs = torch.cat([e, x], 1)
    s32 = dict(condition_latents=f(c), cond_temb=f(ctemb), cond_rotary_emb=crope)
    sbf = dict(condition_latents=c, cond_temb=ctemb, cond_rotary_emb=crope)
    ref = O.single_block_forward(om.single_transformer_blocks[0], f(xs), f(temb), image_rotary_emb=rope, model_config=cfg, **s32)
    emu = O.single_block_forward(oe.single_transformer_blocks[0], f(xs), f(temb), image_rotary_emb=rope, model_config=cfg, **s32)
    tb = O.single_block_forward(ob.single_transformer_blocks[0], xs, temb, image_rotary_emb=rope, model_config=cfg, **sbf)
    hp = single_block_forward(pm.single_transformer_blocks[0], xs, temb, image_rotary_emb=rope, model_config=cfg8, **sbf)
    for name, a, b_, c_, d_ in zip(("main", "cond"), hp, emu, ref, tb):
        check8(a, b_, c_, d_, f"full-width single/{name}")
