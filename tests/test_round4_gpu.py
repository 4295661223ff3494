"""Round 4: the one-wave-per-SIMD GEMM loop (RF_SCHED_W4: 4 waves x 128x128 wave tiles on 16x16x32 MFMAs, csrc/gemm_w4.hpp) must
agree BIT FOR BIT with the 8-wave tile-per-block kernel (same MFMA shape, same K order) on every epilogue, on ragged M / N (its
operand rows beyond M / N come from the buffer range check, not from clamped rows), on K-segment boundaries and token groups."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import assert_close, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _both(fn):
    """(8-wave result, W4 result) -- after checking that the round-5 form W4B (three half-stage barriers per K-tile, csrc/gemm_w4b.hpp)
    gives the W4 result bit for bit too"""
    from reflectionflow_amd import _lib as L, ops
    outs = []
    for sched in (L.RF_SCHED_TILE256, L.RF_SCHED_W4, L.RF_SCHED_W4B):
        with ops.gemm_schedule(sched):
            outs.append(fn())
    assert torch.equal(outs[1], outs[2]), "W4B and W4 disagree"
    return outs[:2]


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (700, 1032, 128), (1000, 768, 512), (4608, 3072, 3072), (130, 104, 192), (513, 3080, 64), (300, 512, 64), (64, 8, 128)])
def test_w4_store_equals_tile256(dev, M, N, K):
    from reflectionflow_amd import ops
    x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
    a, c = _both(lambda: ops.linear(x, W, b, splitk_ws=False))
    assert_close(c, x.float() @ W.float().t() + b.float(), f"w4 {M}x{N}x{K}")
    assert torch.equal(a, c), "W4 and the 8-wave kernel disagree"


@pytest.mark.parametrize("Ks", [(64,), (128,), (64, 64), (64, 128, 64), (192, 64), (320, 64, 128), (1024, 128)])
def test_w4_k_segments_and_epilogues(dev, Ks):
    from reflectionflow_amd import ops
    M, N = 700, 1032
    segs, ref = [], torch.zeros(M, N, device=dev)
    for i, K in enumerate(Ks):
        x, W = rnd(M, K, dev=dev, seed=3 * i + len(Ks)), rnd(N, K, dev=dev, scale=0.05, seed=3 * i + 1)
        segs.append(ops.Seg(x, W))
        ref += x.float() @ W.float().t()
    b, gate, res = rnd(N, dev=dev, seed=99), rnd(N, dev=dev, seed=98), rnd(M, N, dev=dev, seed=97)

    def gelu():
        y = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm([ops.Group(segs, bias=b, out=y)], N, ops.RF_EPI_GELU, splitk_ws=False)
        return y

    def gate_res():
        y = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm([ops.Group(segs, bias=b, out=y, residual=res, gate=gate)], N, ops.RF_EPI_GATE_RES, splitk_ws=False)
        return y
    a, c = _both(gelu)
    assert_close(c, F.gelu(ref + b.float(), approximate="tanh"), f"w4 gelu {Ks}")
    assert torch.equal(a, c)
    a, c = _both(gate_res)
    assert_close(c, res.float() + gate.float() * (ref + b.float()), f"w4 gate_res {Ks}")
    assert torch.equal(a, c)


@pytest.mark.parametrize("H", [2, 1, 3])
def test_w4_token_groups_and_qkv_epilogue(dev, H):
    """Three token groups (text / image / condition) with a LoRA K-segment on the last, through the fused QKV + RMSNorm + RoPE epilogue.
    H = 1, 3: q|k and v|out-of-range strips share a 256-column block, so the 8-wave kernel runs its swapped (register-direct epilogue)
    and natural (V^T) wave orientations side by side in one workgroup."""
    from oracle import flux_oracle as O
    from reflectionflow_amd import ops
    D = 128 * H
    St, Si, Sc = 40, 300, 70
    S = St + Si + Sc
    xs = [rnd(m, D, dev=dev, seed=10 + i) for i, m in enumerate((St, Si, Sc))]
    Ws = [rnd(3 * D, D, dev=dev, scale=0.05, seed=20 + i) for i in range(2)]
    bs = [rnd(3 * D, dev=dev, seed=30 + i) for i in range(2)]
    tl, Bl = rnd(Sc, 64, dev=dev, seed=40), rnd(3 * D, 64, dev=dev, scale=0.05, seed=41)
    nw = [(1 + 0.1 * rnd(128, dev=dev, seed=50 + i).float()).to(BF) for i in range(4)]
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 16, torch.arange(S) % 16], 1).to(dev)
    cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))

    def run():
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        groups = [ops.Group([ops.Seg(xs[0], Ws[0])], bias=bs[0], tok_offset=0, norm_q=nw[2], norm_k=nw[3]),
                  ops.Group([ops.Seg(xs[1], Ws[1])], bias=bs[1], tok_offset=St, norm_q=nw[0], norm_k=nw[1]),
                  ops.Group([ops.Seg(xs[2], Ws[1]), ops.Seg(tl, Bl)], bias=bs[1], tok_offset=St + Si, norm_q=nw[0], norm_k=nw[1])]
        ops.gemm(groups, 3 * D, ops.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE, splitk_ws=False)
        return torch.cat([q.flatten(), k.flatten(), vt.flatten()])
    a, c = _both(run)
    assert torch.isfinite(c.float()).all() and float(c.float().abs().max()) > 0
    assert torch.equal(a, c)


@torch.no_grad()
def test_merged_lora_equals_k_segment_lora_within_bf16(dev):
    """pipe.enable_merged_lora(): the condition rows multiply by bf16(W + s B A) instead of carrying the low-rank K-segment.  On the
    fixture model (reference outputs in tests/golden/transformer_hd128.npz) both forms must meet the same bound against the
    reference's fp32 output, the merged form may differ from the K-segment form only at bf16 level, and LoRA gating must survive
    (without a condition and latent_lora = False the output is exactly the base model's)."""
    from tests.golden_util import T, build, load
    from tests.test_model_gpu import check, g, to_product, bf16_oracle
    from oracle import flux_oracle as O
    from reflectionflow_amd.flux.transformer import tranformer_forward
    z = load("transformer_hd128")
    om = build("hd128", lora=True)
    pipe = to_product(om, dev)
    kw = dict(hidden_states=g(T(z["lat"]), dev), encoder_hidden_states=g(T(z["pe"]), dev), pooled_projections=g(T(z["pooled"]), dev),
              timestep=T(z["t"]).to(dev), guidance=T(z["g"]).to(dev), img_ids=T(z["img_ids"]).to(dev), txt_ids=T(z["txt_ids"]).to(dev),
              joint_attention_kwargs=None, return_dict=False)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    cond = dict(condition_latents=g(T(z["cond"]), dev), condition_ids=T(z["cond_ids"]).to(dev), condition_type_ids=None)
    nocond = dict(condition_latents=None, condition_ids=None, condition_type_ids=None)
    seg = tranformer_forward(pipe.transformer, model_config=cfg, **cond, **kw)[0]
    base_nocond = tranformer_forward(pipe.transformer, model_config=cfg, **nocond, **kw)[0]
    pipe.enable_merged_lora()
    mrg = tranformer_forward(pipe.transformer, model_config=cfg, **cond, **kw)[0]
    mrg_nocond = tranformer_forward(pipe.transformer, model_config=cfg, **nocond, **kw)[0]
    ref = T(z["out_lora_cond_nolatlora"])
    ob = bf16_oracle(om)
    tb = O.tranformer_forward(ob, condition_latents=T(z["cond"]).to(BF), condition_ids=T(z["cond_ids"]), model_config=cfg,
                              hidden_states=T(z["lat"]).to(BF), encoder_hidden_states=T(z["pe"]).to(BF), pooled_projections=T(z["pooled"]).to(BF),
                              timestep=T(z["t"]), guidance=T(z["g"]), img_ids=T(z["img_ids"]), txt_ids=T(z["txt_ids"]))[0]
    e_seg = check(seg, ref, tb, "K-segment LoRA")
    e_mrg = check(mrg, ref, tb, "merged LoRA")
    print(f"  rel-L2 vs the reference fp32 output: K-segment {e_seg[0]:.3e}, merged {e_mrg[0]:.3e}, torch-bf16 {e_seg[1]:.3e}")
    assert float((mrg.float() - seg.float()).norm() / seg.float().norm()) < 1.5e-2
    assert torch.equal(mrg_nocond, base_nocond), "without a condition the merged copies must not be touched"
    assert not torch.equal(mrg, base_nocond)
    pipe.enable_merged_lora(False)
    assert torch.equal(tranformer_forward(pipe.transformer, model_config=cfg, **cond, **kw)[0], seg)
