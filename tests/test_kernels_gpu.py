"""GPU parity of every HIP kernel behind the C ABI, one op at a time.

Reference = fp32 torch math on the same inputs (for the leaf semantics: the oracle's own
RMSNorm / apply_rotary_emb / F.scaled_dot_product_attention restatement).  Tolerances are
stated against the fp32 result and sized by bf16 output rounding (rel 2^-8 = 3.9e-3).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.pins import pins

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def rnd(*shape, dev, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) % 100000))
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(BF)


def assert_close(got, ref, what, rtol=1.2e-2, atol=None):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    if atol is None:
        atol = 1.2e-2 * float(ref.abs().mean()) + 1e-6
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    rel_l2 = float((got - ref).norm() / (ref.norm() + 1e-12))
    if bad.any():
        idx = torch.nonzero(bad)[:8].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements off, max err {float(err.max()):.4g}, "
                             f"rel-L2 {rel_l2:.3e}, first bad idx {idx}, got {got[bad][:4].tolist()} ref {ref[bad][:4].tolist()}")
    assert rel_l2 < 4e-3, f"{what}: rel-L2 {rel_l2:.3e}"
    return rel_l2


# ------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("tile", [128, 256, 257])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 200, 128), (1000, 768, 512), (512, 64, 256), (1, 512, 256),
                                   (1024, 1536, 1024), (130, 100, 64), (513, 3076, 128)])
def test_gemm_store(dev, tile, M, N, K):
    from reflectionflow_amd import _lib, ops
    pins.tile(tile)
    try:
        x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
        y = ops.linear(x, W, b)
        ref = x.float() @ W.float().t() + b.float()
        assert_close(y, ref, f"gemm store {M}x{N}x{K} tile{tile}")
        y2 = ops.linear(x, W, None)
        assert_close(y2, x.float() @ W.float().t(), "gemm no-bias")
    finally:
        pins.reset()


def test_gemm_transpose_detecting(dev):
    """A = I against an asymmetric W catches a swapped C layout (guide G9)."""
    from reflectionflow_amd import ops
    K = 128
    x = torch.eye(K, device=dev, dtype=BF)
    W = (torch.arange(K * 192, device=dev).reshape(192, K) % 251).to(BF) / 64
    y = ops.linear(x, W)
    assert torch.equal(y.float(), W.float().t())


@pytest.mark.parametrize("tile", [128, 256, 257])
def test_gemm_epilogues_and_segments(dev, tile):
    from reflectionflow_amd import _lib, ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, Group, Seg
    pins.tile(tile)
    try:
        M, N, K = 520, 384, 256
        x, W, b = rnd(M, K, dev=dev), rnd(N, K, dev=dev, scale=0.05), rnd(N, dev=dev)
        acc = x.float() @ W.float().t() + b.float()
        assert_close(ops.linear(x, W, b, epilogue=RF_EPI_GELU), F.gelu(acc, approximate="tanh"), "gelu")
        res, gate = rnd(M, N, dev=dev), rnd(N, dev=dev)
        assert_close(ops.linear(x, W, b, epilogue=RF_EPI_GATE_RES, residual=res, gate=gate),
                     res.float() + gate.float() * acc, "gate_res")
        # in place on the residual (how the engine uses it)
        res2 = res.clone()
        ops.linear(x, W, b, epilogue=RF_EPI_GATE_RES, residual=res2, gate=gate, out=res2)
        assert_close(res2, res.float() + gate.float() * acc, "gate_res in place")
        # residual == NULL means 0
        assert_close(ops.linear(x, W, b, epilogue=RF_EPI_GATE_RES, gate=gate), gate.float() * acc, "gate no residual")
        # three K segments: [x | x2 | t] . [W | W2 | B]^T   (concat input + LoRA)
        x2, W2 = rnd(M, 128, dev=dev), rnd(N, 128, dev=dev, scale=0.05)
        t, Bm = rnd(M, 64, dev=dev), rnd(N, 64, dev=dev, scale=0.05)
        y = ops.linear(x, W, b, extra=[Seg(x2, W2), Seg(t, Bm)])
        ref = acc + x2.float() @ W2.float().t() + t.float() @ Bm.float().t()
        assert_close(y, ref, "3 segments")
        # strided views: A and W as column slices of wider matrices (single-block proj_out)
        big_a, big_w = rnd(M, 640, dev=dev), rnd(N, 640, dev=dev, scale=0.05)
        y = ops.linear(big_a[:, 128:384], big_w[:, 128:384], b)
        assert_close(y, big_a[:, 128:384].float() @ big_w[:, 128:384].float().t() + b.float(), "strided operands")
        # grouped: three token groups with their own weights / outputs, one with an extra segment
        Ms = [96, 700, 130]
        xs = [rnd(m, K, dev=dev, seed=10 + i) for i, m in enumerate(Ms)]
        Ws = [rnd(N, K, dev=dev, scale=0.05, seed=20 + i) for i in range(3)]
        bs = [rnd(N, dev=dev, seed=30 + i) for i in range(3)]
        outs = [torch.empty(m, N, device=dev, dtype=BF) for m in Ms]
        tl, Bl = rnd(Ms[2], 64, dev=dev), rnd(N, 64, dev=dev, scale=0.05)
        groups = [Group([Seg(xs[0], Ws[0])], bias=bs[0], out=outs[0]),
                  Group([Seg(xs[1], Ws[1])], bias=bs[1], out=outs[1]),
                  Group([Seg(xs[2], Ws[2]), Seg(tl, Bl)], bias=bs[2], out=outs[2])]
        ops.gemm(groups, N)
        for i in range(3):
            ref = xs[i].float() @ Ws[i].float().t() + bs[i].float()
            if i == 2:
                ref = ref + tl.float() @ Bl.float().t()
            assert_close(outs[i], ref, f"group {i}")
    finally:
        pins.reset()


def vt_unpermute(vt, S):
    """[H, S_pad/64, 128, 64] key-permuted V^T tiles -> [H, S, 128]."""
    H, nt = vt.shape[0], vt.shape[1]
    kv = torch.arange(64, device=vt.device)
    pos = (kv & 51) | ((kv & 4) << 1) | ((kv & 8) >> 1)
    v = vt[:, :, :, pos]                      # [H, nt, 128, 64(kv)]
    return v.permute(0, 1, 3, 2).reshape(H, nt * 64, 128)[:, :S]


@pytest.mark.parametrize("tile", [128, 256, 257])
@pytest.mark.parametrize("St,Si,Sc", [(32, 64, 16), (30, 70, 13), (512, 256, 0)])
def test_gemm_qkv_epilogues(dev, tile, St, Si, Sc):
    from reflectionflow_amd import _lib, ops
    from reflectionflow_amd.ops import RF_EPI_QKV, RF_EPI_QKV_GELU, Group, Seg
    pins.tile(tile)
    try:
        H, D, MLP = 2, 256, 512
        S = St + Si + Sc
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        xt, xi = rnd(St, D, dev=dev, seed=1), rnd(Si, D, dev=dev, seed=2)
        Wt, Wi = rnd(3 * D, D, dev=dev, scale=0.05, seed=3), rnd(3 * D, D, dev=dev, scale=0.05, seed=4)
        bt, bi = rnd(3 * D, dev=dev, seed=5), rnd(3 * D, dev=dev, seed=6)
        groups = [Group([Seg(xt, Wt)], bias=bt, tok_offset=0), Group([Seg(xi, Wi)], bias=bi, tok_offset=St)]
        refs = [xt.float() @ Wt.float().t() + bt.float(), xi.float() @ Wi.float().t() + bi.float()]
        if Sc:
            xc = rnd(Sc, D, dev=dev, seed=7)
            groups.append(Group([Seg(xc, Wi)], bias=bi, tok_offset=St + Si))
            refs.append(xc.float() @ Wi.float().t() + bi.float())
        ops.gemm(groups, 3 * D, RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad)
        ref = torch.cat(refs, 0)                       # [S, 3D]
        rq, rk, rv = (ref[:, i * D:(i + 1) * D].reshape(S, H, 128).permute(1, 0, 2) for i in range(3))
        assert_close(q[:, :S], rq, "q head-major")
        assert_close(k[:, :S], rk, "k head-major")
        assert_close(vt_unpermute(vt, S), rv, "v^T tiles")
        # fused single-block flavour: [q|k|v|mlp] with GELU on the mlp columns
        q2, k2, vt2, _ = ops.alloc_attn_operands(H, S, dev)
        xm = torch.cat([xt, xi], 0)
        Wf, bf_ = rnd(3 * D + MLP, D, dev=dev, scale=0.05, seed=8), rnd(3 * D + MLP, dev=dev, seed=9)
        hid = torch.empty(S, MLP, device=dev, dtype=BF)
        groups = [Group([Seg(xm, Wf)], bias=bf_, out=hid[:St + Si], tok_offset=0)]
        if Sc:
            groups.append(Group([Seg(xc, Wf)], bias=bf_, out=hid[St + Si:], tok_offset=St + Si))
        ops.gemm(groups, 3 * D + MLP, RF_EPI_QKV_GELU, n_split=3 * D, q=q2, k=k2, vt=vt2, heads=H, s_pad=s_pad)
        xa = torch.cat([xm, xc], 0) if Sc else xm
        ref = xa.float() @ Wf.float().t() + bf_.float()
        rq, rk, rv = (ref[:, i * D:(i + 1) * D].reshape(S, H, 128).permute(1, 0, 2) for i in range(3))
        assert_close(q2[:, :S], rq, "fused q")
        assert_close(k2[:, :S], rk, "fused k")
        assert_close(vt_unpermute(vt2, S), rv, "fused v^T")
        assert_close(hid, F.gelu(ref[:, 3 * D:], approximate="tanh"), "fused mlp gelu")
    finally:
        pins.reset()


# ------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("rows,D", [(5, 256), (130, 3072), (64, 1024), (7, 2560)])
def test_layernorm_modulate(dev, rows, D):
    from reflectionflow_amd import ops
    x, sc, sh = rnd(rows, D, dev=dev, scale=2.0), rnd(D, dev=dev, scale=0.3), rnd(D, dev=dev, scale=0.3)
    y = ops.layernorm_modulate(x, sc, sh)
    ref = F.layer_norm(x.float(), (D,), eps=1e-6) * (1 + sc.float()) + sh.float()
    assert_close(y, ref, f"ln_mod {rows}x{D}")


@pytest.mark.parametrize("S,n_added", [(70, 30), (128, 0), (513, 512)])
def test_qk_rmsnorm_rope(dev, S, n_added):
    from oracle import flux_oracle as O
    from reflectionflow_amd import ops
    H = 3
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q[:, :S] = rnd(H, S, 128, dev=dev, seed=1)
    k[:, :S] = rnd(H, S, 128, dev=dev, seed=2)
    q0, k0 = q.clone(), k.clone()
    ws = [(1 + 0.1 * torch.randn(128, generator=torch.Generator().manual_seed(i))).to(dev).to(BF) for i in range(4)]
    ids = torch.cat([torch.zeros(n_added, 3), O.prepare_latent_image_ids(1, S - n_added) * 3], 0)
    ids[:, 1] = torch.arange(S) % 17
    cos, sin = O.FluxPosEmbed(10000, (16, 56, 56))(ids)
    cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()
    ops.qk_rmsnorm_rope(q, k, S, n_added, ws[0], ws[1], ws[2], ws[3], cos, sin)

    def ref_one(x, w_main, w_added):
        x = x[:, :S].float()[None]                      # [1,H,S,128]
        var = x.pow(2).mean(-1, keepdim=True)
        xn = x * torch.rsqrt(var + 1e-6)
        wsel = torch.where((torch.arange(S, device=dev) < n_added)[:, None], w_added.float()[None], w_main.float()[None])
        return O.apply_rotary_emb(xn * wsel[None, None], (cos, sin))[0]

    assert_close(q[:, :S], ref_one(q0, ws[0], ws[2]), "q norm+rope")
    assert_close(k[:, :S], ref_one(k0, ws[1], ws[3]), "k norm+rope")
    if s_pad > S:
        assert torch.equal(q[:, S:], q0[:, S:]), "padding rows must stay untouched"


def test_elementwise(dev):
    from reflectionflow_amd import ops
    for n in (64 * 4096, 1000, 7):
        x, v = rnd(n, dev=dev, seed=1), rnd(n, dev=dev, seed=2)
        ref = (x.float() + (-0.02) * v.float())
        ops.euler_step_(x, v, -0.02)
        assert_close(x, ref, f"euler n={n}", atol=1e-2)
        a, b = rnd(n, dev=dev, seed=3), rnd(n, dev=dev, seed=4)
        ref = a.float() + b.float()
        assert_close(ops.add_(a, b), ref, "add", atol=2e-2)
        s = rnd(n, dev=dev, seed=5, scale=3)
        assert_close(ops.silu(s), F.silu(s.float()), "silu", atol=2e-2)


# ------------------------------------------------------------------------------------- attention
def make_qkv(H, S, dev, seed=0, qscale=1.0):
    from reflectionflow_amd import ops
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    qf = rnd(H, S, 128, dev=dev, seed=seed + 1, scale=qscale)
    kf = rnd(H, S, 128, dev=dev, seed=seed + 2)
    vf = rnd(H, S, 128, dev=dev, seed=seed + 3)
    q[:, :S], k[:, :S] = qf, kf
    kv = torch.arange(s_pad, device=dev)
    pos = (kv & ~63) | (kv & 51) | ((kv & 4) << 1) | ((kv & 8) >> 1)
    vpad = torch.zeros(H, s_pad, 128, device=dev, dtype=BF)
    vpad[:, :S] = vf
    vperm = torch.empty_like(vpad)
    vperm[:, pos] = vpad                               # permuted position <- key
    vt.copy_(vperm.reshape(H, s_pad // 64, 64, 128).permute(0, 1, 3, 2))
    return q, k, vt, qf, kf, vf


def sdpa_ref(qf, kf, vf, mask=None):
    o = F.scaled_dot_product_attention(qf.float()[None], kf.float()[None], vf.float()[None], attn_mask=mask)
    return o[0].permute(1, 0, 2).reshape(qf.shape[1], -1)


@pytest.fixture(params=[0, 1], ids=["attn_v1", "attn_v2"])
def attn_impl(request, dev):
    """Pin the online-softmax kernel of the attention launches of a test (rf_attn_desc.kernel)."""
    from reflectionflow_amd import _lib, ops
    with ops.attn_kernel(_lib.RF_ATTN_ONLINE256 if request.param else _lib.RF_ATTN_ONLINE128):
        yield request.param


@pytest.mark.parametrize("S", [64, 100, 128, 333, 768, 1500])
def test_attention_plain(dev, attn_impl, S):
    from reflectionflow_amd import ops
    H = 2
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=S)
    o = ops.attention(q, k, vt, S)
    assert_close(o, sdpa_ref(qf, kf, vf), f"attention S={S}", atol=4e-3)


def test_attention_peaked_rows(dev, attn_impl):
    """Large logits: exercises the online-softmax rescale (running max jumps between tiles)."""
    from reflectionflow_amd import ops
    H, S = 1, 512
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=5, qscale=6.0)
    o = ops.attention(q, k, vt, S)
    assert_close(o, sdpa_ref(qf, kf, vf), "attention peaked", atol=1.5e-2)


@pytest.mark.parametrize("S,n_main", [(192, 128), (200, 150), (333, 300), (640, 512)])
@pytest.mark.parametrize("mode", [1, 2])
def test_attention_cond_modes(dev, attn_impl, S, n_main, mode):
    """block.py:106-122: additive log(c_factor) bias / block mask between main and condition tokens."""
    from reflectionflow_amd import ops
    H = 2
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=S + mode)
    bias = math.log(1.5)
    o = ops.attention(q, k, vt, S, n_main=n_main, mode=mode, cross_bias=bias)
    n = S - n_main
    if mode == 1:
        mask = torch.zeros(S, S, device=dev)
        mask[-n:, :-n] = bias
        mask[:-n, -n:] = bias
    else:
        mask = torch.ones(S, S, device=dev, dtype=torch.bool)
        mask[-n:, :-n] = False
        mask[:-n, -n:] = False
    assert_close(o, sdpa_ref(qf, kf, vf, mask), f"attention mode{mode} S={S}", atol=4e-3)


def test_errors_are_loud(dev):
    from reflectionflow_amd import ops
    with pytest.raises(ops.RFError):
        ops.linear(torch.zeros(4, 64, dtype=BF), torch.zeros(8, 64, dtype=BF))       # CPU tensors
    with pytest.raises(ops.RFError):
        ops.linear(torch.zeros(4, 72, dtype=BF, device=dev), torch.zeros(8, 72, dtype=BF, device=dev))  # K % 64
    with pytest.raises(ops.RFError):
        ops.linear(torch.zeros(4, 64, device=dev), torch.zeros(8, 64, device=dev))    # fp32


@pytest.mark.parametrize("tile", [128, 256, 257])
@pytest.mark.parametrize("St,Si,Sc", [(32, 64, 16), (30, 70, 13), (512, 320, 0)])
def test_gemm_qkv_fused_norm_rope(dev, tile, St, Si, Sc):
    """QKV GEMM with per-head RMSNorm (text rows: added-norm weights) + RoPE fused into the epilogue
    == plain QKV GEMM followed by the oracle's RMSNorm / apply_rotary_emb (block.py:38-41,60-67,74-78)."""
    from oracle import flux_oracle as O
    from reflectionflow_amd import _lib, ops
    from reflectionflow_amd.ops import RF_EPI_QKV, Group, Seg
    pins.tile(tile)
    try:
        H, D = 2, 256
        S = St + Si + Sc
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        xs = [rnd(m, D, dev=dev, seed=40 + i) for i, m in enumerate((St, Si, Sc)) if m]
        Wt, Wi = rnd(3 * D, D, dev=dev, scale=0.05, seed=3), rnd(3 * D, D, dev=dev, scale=0.05, seed=4)
        bt, bi = rnd(3 * D, dev=dev, seed=5), rnd(3 * D, dev=dev, seed=6)
        nw = [(1 + 0.1 * torch.randn(128, generator=torch.Generator().manual_seed(70 + i))).to(dev).to(BF) for i in range(4)]
        ids = torch.zeros(S, 3)
        ids[:, 1] = torch.arange(S) % 23
        ids[:, 2] = torch.arange(S) // 7
        cos, sin = O.FluxPosEmbed(10000, (16, 56, 56))(ids)
        cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()
        offs = [0, St, St + Si]
        groups, refs = [], []
        for gi, x in enumerate(xs):
            txt = gi == 0
            groups.append(Group([Seg(x, Wt if txt else Wi)], bias=bt if txt else bi, tok_offset=offs[gi],
                                norm_q=nw[2] if txt else nw[0], norm_k=nw[3] if txt else nw[1]))
            refs.append(x.float() @ (Wt if txt else Wi).float().t() + (bt if txt else bi).float())
        ops.gemm(groups, 3 * D, RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin))
        ref = torch.cat(refs, 0)
        is_txt = (torch.arange(S, device=dev) < St)[:, None]

        def nr(x, w_main, w_added):
            x = x.reshape(S, H, 128).permute(1, 0, 2)[None]                  # [1,H,S,128]
            xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
            w = torch.where(is_txt, w_added.float()[None], w_main.float()[None])
            return O.apply_rotary_emb(xn * w[None, None], (cos, sin))[0]

        assert_close(q[:, :S], nr(ref[:, :D], nw[0], nw[2]), "fused q")
        assert_close(k[:, :S], nr(ref[:, D:2 * D], nw[1], nw[3]), "fused k")
        assert_close(vt_unpermute(vt, S), ref[:, 2 * D:].reshape(S, H, 128).permute(1, 0, 2), "v^T (untouched by rope)")
    finally:
        pins.reset()


def test_attention_prescaled_q_matches_scaled_path(dev, attn_impl):
    """q carrying softmax_scale*log2(e) (as the engine's QKV epilogue writes it) + q_prescaled=1 gives the same
    attention as unscaled q with the scale applied inside the kernel (up to q's bf16 rounding)."""
    from reflectionflow_amd import ops
    H, S = 2, 640
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=77)
    o_ref = ops.attention(q, k, vt, S)
    q2 = q.clone()
    q2[:, :S] = (qf.float() * ops.QK_PRESCALE).to(BF)
    o_pre = ops.attention(q2, k, vt, S, q_prescaled=True)
    assert_close(o_pre, sdpa_ref(qf, kf, vf), "attention with prescaled q", atol=6e-3)
    assert_close(o_pre, o_ref.float(), "prescaled vs in-kernel scale", atol=6e-3)


# ------------------------------------------------------------------------------------- split-K (LoRA down)
@pytest.mark.parametrize("M,N,K,K2,ws_mib", [(1024, 128, 3072, 0, 32), (1024, 128, 3072, 12288, 32), (1000, 64, 2048, 0, 32),
                                             (517, 256, 1024, 512, 32), (1024, 128, 12288, 0, 2), (4608, 128, 3072, 0, 32)])
def test_gemm_splitk_lora_down(dev, M, N, K, K2, ws_mib):
    """Few-tile / long-K STORE GEMM (x . lora_A^T, one or two activation segments as in the single-block
    proj_out) sliced over K: same result as the unsplit launch up to one bf16 rounding of the fp32 sum, and
    bit-identical from run to run (slices are summed in index order)."""
    from reflectionflow_amd import ops
    x, A, b = rnd(M, K, dev=dev), rnd(N, K + K2, dev=dev, scale=0.05), rnd(N, dev=dev)
    segs = [ops.Seg(x, A[:, :K])]
    ref = x.float() @ A[:, :K].float().t() + b.float()
    if K2:
        x2 = rnd(M, K2, dev=dev, seed=7)
        segs.append(ops.Seg(x2, A[:, K:]))
        ref = ref + x2.float() @ A[:, K:].float().t()
    ws = torch.empty(ws_mib << 18, dtype=torch.float32, device=dev)
    outs = []
    for scratch in (ws, ws, False):
        ws[1024:].fill_(float("nan"))  # partial tiles must be fully overwritten before they are summed
        ws[:1024].zero_()             # (first 4 KiB = stream-K flags, zero by contract)
        y = torch.full((M, 256), 7.0, dtype=BF, device=dev)  # ldo > N as in the engine's LT buffer
        ops.gemm([ops.Group(segs, bias=b, out=y[:, :N])], N, ops.RF_EPI_STORE, splitk_ws=scratch)
        outs.append(y)
    assert_close(outs[0][:, :N], ref, "split-K vs fp32")
    assert torch.equal(outs[0], outs[1]), "split-K is not deterministic"
    assert (outs[0][:, N:] == 7.0).all(), "split-K reduce wrote outside its N columns"
    assert_close(outs[0][:, :N], outs[2][:, :N].float(), "split-K vs unsplit", rtol=8e-3)


# ------------------------------------------------------------------------------------- stream-K
@pytest.fixture
def force_sk(dev):
    from reflectionflow_amd import _lib
    lib = _lib.load()
    pins.tile(256)
    pins.sk(1)
    yield pins
    pins.reset()


@pytest.mark.parametrize("rows,N,K,K2", [((4608,), 3072, 3072, 0),            # cfg2 out-proj: 216 tiles on 256 CUs
                                         ((512, 4096, 1024), 3072, 1024, 128),  # cfg4: 264 tiles, LoRA segment on group 2
                                         ((1500,), 2048, 2048, 0),              # 48 tiles: every tile cut in ~5 pieces
                                         ((700, 300), 1000, 8192, 0),           # ragged M and N
                                         ((4608,), 9216, 512, 0)])              # 648 tiles: whole tiles + head + tail
def test_gemm_stream_k_gate_res(dev, force_sk, rows, N, K, K2):
    """Stream-K launch (K-tile iterations cut evenly over the CUs, partial tiles fixed up in index order) of the
    gated-residual GEMM: equals the one-tile-per-block launch up to fp32 summation order, the fp32 reference within
    bf16 rounding, and itself bit-for-bit from run to run -- with fresh inputs every round, so a stale partial
    tile or a missed flag cannot hide."""
    from reflectionflow_amd import ops
    lib = force_sk
    for rnd_i in range(3):
        groups, refs, outs = [], [], []
        for gi, M in enumerate(rows):
            x, W = rnd(M, K, dev=dev, seed=100 * rnd_i + gi), rnd(N, K, dev=dev, scale=0.05, seed=50 + 100 * rnd_i + gi)
            b, gate, res = rnd(N, dev=dev, seed=gi + 7), rnd(N, dev=dev, seed=gi + 9), rnd(M, N, dev=dev, seed=gi + 11)
            segs = [ops.Seg(x, W)]
            y = x.float() @ W.float().t()
            if K2 and gi == len(rows) - 1:
                t, B = rnd(M, K2, dev=dev, seed=gi + 13), rnd(N, K2, dev=dev, scale=0.05, seed=gi + 15)
                segs.append(ops.Seg(t, B))
                y = y + t.float() @ B.float().t()
            out = torch.empty(M, N, dtype=BF, device=dev)
            groups.append(ops.Group(segs, bias=b, out=out, residual=res, gate=gate))
            refs.append(res.float() + gate.float() * (y + b.float()))
            outs.append(out)
        got = []
        for mode in (1, 1, 0):
            pins.sk(mode)
            ops.gemm(groups, N, ops.RF_EPI_GATE_RES)
            assert pins.last_path() == (2 if mode else 0), "stream-K path was not (de)selected"
            got.append([o.clone() for o in outs])
        for gi in range(len(rows)):
            assert_close(got[0][gi], refs[gi], f"stream-K group {gi} vs fp32")
            assert torch.equal(got[0][gi], got[1][gi]), "stream-K is not deterministic"
            assert_close(got[0][gi], got[2][gi].float(), f"stream-K vs tile-per-block, group {gi}", rtol=8e-3)


def test_gemm_stream_k_qkv_fused(dev, force_sk):
    """Stream-K under the fused QKV + RMSNorm + RoPE epilogue (text / image / condition groups): q, k, V^T equal
    the tile-per-block launch within bf16 rounding of the fp32-sum reorder."""
    from reflectionflow_amd import ops
    lib = force_sk
    H, St, Si, Sc, K = 4, 512, 1024, 256, 3072
    D, S = H * 128, St + Si + Sc
    cos, sin = torch.rand(S, 128, device=dev), torch.rand(S, 128, device=dev)
    res = []
    for mode in (1, 0):
        pins.sk(mode)
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        groups = []
        for gi, (M, off) in enumerate([(St, 0), (Si, St), (Sc, St + Si)]):
            x, W, b = rnd(M, K, dev=dev, seed=gi), rnd(3 * D, K, dev=dev, scale=0.05, seed=10 + gi), rnd(3 * D, dev=dev, seed=20 + gi)
            nq, nk_ = rnd(128, dev=dev, seed=30 + gi) + 1.0, rnd(128, dev=dev, seed=40 + gi) + 1.0
            groups.append(ops.Group([ops.Seg(x, W)], bias=b, tok_offset=off, norm_q=nq, norm_k=nk_))
        ops.gemm(groups, 3 * D, ops.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE)
        assert pins.last_path() == (2 if mode else 0)
        res.append((q.clone(), k.clone(), vt.clone()))
    for a, b, name in zip(res[0], res[1], "q k vt".split()):
        assert_close(a, b.float(), f"stream-K {name}", rtol=8e-3)


# ------------------------------------------------------------------------------------- ping-pong loop edge cases
@pytest.mark.parametrize("tile", [256, 257])
@pytest.mark.parametrize("Ks", [(64,), (128,), (64, 64), (64, 128, 64), (192, 64), (64, 64, 64), (320, 64, 128), (1024, 128)])
def test_gemm_k_segment_boundaries(dev, tile, Ks):
    """K-tile counts 1..17 and every segment-boundary position of the staging cursors (the ping-pong loop stages tile
    t+1 and t+2 from two cursors that may sit in different segments; its prologue/tail switch between counted and
    draining vmcnt).  Ragged M and N, GELU epilogue, vs fp32; tile 257 = plain-loop twin, must agree bit for bit."""
    from reflectionflow_amd import _lib, ops
    lib = _lib.load()
    M, N = 700, 1032
    segs, ref = [], torch.zeros(M, N, device=dev)
    for i, K in enumerate(Ks):
        x, W = rnd(M, K, dev=dev, seed=3 * i + len(Ks)), rnd(N, K, dev=dev, scale=0.05, seed=3 * i + 1)
        segs.append(ops.Seg(x, W))
        ref += x.float() @ W.float().t()
    b = rnd(N, dev=dev, seed=99)
    ref = F.gelu(ref + b.float(), approximate="tanh")
    outs = {}
    for t in (tile, 256 + 257 - tile):
        pins.tile(t)
        try:
            y = torch.empty(M, N, dtype=BF, device=dev)
            ops.gemm([ops.Group(segs, bias=b, out=y)], N, ops.RF_EPI_GELU, splitk_ws=False)
            outs[t] = y
        finally:
            pins.tile(0)
    assert_close(outs[tile], ref, f"K segments {Ks}, tile {tile}")
    assert torch.equal(outs[256], outs[257]), "ping-pong and plain loops disagree"


def test_gemm_stream_k_graph_replay(dev, force_sk):
    """A stream-K launch leaves its flags as it found them (the single consumer of a partial resets its flag), so it
    can be captured in a hipGraph and replayed: three back-to-back launches, replayed twice with fresh inputs."""
    from reflectionflow_amd import ops
    lib = force_sk
    M, N, K = 1500, 2048, 2048
    x, W = torch.empty(M, K, dtype=BF, device=dev), rnd(N, K, dev=dev, scale=0.05, seed=2)
    outs = [torch.empty(M, N, dtype=BF, device=dev) for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.copy_(rnd(M, K, dev=dev, seed=1))
        ops.gemm([ops.Group([ops.Seg(x, W)], out=outs[0])], N)   # allocates this stream's scratch outside the capture
        assert pins.last_path() == 2
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for o in outs:
            ops.gemm([ops.Group([ops.Seg(x, W)], out=o)], N)
    for seed in (5, 6):
        xin = rnd(M, K, dev=dev, seed=seed)
        x.copy_(xin)
        graph.replay()
        torch.cuda.synchronize()
        ref = xin.float() @ W.float().t()
        for o in outs:
            assert_close(o, ref, f"graph replay, seed {seed}")
            assert torch.equal(o, outs[0])


def test_gemm_pingpong_cold_cache_stress(dev):
    """Hazard screen for the ping-pong loop's counted-vmcnt schedule: with operands that are cold in L2 / Infinity
    Cache (a 1.5 GiB flush between launches) the LDS-DMA pieces arrive late and out of step, which is when a read
    placed too early after its wait would show.  Fresh inputs every round; the plain loop (one drain + barrier per
    K-tile) is the bit-exact reference."""
    from reflectionflow_amd import _lib, ops
    lib = _lib.load()
    M, N, K = 4608, 3072, 6144
    flush = torch.empty(3 << 28, dtype=torch.float32, device=dev)  # 3 GiB > 256 MiB Infinity Cache
    try:
        for it in range(8):
            x, W = rnd(M, K, dev=dev, seed=200 + it), rnd(N, K, dev=dev, scale=0.05, seed=300 + it)
            b, gate, res = rnd(N, dev=dev, seed=it), rnd(N, dev=dev, seed=it + 50), rnd(M, N, dev=dev, seed=it + 70)
            outs = []
            for tile in (256, 257, 256):
                pins.tile(tile)
                flush.fill_(float(it))
                y = torch.empty(M, N, dtype=BF, device=dev)
                ops.gemm([ops.Group([ops.Seg(x, W)], bias=b, out=y, residual=res, gate=gate)], N, ops.RF_EPI_GATE_RES, splitk_ws=False)
                outs.append(y)
            assert torch.equal(outs[0], outs[1]), f"round {it}: ping-pong loop differs from the plain loop on cold operands"
            assert torch.equal(outs[0], outs[2]), f"round {it}: ping-pong loop is not reproducible on cold operands"
    finally:
        pins.tile(0)
        del flush
        torch.cuda.empty_cache()


@pytest.mark.parametrize("rows,N,K", [((512, 4096), 9216, 3072), ((4608,), 3072, 3072), ((300, 5000, 77), 1536, 256)])
def test_gemm_persistent_whole_tiles_equals_tile_per_block(dev, rows, N, K):
    """RF_SCHED_PERSISTENT: the persistent launch that walks whole 256x256 tiles (no stream-K region) computes
    every tile with the same loop and epilogue as the one-tile-per-block launch -> bit-identical outputs, also with
    several token groups and a partial last round."""
    from reflectionflow_amd import _lib, ops
    from reflectionflow_amd.ops import RF_EPI_GATE_RES, Group, Seg
    lib = _lib.load()
    xs = [rnd(m, K, dev=dev, seed=50 + i) for i, m in enumerate(rows)]
    Ws = [rnd(N, K, dev=dev, scale=0.05, seed=60 + i) for i in range(len(rows))]
    b, gate = rnd(N, dev=dev), rnd(N, dev=dev)
    res = [rnd(m, N, dev=dev, seed=70 + i) for i, m in enumerate(rows)]
    outs = {}
    for mode in (0, 2):
        pins.sk(mode)
        pins.tile(256)
        try:
            o = [r_.clone() for r_ in res]
            ops.gemm([Group([Seg(xs[i], Ws[i])], bias=b, out=o[i], residual=o[i], gate=gate) for i in range(len(rows))], N, RF_EPI_GATE_RES)
            assert pins.last_path() == mode
            outs[mode] = o
        finally:
            pins.sk(-1)
            pins.tile(0)
    for i in range(len(rows)):
        assert torch.equal(outs[0][i], outs[2][i]), f"group {i}: persistent != tile-per-block"
        assert_close(outs[2][i], res[i].float() + gate.float() * (xs[i].float() @ Ws[i].float().t() + b.float()), f"persistent group {i}")
