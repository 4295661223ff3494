"""Round 5 GPU tests: bit-stability of launches that share a CU between workgroups, and the optimizer kernels.

The 128 x 128 GEMM form is the only MFMA kernel of the library of which TWO workgroups fit a CU (67.5 KiB LDS, 224 registers).  With
more than one tile per CU its fused RMSNorm + RoPE epilogue (the QKV GEMM of every block at small token counts: 512 text + 256 image
tokens = 256 x 256 pixels) returned a few wrong q / k values, different on every run -- found by the 50-step fast-path == per-step
test (tests/test_fullsize_gpu.py), traced to packed-fp32 VALU ops executing next to another workgroup's K loop
(csrc/gemm_bf16.hip, comment above gemm_bf16_kernel).  No parity test had caught it: ~200 of 2.4 M elements, rel-L2 < 1e-2."""
import ctypes as C

import pytest
import torch

from oracle import flux_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rnd(dev, g, *s, sc=1.0):
    return (torch.randn(*s, generator=g, device=dev) * sc).to(BF)


@pytest.mark.parametrize("St,Si", [(512, 256), (256, 256), (512, 400), (512, 128), (0, 640)])
def test_fused_qkv_epilogue_is_bit_stable_on_every_schedule(dev, St, Si):
    """rf_gemm_bf16 with RF_EPI_QKV + fused per-head RMSNorm + RoPE over (text | image) token groups, FLUX width (N = 9216, K = 3072,
    24 heads): 12 launches per schedule must give identical q / k / v^T, the 128^2 and 256^2 forms must agree with each other to bf16
    rounding in EVERY element (the defect showed as ~200 elements off by up to half the value range), and so must what AUTO picks."""
    from reflectionflow_amd import _lib as L, ops
    g = torch.Generator(device=dev).manual_seed(0)
    D, H = 3072, 24
    S = St + Si
    s_pad = (S + 63) // 64 * 64
    xt, xi = _rnd(dev, g, max(St, 1), D), _rnd(dev, g, Si, D)
    Wt, Wi = _rnd(dev, g, 3 * D, D, sc=0.02), _rnd(dev, g, 3 * D, D, sc=0.02)
    bt, bi = _rnd(dev, g, 3 * D), _rnd(dev, g, 3 * D)
    nq = (1 + 0.02 * torch.randn(128, device=dev, generator=g)).to(BF)
    nk = (1 + 0.02 * torch.randn(128, device=dev, generator=g)).to(BF)
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 32, torch.arange(S) % 32], 1).to(dev)
    cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
    lib = L.load()

    def run(sched):
        q = torch.zeros(H, s_pad, 128, dtype=BF, device=dev)
        k, vt = torch.zeros_like(q), torch.zeros_like(q)
        groups = []
        if St:
            groups.append(ops.Group([ops.Seg(xt, Wt)], bias=bt, norm_q=nq, norm_k=nk))
        groups.append(ops.Group([ops.Seg(xi, Wi)], bias=bi, norm_q=nq, norm_k=nk, tok_offset=St))
        d = ops.build_gemm_desc(groups, 3 * D, L.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE,
                                splitk_ws=ops.splitk_scratch(dev), schedule=sched)
        L.check(lib.rf_gemm_bf16(C.byref(d), ops.stream_ptr()), "rf_gemm_bf16")
        torch.cuda.synchronize()
        return q, k, vt
    first = {}
    for name, sched in (("tile128", L.RF_SCHED_TILE128), ("tile256", L.RF_SCHED_TILE256), ("auto", L.RF_SCHED_AUTO)):
        outs = [run(sched) for _ in range(12)]
        bad = [i for i in range(1, 12) if not all(torch.equal(a, b) for a, b in zip(outs[i], outs[0]))]
        assert not bad, f"{name}: launches {bad} of 12 differ from launch 0 at {St}+{Si} tokens"
        first[name] = outs[0]
    # the two forms sum K in different orders (32x32x16 vs 16x16x32 MFMAs): equal to bf16 rounding, not bit-equal
    for a, b in zip(first["tile128"], first["tile256"]):
        err = float((a.float() - b.float()).norm() / b.float().norm())
        assert err < 4e-3, f"128^2 and 256^2 forms disagree: rel-L2 {err:.2e}"
        assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -6 * float(b.float().abs().max()), "an element is off by more than bf16 rounding"
    for a, b in zip(first["auto"], first["tile256"]):            # AUTO: one of the two, or the stream-K cut of the 256^2 tiles
        assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -6 * float(b.float().abs().max())


@pytest.mark.parametrize("nd,ns", [(1, 0), (0, 1), (2, 2)])
def test_forward_is_bit_stable_at_small_token_counts(dev, nd, ns):
    """rf_flux_forward at 512 + 256 / 512 + 400 / 512 + 1024 tokens (256^2, 320^2, 512^2 images) on FLUX-width blocks: 8 runs, one
    result.  (tests/test_fullsize_gpu.py holds the same check for the 57-block model at cfg2's 512 + 4096.)"""
    import bench
    from reflectionflow_amd import engine as E
    pipe = bench.build_model(dev, dict(num_layers=nd, num_single_layers=ns), seed=0)
    eng = E.engine_for(pipe.transformer)
    for side in (16, 20, 32):
        St, Si = 512, side * side
        gen = torch.Generator().manual_seed(5)
        pe = torch.randn(1, St, 4096, generator=gen).to(dev).to(BF)
        pooled = torch.randn(1, 768, generator=gen).to(dev).to(BF)
        lat = torch.randn(1, Si, 64, generator=gen).to(dev).to(BF)
        t, gd = torch.tensor([0.5], device=dev), torch.tensor([4.0], device=dev)
        mod = eng.mod_table(eng.temb(t.to(BF) * 1000, gd.to(BF) * 1000, pooled))[0].contiguous()
        cos, sin = eng.rope_tables(torch.zeros(St, 3), O.prepare_latent_image_ids(side, side))
        outs = []
        for _ in range(8):
            outs.append(eng.forward(lat[0], pe[0], mod, cos, sin).clone())
            torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), f"{nd}+{ns} blocks at {St}+{Si} tokens: forward is not bit-stable"
    del pipe, eng
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------- optimizer kernels (csrc/optim.hip)
def _ulps_bf16(a, b):
    """largest distance between two bf16 tensors in units of the bf16 spacing at the larger magnitude.  Elements that both lie within
    2^-20 of the tensor's largest magnitude are skipped: there (results of a cancellation) the fp32 arithmetic's own last bit, which
    legitimately differs between two implementations of the same formula, is wider than the element's bf16 spacing."""
    af, bf = a.float(), b.float()
    mag = torch.maximum(af.abs(), bf.abs())
    floor = float(mag.max()) * 2.0 ** -20
    ulp = torch.exp2(torch.floor(torch.log2(mag.clamp_min(1e-30))) - 7)
    d = (af - bf).abs()
    return float(torch.where(d <= floor, torch.zeros_like(d), d / ulp).max())


def _params(dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    shapes = [(32, 3072), (3072, 32), (32, 64), (9216, 32), (32, 15360), (7, 24)]        # LoRA-factor shapes + one ragged tensor
    ps = [torch.nn.Parameter((torch.randn(s, generator=g, device=dev) * 0.05).to(BF)) for s in shapes]
    with torch.no_grad():
        for p in ps:
            if p.shape[-1] == 32:
                p.zero_()            # lora_B starts at zero (peft): the factors an optimizer with a tiny first step (Prodigy, d0 = 1e-6) can
    return ps, g                     # move at all in bf16 -- 0.05 + 1e-6 rounds back to 0.05


def test_lora_adamw_kernel_against_torch_adamw(dev):
    """rf_lora_adamw over the flat bucket vs torch.optim.AdamW on the same bf16 parameters (its fused AND its single-tensor
    implementation -- which differ from each other in the last bit): 8 steps, each step started from torch's own state so the
    comparison is per step, parameters and both moments within ONE bf16 ulp of the fused implementation; and against the PINNED fp32
    oracle (oracle/optim_oracle.adamw_step) with fp32 state, where only the parameter's final bf16 rounding separates the two."""
    from oracle import optim_oracle as OO
    from reflectionflow_amd.train.optim import FlatLoraBucket, LoraAdamW
    kw = dict(lr=2e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    ps_t, g = _params(dev)
    ps_h = [torch.nn.Parameter(p.detach().clone()) for p in ps_t]
    opt_t = torch.optim.AdamW(ps_t, fused=True, **kw)
    opt_h = LoraAdamW(FlatLoraBucket(ps_h), **kw)
    bucket = opt_h.bucket
    assert all(p.data_ptr() == bucket.param.data_ptr() + 2 * o for p, o in zip(ps_h, bucket.offsets))
    worst = 0.0
    for step in range(1, 9):
        grads = [(torch.randn(p.shape, generator=g, device=dev) * 0.1).to(BF) for p in ps_t]
        # start the HIP step from torch's state (per-step comparison; drift is measured separately below)
        with torch.no_grad():
            for ph, pt, o in zip(ps_h, ps_t, bucket.offsets):
                ph.copy_(pt)
                if step > 1:
                    opt_h.exp_avg[o:o + pt.numel()].copy_(opt_t.state[pt]["exp_avg"].reshape(-1))
                    opt_h.exp_avg_sq[o:o + pt.numel()].copy_(opt_t.state[pt]["exp_avg_sq"].reshape(-1))
        for pt, ph, gr in zip(ps_t, ps_h, grads):
            pt.grad = gr.clone()
            ph.grad.copy_(gr)
        opt_t.step()
        opt_h.step()
        for pt, ph, o in zip(ps_t, ps_h, bucket.offsets):
            n = pt.numel()
            u = max(_ulps_bf16(ph.detach(), pt.detach()), _ulps_bf16(opt_h.exp_avg[o:o + n].view_as(pt), opt_t.state[pt]["exp_avg"]),
                    _ulps_bf16(opt_h.exp_avg_sq[o:o + n].view_as(pt), opt_t.state[pt]["exp_avg_sq"]))
            worst = max(worst, u)
    print(f"  rf_lora_adamw vs torch.optim.AdamW(fused=True), bf16 state, 8 steps x 6 tensors: worst distance {worst:.2f} bf16 ulp")
    assert worst <= 1.0
    # fp32 state vs the pinned oracle, free-running for 6 steps
    ps2, g2 = _params(dev, seed=3)
    opt2 = LoraAdamW(ps2, state_dtype=torch.float32, **kw)
    ref = [p.detach().float().clone() for p in ps2]
    ms, vs = [torch.zeros_like(r) for r in ref], [torch.zeros_like(r) for r in ref]
    for step in range(1, 7):
        grads = [(torch.randn(p.shape, generator=g2, device=dev) * 0.1).to(BF) for p in ps2]
        for p, gr in zip(ps2, grads):
            p.grad.copy_(gr)
        # the oracle steps from the HIP path's bf16 parameters (what the next forward sees), in fp32
        ref = [p.detach().float().clone() for p in ps2]
        opt2.step()
        for r, gr, m, v in zip(ref, grads, ms, vs):
            OO.adamw_step(r, gr.float(), m, v, step, lr=kw["lr"], beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.05)
        for p, r, m, v, o in zip(ps2, ref, ms, vs, opt2.bucket.offsets):
            n = p.numel()
            assert _ulps_bf16(p.detach(), r.to(BF)) <= 1.0
            assert float((opt2.exp_avg[o:o + n].view_as(m) - m).abs().max()) <= 2e-6 * float(m.abs().max())
            assert float((opt2.exp_avg_sq[o:o + n].view_as(v) - v).abs().max()) <= 2e-6 * float(v.abs().max())
    # grad_scale = the all-reduce's averaging: a SUM of 4 identical ranks with grad_scale 1/4 is the single-rank step
    pa, ga = _params(dev, seed=5)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = LoraAdamW(pa, **kw), LoraAdamW(pb, **kw)
    ob.grad_scale = 0.25
    for p, q in zip(pa, pb):
        gr = (torch.randn(p.shape, generator=ga, device=dev) * 0.1).to(BF)
        p.grad.copy_(gr)
        q.grad.copy_(gr * 4)
    oa.step(), ob.step()
    assert torch.equal(oa.bucket.param, ob.bucket.param)


def test_lora_prodigy_kernels_against_the_oracle(dev):
    """rf_lora_prodigy (moments + two global sums, the new d on the device, the update) vs oracle/optim_oracle.prodigy_step on one
    flat tensor, with the reference's options (config.yaml:55-61: lr 1, use_bias_correction, safeguard_warmup, weight_decay 0.01),
    fp32 state: 160 steps of a noisy quadratic (d grows from 1e-6 by > 3 orders of magnitude); d_denom to 1e-5, d / d_hat / d_numerator to 1e-4 relative each step (fp32 partial sums vs fp64),
    parameters within one bf16 ulp per step; the all-zero step moves nothing; the run is bit-reproducible; bf16 state stays within
    bf16 rounding of the fp32-state run."""
    from oracle import optim_oracle as OO
    from reflectionflow_amd.train.optim import LoraProdigy
    kw = dict(lr=1.0, use_bias_correction=True, safeguard_warmup=True, weight_decay=0.01)

    def run(state_dtype, check):
        ps, g = _params(dev, seed=7)
        opt = LoraProdigy(ps, state_dtype=state_dtype, **kw)
        b = opt.bucket
        target = (torch.randn(b.numel, generator=g, device=dev) * 0.05).to(BF)
        x = b.param.detach().float().clone()
        st = OO.prodigy_init(x)
        opt.step()                                                    # gradients are all zero: the skipped step
        assert opt.d_state()["k"] == 0 and torch.equal(b.param, opt.p0)
        traj = []
        for step in range(160):
            gr = ((b.param.float() - target.float()) + 0.01 * torch.randn(b.numel, generator=g, device=dev)).to(BF)
            b.grad.copy_(gr)
            if check:
                x = b.param.detach().float().clone()                  # the oracle steps from the HIP path's bf16 parameters and from
                st["p0"] = opt.p0.float()                             # its scalars: a per-step comparison, no accumulated drift
                prev = opt.d_state()
                st.update(d=prev["d"], d_max=prev["d_max"], d_numerator=prev["d_numerator"], k=prev["k"])
            opt.step()
            ds = opt.d_state()
            traj.append(ds["d"])
            if check:
                OO.prodigy_step(x, gr.float(), st, **kw)
                # d_denom is a sum of magnitudes (fp32 block partials vs fp64: 1e-5); the numerator is a dot product WITH cancellation
                # (<g, x0 - x> over 1.1 M terms of both signs), so it and what derives from it get 1e-4
                for key, tol in (("d_denom", 1e-5), ("d_numerator", 1e-4), ("d_hat", 1e-4), ("d", 1e-4)):
                    assert abs(ds[key] - st[key]) <= tol * abs(st[key]) + 1e-30, (step, key, ds[key], st[key])
                assert ds["k"] == st["k"] == step + 1
                assert _ulps_bf16(b.param, x.to(BF)) <= 1.0, step
                assert float((opt.exp_avg - st["m"]).abs().max()) <= 1e-5 * float(st["m"].abs().max()) + 1e-30
                assert float((opt.s - st["s"]).abs().max()) <= 1e-5 * float(st["s"].abs().max()) + 1e-30
        return b.param.clone(), traj
    p32, t32 = run(torch.float32, True)
    print(f"  Prodigy on the device: d {t32[0]:.3e} -> {t32[-1]:.3e} over 160 steps (d0 1e-6)")
    assert t32[-1] > 1e3 * 1e-6 and all(b >= a for a, b in zip(t32, t32[1:]))
    again, t_again = run(torch.float32, False)
    assert torch.equal(p32, again) and t32 == t_again, "the three-launch step must be bit-reproducible (fixed-order sums)"
    p16, t16 = run(BF, False)
    assert abs(t16[-1] - t32[-1]) <= 0.05 * t32[-1]
    assert float((p16.float() - p32.float()).norm() / p32.float().norm()) < 2e-2


# ------------------------------------------------------------------------------------------------- co-residency stress (hardening)
def _stable(fn, n=10):
    outs = []
    for _ in range(n):
        o = fn()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (o if isinstance(o, (list, tuple)) else [o])])
    return [i for i in range(1, n) if not all(torch.equal(a, b) for a, b in zip(outs[i], outs[0]))]


def test_kernels_that_share_a_cu_are_bit_stable(dev):
    """The round-5 defect needed TWO workgroups on one CU (one in its K loop, one in its epilogue).  Every other MFMA kernel of the
    library whose workgroups can share a CU is launched here 10 times at a size with more workgroups than CUs and must give one
    result: the 128 x 128 GEMM on every epilogue (STORE / GELU / gated residual / QKV without and with the fused RoPE / QKV + GELU /
    split-K), the online-softmax attention kernels (masked, biased and ragged launches), the T5 attention + GEMMs through a
    2-layer encoder, the VAE's one-head attention through a decode, the token-axis skinny GEMM, and the attention backward's
    two-per-CU dK / dV form."""
    from reflectionflow_amd import _lib as L, ops
    from reflectionflow_amd.train import kernels as K
    g = torch.Generator(device=dev).manual_seed(0)
    bad = {}
    # --- 128 x 128 GEMM, 576 tiles
    M, N, Kd = 1536, 3072, 1024
    x, W, b = _rnd(dev, g, M, Kd), _rnd(dev, g, N, Kd, sc=0.03), _rnd(dev, g, N)
    gate, res = _rnd(dev, g, N), _rnd(dev, g, M, N)
    with ops.gemm_schedule(L.RF_SCHED_TILE128):
        bad["gemm128 store"] = _stable(lambda: ops.linear(x, W, b, splitk_ws=False))
        bad["gemm128 gelu"] = _stable(lambda: ops.linear(x, W, b, epilogue=L.RF_EPI_GELU, splitk_ws=False))

        def gr():
            y = res.clone()
            ops.gemm([ops.Group([ops.Seg(x, W)], bias=b, out=y, residual=y, gate=gate)], N, L.RF_EPI_GATE_RES, splitk_ws=False)
            return y
        bad["gemm128 gate_res"] = _stable(gr)
        H, S = 8, 1536
        D = H * 128
        xq, Wq, bq = _rnd(dev, g, S, D), _rnd(dev, g, 3 * D + 2048, D, sc=0.03), _rnd(dev, g, 3 * D + 2048)
        nw = (1 + 0.05 * torch.randn(128, device=dev, generator=g)).to(BF)
        ids = torch.stack([torch.zeros(S), torch.arange(S) // 32, torch.arange(S) % 32], 1).to(dev)
        cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))

        def qkv(rope, gelu):
            q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
            hid = torch.empty(S, 2048, dtype=BF, device=dev)
            n_ = 3 * D + (2048 if gelu else 0)
            ops.gemm([ops.Group([ops.Seg(xq, Wq[:n_])], bias=bq[:n_], out=hid if gelu else None, norm_q=nw, norm_k=nw)], n_,
                     L.RF_EPI_QKV_GELU if gelu else L.RF_EPI_QKV, n_split=3 * D if gelu else 0, q=q, k=k, vt=vt, heads=H, s_pad=s_pad,
                     rope=(cos, sin) if rope else None, q_scale=ops.QK_PRESCALE, splitk_ws=False)
            return [q, k, vt] + ([hid] if gelu else [])
        bad["gemm128 qkv"] = _stable(lambda: qkv(False, False))
        bad["gemm128 qkv+rope"] = _stable(lambda: qkv(True, False))
        bad["gemm128 qkv+rope+gelu"] = _stable(lambda: qkv(True, True))
    xs, As = _rnd(dev, g, 1024, 12288), _rnd(dev, g, 64, 12288, sc=0.03)
    bad["gemm128 split-K (LoRA down)"] = _stable(lambda: ops.lora_down(xs, As))
    # --- attention: online-softmax kernels (48 KiB / 80 KiB of LDS: several workgroups per CU), masked / biased / ragged
    for H, S, mode, nm in ((24, 1500, 0, None), (24, 2048, 1, 1024), (16, 2304, 2, 1280)):
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        q.normal_(generator=g), k.normal_(generator=g), vt.normal_(generator=g)
        for kern in (L.RF_ATTN_ONLINE128, L.RF_ATTN_ONLINE256):
            try:
                bad[f"attention kernel {kern} H={H} S={S} mode={mode}"] = _stable(
                    lambda: ops.attention(q, k, vt, S, n_main=nm, mode=mode, cross_bias=0.3 if mode else 0.0, kernel=kern))
            except ops.RFError:
                pass
    # --- token-axis skinny GEMM and the attention backward's two-per-CU form
    big, sk = _rnd(dev, g, 5632, 12288), _rnd(dev, g, 5632, 64)
    bad["gemm_tn_skinny"] = _stable(lambda: K.gemm_tn(big, sk))
    H, S = 24, 2560
    raw = _rnd(dev, g, S, 3 * H * 128, sc=1.5)
    w1 = torch.ones(128, device=dev).to(BF)
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 64, torch.arange(S) % 64], 1).to(dev)
    cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
    a = K.qkv_train_fwd(raw, H, 0, (w1, w1, None, None), cos, sin)
    out = ops.attention(a.q, a.k, a.vt, S, q_prescaled=True)
    dout = _rnd(dev, g, S, H * 128)
    bad["attention_bwd dq128 + dkv128x2"] = _stable(lambda: list(K.attention_bwd(a, out, dout, kernel=L.RF_ATTN_BWD_DQ_128 | L.RF_ATTN_BWD_DKV_128X2)))
    # --- T5 (attn64 + small GEMMs) and the VAE (one-head attention + conv GEMMs) through their encoders
    from oracle import text_oracle as TXO
    from reflectionflow_amd.flux.text_hip import HipT5Encoder
    sd = {k_: v.to(BF).float() for k_, v in TXO.synthetic_t5_state(300, 512, 64, 8, 1024, 2, 6).items()}
    enc = HipT5Encoder(sd, 8, dev)
    ids_t = torch.randint(0, 300, (8, 512), generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        bad["t5 encoder (8 x 512 tokens)"] = _stable(lambda: enc.encode(ids_t), n=6)
        from reflectionflow_amd.flux import vae as V
        from reflectionflow_amd.flux.vae_hip import HipVAE
        vae = V.init_synthetic_vae_(V.AutoencoderKL(block_out_channels=(64, 128, 512), norm_num_groups=32, mid_block_add_attention=True), seed=1).eval()
        hv = HipVAE(vae.to(dev).to(BF))
        z = torch.randn(2, 16, 32, 32, generator=torch.Generator().manual_seed(2)).to(dev).to(BF)
        bad["vae decode (2 x 32 x 32 latents)"] = _stable(lambda: hv.decode(z).sample, n=6)
    failed = {k_: v for k_, v in bad.items() if v}
    print(f"  {len(bad)} kernel configurations x 6-10 launches: all bit-stable" if not failed else f"  NOT bit-stable: {failed}")
    assert not failed, failed


@torch.no_grad()
def test_search_pipeline_folds_its_static_lora_by_default(dev, tmp_path):
    """runner.build_pipeline (what the tts entry points call): a `lora_path` is loaded AND folded into per-token-group weight copies by
    default (the search never changes its one LoRA; +11 GB at FLUX.1-dev size on a 288 GB part) -- `"merged_lora": false` keeps the
    K-segment form.  Both give the same conditioned latents to bf16 rounding of the merged sum, and image-only generation (no condition
    rows: LoRA gated off) is BIT-equal between them: the reference's enable_lora gating survives the fold."""
    import json
    import os
    from safetensors.torch import save_file
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.flux.pipeline import synthetic_lora_state_dict
    from reflectionflow_amd.tts import runner
    cfg = json.load(open(os.path.join(os.path.dirname(runner.__file__), "configs", "flux1_dev_mi355x.json")))
    probe = runner.build_pipeline(cfg, dev, synthetic=True, small=True)
    lora_file = str(tmp_path / "corrector.safetensors")
    save_file({k: v.contiguous() for k, v in synthetic_lora_state_dict(probe.transformer, r=8, seed=3).items()}, lora_file)
    outs = {}
    for merged in (True, False):
        c = json.loads(json.dumps(cfg))
        c["pipeline_args"]["lora_path"] = lora_file
        if not merged:
            c["pipeline_args"]["merged_lora"] = False
        pipe = runner.build_pipeline(c, dev, synthetic=True, small=True)
        assert bool(getattr(pipe.transformer, "_rf_merged_lora", False)) == merged
        g = torch.Generator().manual_seed(4)
        tr = pipe.transformer
        pe = torch.randn(1, 64, tr.config.joint_attention_dim, generator=g).to(dev).to(BF)
        pooled = torch.randn(1, tr.config.pooled_projection_dim, generator=g).to(dev).to(BF)
        lat = torch.randn(1, 256, 64, generator=g).to(dev).to(BF)
        cond = torch.randn(1, 64, 64, generator=g).to(dev).to(BF)
        mc = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
        kw = dict(model_config=mc, default_lora=True, height=256, width=256, num_inference_steps=3, guidance_scale=3.5, prompt_embeds=pe,
                  pooled_prompt_embeds=pooled, output_type="latent")
        conds = [Condition("cot", tokens=cond, ids=O.condition_ids_for(128).to(dev))]
        outs[merged] = (generate(pipe, conditions=conds, latents=lat.clone(), **kw).images, generate(pipe, conditions=None, latents=lat.clone(), **kw).images)
    rel = float((outs[True][0].float() - outs[False][0].float()).norm() / outs[False][0].float().norm())
    print(f"  merged vs K-segment LoRA through build_pipeline: rel-L2 {rel:.2e} (conditioned), image-only bit-equal {bool(torch.equal(outs[True][1], outs[False][1]))}")
    assert 0 < rel < 2e-2
    assert torch.equal(outs[True][1], outs[False][1])
