"""GPU parity of the product branches round 1 left untested (VERDICT r1 "What's weak"):
  * image-CFG second forward (reference generate.py:250-272) and condition_scale -> attn.c_factor through
    generate() (generate.py:86-90,312-316) -- the oracle's versions of both are pinned bit-exact to the reference
    by tests/golden/make_golden.py (fixtures final_cond_imgcfg / final_cond_cscale);
  * checkpoints and LoRA files loaded from disk (`from_pretrained`, `load_lora_weights(path)`) run the same
    kernels on the same bits as an in-memory model;
  * the attention kernels directly at the BASELINE sequence lengths (4608, 5632; 17920 on one head group);
  * the in-sequence timing hook (rf_profile_begin / rf_profile_end) bench.py's roofline is built on.
Tolerance: as tests/test_model_gpu.py (rel-L2 vs the fp32 oracle <= 2 x eager-bf16's own error + 2e-3)."""
import json
import os

import pytest
import torch

from oracle import flux_oracle as O
from tests.golden_util import GEOMS, SHAPES, T, build, load
from tests.test_model_gpu import BF, bf16_oracle, check, g, rel_l2, to_product
from tests.test_kernels_gpu import assert_close, make_qkv, sdpa_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@torch.no_grad()
@pytest.mark.parametrize("tag,kw", [("imgcfg", dict(image_guidance_scale=1.5)), ("cscale", dict(condition_scale=1.5))])
def test_generate_image_cfg_and_condition_scale(dev, tag, kw):
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    geom = "hd128"
    z = load(f"loop_{geom}")
    s = SHAPES[geom]
    H, W = s["gh"] * 16, s["gw"] * 16
    om = build(geom, lora=True)
    ob = bf16_oracle(om)
    pipe = to_product(om, dev)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    okw = dict(guidance_scale=3.5, condition_ids=T(z["cond_ids"]), model_config=cfg, image_hw=(s["gh"], s["gw"]), **kw)
    tb = O.denoise(ob, T(z["lat"]).to(BF), T(z["pe"]).to(BF), T(z["pooled"]).to(BF), 4,
                   condition_latents=T(z["cond"]).to(BF), **okw)
    ref = O.denoise(om, T(z["lat"]).clone(), T(z["pe"]), T(z["pooled"]), 4, condition_latents=T(z["cond"]),
                    conditioning_dtype=BF, **okw)
    # the fp32 oracle with fp32 conditioning scalars IS the reference-derived fixture (CPU test pins it); the
    # bf16-conditioning variant used as `ref` here differs from it only through (t*1000, g*1000)
    plain = O.denoise(om, T(z["lat"]).clone(), T(z["pe"]), T(z["pooled"]), 4, condition_latents=T(z["cond"]),
                      conditioning_dtype=BF, guidance_scale=3.5, condition_ids=T(z["cond_ids"]), model_config=cfg,
                      image_hw=(s["gh"], s["gw"]))
    assert rel_l2(ref, plain) > 1e-3, f"{tag}: the option must move the result"
    conds = [Condition("cot", tokens=g(T(z["cond"]), dev), ids=T(z["cond_ids"]).to(dev))]
    hp = generate(pipe, conditions=conds, model_config=cfg, default_lora=True, height=H, width=W, num_inference_steps=4,
                  guidance_scale=3.5, latents=g(T(z["lat"]), dev), prompt_embeds=g(T(z["pe"]), dev),
                  pooled_prompt_embeds=g(T(z["pooled"]), dev), output_type="latent", **kw).images
    e = check(hp, ref, tb, f"generate [{tag}]")
    print(f"  generate[{tag}] hip {e[0]:.3e}  torch-bf16 {e[1]:.3e}")
    # generate() must leave no c_factor behind (generate.py:312-316)
    assert not any(hasattr(m, "c_factor") for m in pipe.transformer.modules())


@torch.no_grad()
def test_files_on_disk_run_like_in_memory_model(dev, tmp_path):
    """from_pretrained(dir) + load_lora_weights(dir) (the calls of tts_reflectionflow.py:498-505) vs the same
    weights handed over in memory: identical latents, bit for bit."""
    from safetensors.torch import save_file
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.flux.pipeline import FluxPipeline, synthetic_lora_state_dict
    from reflectionflow_amd.flux import modules as M
    cfg = dict(GEOMS["hd128"])
    src = M.FluxTransformer2DModel(**cfg).to(BF)
    M.init_synthetic_(src, seed=4, std=0.05)
    root = tmp_path / "ckpt" / "transformer"
    os.makedirs(root)
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, open(root / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in src.state_dict().items()}, str(root / "diffusion_pytorch_model.safetensors"))
    lora = synthetic_lora_state_dict(src, r=8, seed=2)
    os.makedirs(tmp_path / "lora")
    save_file({k: v.contiguous() for k, v in lora.items()}, str(tmp_path / "lora" / "pytorch_lora_weights.safetensors"))
    p_disk = FluxPipeline.from_pretrained(str(tmp_path / "ckpt"), torch_dtype=BF).to(dev)
    p_disk.load_lora_weights(str(tmp_path / "lora"), adapter_name="reflection")
    p_mem = FluxPipeline(src).to(dev)
    p_mem.load_lora_weights(lora, adapter_name="reflection")
    gen = torch.Generator().manual_seed(3)
    lat, cond = torch.randn(1, 64, 64, generator=gen), torch.randn(1, 16, 64, generator=gen)
    pe, pooled = torch.randn(1, 32, cfg["joint_attention_dim"], generator=gen), torch.randn(1, cfg["pooled_projection_dim"], generator=gen)
    cond_ids = O.condition_ids_for(64)
    outs = []
    for p in (p_disk, p_mem):
        conds = [Condition("cot", tokens=g(cond, dev), ids=cond_ids.to(dev))]
        outs.append(generate(p, conditions=conds, model_config={"latent_lora": False}, default_lora=True, height=128,
                             width=128, num_inference_steps=3, latents=g(lat, dev), prompt_embeds=g(pe, dev),
                             pooled_prompt_embeds=g(pooled, dev), output_type="latent").images)
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])
    # and the LoRA is really applied: same run without it differs
    p_base = FluxPipeline.from_pretrained(str(tmp_path / "ckpt"), torch_dtype=BF).to(dev)
    conds = [Condition("cot", tokens=g(cond, dev), ids=cond_ids.to(dev))]
    base = generate(p_base, conditions=conds, model_config={}, default_lora=True, height=128, width=128,
                    num_inference_steps=3, latents=g(lat, dev), prompt_embeds=g(pe, dev),
                    pooled_prompt_embeds=g(pooled, dev), output_type="latent").images
    assert not torch.equal(base, outs[0])


@pytest.mark.parametrize("impl", [0, 1], ids=["attn_v1", "attn_v2"])
@pytest.mark.parametrize("S,n_main,mode", [(4608, 4608, 0), (5632, 4608, 0), (5632, 4608, 1), (17920, 16896, 0)])
def test_attention_at_baseline_sequence_lengths(dev, impl, S, n_main, mode):
    """rf_attention_fwd directly (not through a block) at the cfg2 / cfg4 / cfg5 joint sequence lengths."""
    import math
    from reflectionflow_amd import _lib, ops
    H = 8 if S < 10000 else 2
    with ops.attn_kernel(_lib.RF_ATTN_ONLINE256 if impl else _lib.RF_ATTN_ONLINE128):
        q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=S + mode)
        bias = math.log(1.5)
        o = ops.attention(q, k, vt, S, n_main=n_main, mode=mode, cross_bias=bias)
        o2 = ops.attention(q, k, vt, S, n_main=n_main, mode=mode, cross_bias=bias)
        mask = None
        if mode == 1:
            n = S - n_main
            mask = torch.zeros(S, S, device=dev)
            mask[-n:, :-n] = bias
            mask[:-n, -n:] = bias
        assert_close(o, sdpa_ref(qf, kf, vf, mask), f"attention S={S} mode{mode}", atol=2e-3)
        assert torch.equal(o, o2), "attention is not bit-stable run to run"


@torch.no_grad()
def test_profile_hook_counts_and_times_every_launch(dev):
    """rf_profile_begin/_end: one record per library launch, classes and algorithmic work as documented in
    include/rf_flux.h, durations positive and <= the wall time of the region."""
    import time
    from reflectionflow_amd import ops
    M_, N, K = 2048, 3072, 3072
    x = (torch.randn(M_, K, device=dev)).to(BF)
    W = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    q, k, vt, qf, kf, vf = make_qkv(2, 512, dev, seed=1)
    sc, sh = torch.zeros(K, device=dev, dtype=BF), torch.zeros(K, device=dev, dtype=BF)
    ops.linear(x, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ops.profile(64) as pr:
        for _ in range(3):
            ops.linear(x, W)                      # 96 tiles of 256^2 < 200 -> 128-tile kernel: "gemm_small"
        y = ops.linear(torch.cat([x, x, x]), W)   # 288 tiles -> "gemm_main"
        ops.attention(q, k, vt, 512)
        ops.layernorm_modulate(x, sc, sh)
        torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) * 1e6
    c = pr.classes
    assert pr.dropped == 0
    assert c["gemm_small"]["launches"] == 3 and c["gemm_main"]["launches"] == 1
    assert c["attention"]["launches"] == 1 and c["rowop"]["launches"] == 1
    assert c["gemm_small"]["work"] == pytest.approx(3 * 2.0 * M_ * N * K)
    assert c["gemm_main"]["work"] == pytest.approx(2.0 * 3 * M_ * N * K)
    assert c["attention"]["work"] == pytest.approx(4.0 * 512 * 512 * 128 * 2)
    assert c["rowop"]["work"] == pytest.approx(4.0 * M_ * K)
    tot = sum(v["us"] for v in c.values())
    assert all(v["us"] > 0 for v in c.values()) and tot <= wall_us      # event-to-event durations tile the region
    # a second profile can be opened after the first closed; nesting is refused loudly
    with ops.profile(4):
        with pytest.raises(ops.RFError):
            with ops.profile(4):
                pass
    assert torch.isfinite(y.float()).all()


def _prescaled_case(H, S, dev, seed, qscale=1.0):
    """q carries softmax_scale*log2(e) (what the QKV epilogue produces); reference on the rounded prescaled q."""
    import math
    import torch.nn.functional as F
    from reflectionflow_amd import ops
    q, k, vt, qf, kf, vf = make_qkv(H, S, dev, seed=seed, qscale=qscale)
    qp = (qf.float() * ops.QK_PRESCALE).to(BF)
    q[:, :S] = qp
    ref = F.scaled_dot_product_attention((qp.float() * math.log(2.0))[None], kf.float()[None], vf.float()[None], scale=1.0)
    ref = ref[0].permute(1, 0, 2).reshape(S, -1)
    bound = float((qp.float().norm(dim=-1).max() * kf.float().norm(dim=-1).max())) * 1.01
    return q, k, vt, ref, bound


@pytest.mark.parametrize("S,H", [(256, 3), (1024, 2), (4608, 8), (5632, 4), (17920, 2)])
def test_attention_bounded_score_kernel(dev, S, H):
    """v4 (rf_attention_fwd with a proven score bound: no running maximum) vs fp32 SDPA and vs the online-softmax
    kernel on the same operands; bit-stable run to run."""
    from reflectionflow_amd import _lib, ops
    q, k, vt, ref, bound = _prescaled_case(H, S, dev, seed=S)
    assert bound < 100
    lib = _lib.load()
    o4 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, scratch=False)
    # (5 = plain grid, 10 = the mixed-size grid of the same kernel, taken when its dispatch simulation predicts >= 4 %)
    assert lib.rf_debug_last_attn_path() in (5, 10), "AUTO with a proven bound must take the bounded kernel"
    o4b = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, scratch=False)
    o2 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, kernel=_lib.RF_ATTN_ONLINE256)
    assert lib.rf_debug_last_attn_path() == 2
    assert_close(o4, ref, f"attention v4 S={S}", atol=2e-3)
    assert torch.equal(o4, o4b), "v4 is not bit-stable run to run"
    e4, e2 = rel_l2(o4, ref), rel_l2(o2, ref)
    print(f"  S={S}: rel-L2 vs fp32 SDPA: bounded-score {e4:.3e}, online-softmax {e2:.3e}")
    assert e4 <= 1.5 * e2 + 1e-4


@pytest.mark.parametrize("S,H", [(2048, 8), (4608, 8), (5632, 24)])
def test_attention_split_launch_and_mfma_shapes(dev, S, H):
    """The bounded-score kernel three ways on the same operands: 16x16x32 MFMAs one workgroup per (head, query block)
    [path 5], the same as ONE persistent workgroup per CU over equal shares of the (block, key range) space with partial
    (O, l, m) added by a second launch [path 6: scratch attached, kernel pinned], and the 32x32x16 form [path 4] -- all
    against fp32 SDPA, and the split launch bit-stable run to run.  A split request without scratch must FAIL, not fall back."""
    from reflectionflow_amd import _lib, ops
    q, k, vt, ref, bound = _prescaled_case(H, S, dev, seed=S + H)
    lib = _lib.load()
    outs = {}
    for name, kern, path in (("plain", _lib.RF_ATTN_BOUNDED16, 5), ("split", _lib.RF_ATTN_BOUNDED16_SPLIT, 6),
                             ("split again", _lib.RF_ATTN_BOUNDED16_SPLIT, 6), ("32x32x16", _lib.RF_ATTN_BOUNDED32, 4)):
        outs[name] = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, kernel=kern)
        assert lib.rf_debug_last_attn_path() == path, (name, lib.rf_debug_last_attn_path())
    with pytest.raises(ops.RFError):
        ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, scratch=False, kernel=_lib.RF_ATTN_BOUNDED16_SPLIT)
    with pytest.raises(ops.RFError):   # a bounded kernel without a usable bound is refused as well
        ops.attention(q, k, vt, S, q_prescaled=True, score_bound=150.0, kernel=_lib.RF_ATTN_BOUNDED16)
    for name, o in outs.items():
        assert_close(o, ref, f"attention {name} S={S}", atol=2e-3)
    assert torch.equal(outs["split"], outs["split again"]), "the split launch is not bit-stable"
    assert (outs["split"].float() - outs["plain"].float()).abs().max() < 2e-3


def test_attention_bounded_score_extremes(dev):
    """The shift-free softmax at the edges of its contract: scores near +bound and near -bound in the same rows
    (|s| up to ~60 in the exp2 domain: P spans 2^-60 .. 2^60), and rows whose scores are ALL very negative."""
    from reflectionflow_amd import _lib, ops
    H, S = 2, 512
    q, k, vt, ref, bound = _prescaled_case(H, S, dev, seed=7, qscale=3.5)
    assert 30 < bound < 100, bound
    lib = _lib.load()
    o4 = ops.attention(q, k, vt, S, q_prescaled=True, score_bound=bound, kernel=_lib.RF_ATTN_BOUNDED16)
    assert_close(o4, ref, "attention v4 peaked", atol=1.5e-2)
    # all-negative rows: q = -c * k_mean direction -> every score << 0, P tiny but the row still normalises
    q2, k2, vt2, qf, kf, vf = make_qkv(1, 256, dev, seed=9)
    import math
    import torch.nn.functional as F
    kdir = kf.float().mean(1, keepdim=True)
    kf2 = (kdir + 0.05 * kf.float()).to(BF)
    k2[:, :256] = kf2
    qp = (-(kdir / kdir.norm()) * 5.0).expand(1, 256, 128).to(BF).contiguous()
    q2[:, :256] = qp
    ref2 = F.scaled_dot_product_attention((qp.float() * math.log(2.0))[None], kf2.float()[None], vf.float()[None], scale=1.0)
    ref2 = ref2[0].permute(1, 0, 2).reshape(256, -1)
    b2 = float(qp.float().norm(dim=-1).max() * kf2.float().norm(dim=-1).max()) * 1.01
    o = ops.attention(q2, k2, vt2, 256, q_prescaled=True, score_bound=b2, kernel=_lib.RF_ATTN_BOUNDED16)
    assert_close(o, ref2, "attention v4 all-negative rows", atol=4e-3)


def test_engine_passes_a_valid_score_bound(dev):
    """The bound the engine hands to rf_attention_fwd must really bound the scores the QKV epilogue produces."""
    from reflectionflow_amd import ops
    om = build("hd128")
    pipe = to_product(om, dev)
    blk = pipe.transformer.transformer_blocks[0]
    a = blk.attn
    bound = ops.qk_score_bound((a.norm_q.weight, a.norm_added_q.weight), (a.norm_k.weight, a.norm_added_k.weight))
    # worst case over random inputs of wildly different scale: the RMSNorm makes the bound input independent
    H, D, S = a.heads, a.heads * 128, 192
    gen = torch.Generator().manual_seed(3)
    worst = 0.0
    for scale in (1e-3, 1.0, 300.0):
        x = (torch.randn(S, D, generator=gen) * scale).to(dev).to(BF)
        w = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0)
        b = torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias], 0)
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        ids = torch.cat([torch.zeros(64, 3), O.prepare_latent_image_ids(8, 16)])
        cos, sin = (t.to(dev).contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
        ops.gemm([ops.Group([ops.Seg(x, w)], bias=b, tok_offset=0, norm_q=a.norm_q.weight, norm_k=a.norm_k.weight)],
                 3 * D, ops.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE)
        s = torch.einsum("hqd,hkd->hqk", q[:, :S].float(), k[:, :S].float())
        worst = max(worst, float(s.abs().max()))
    print(f"  engine bound {bound:.2f}, largest |score| seen {worst:.2f}")
    assert worst <= bound and bound < 100


@torch.no_grad()
def test_pil_output_and_reference_round_handoff_with_vae(dev, tmp_path):
    """With a VAE on the pipeline (PyTorch-ROCm modules, random-init) the calls either side of the loop are the
    reference's: generate(output_type="pil") decodes (generate.py:302-307), and the reflection rounds condition on
    decode -> 8-bit image -> resize(condition_size) -> VAE-encode (tts_reflectionflow.py:273-279, condition.py:96-132)
    instead of the latent-space stand-in."""
    import json
    from PIL import Image
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.flux.pipeline import FluxPipeline, synthetic_lora_state_dict
    from reflectionflow_amd.tts import runner, search
    small = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=256, pooled_projection_dim=64)
    vcfg = dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)
    pipe = FluxPipeline.synthetic(small, seed=0, device=dev, with_vae=True, vae_config=vcfg)
    pipe.set_progress_bar_config(disable=True)
    pipe.load_lora_weights(synthetic_lora_state_dict(pipe.transformer, r=8, seed=3), adapter_name="reflection")
    kw = dict(model_config={}, height=256, width=192, num_inference_steps=3, guidance_scale=3.5, prompt=["a red cube"])
    lat = generate(pipe, output_type="latent", generator=torch.Generator().manual_seed(4), **kw).images
    img = generate(pipe, output_type="pil", generator=torch.Generator().manual_seed(4), **kw).images[0]
    assert isinstance(img, Image.Image) and img.size == (192, 256)
    z = pipe._unpack_latents(lat, 256, 192, 8) / pipe.vae.config.scaling_factor + pipe.vae.config.shift_factor
    ref = pipe.image_processor.postprocess(pipe.vae.decode(z, return_dict=False)[0], output_type="pil")[0]
    assert img.tobytes() == ref.tobytes()
    # the round hand-off is the reference computation: a PIL condition of condition_size, delta [0, -size/16]
    c = runner.candidate_condition(pipe, lat, 256, 192, 128)
    assert isinstance(c, Condition) and c.tokens is None and c.condition.size == (128, 128) and list(c.position_delta) == [0, -8]
    torch.manual_seed(1)
    tok, ids, _ = c.encode(pipe)
    assert tok.shape == (1, 64, 64) and ids.shape == (64, 3) and float(ids[:, 2].min()) == -8.0
    # whole search with the VAE in the loop: PNGs like the reference, deterministic
    cfg = json.load(open(os.path.join(os.path.dirname(runner.__file__), "configs", "flux1_dev_mi355x.json")))
    cfg["pipeline_args"].update(height=256, width=256, condition_size=128, num_inference_steps=2)
    cfg["search_args"].update(search_branch=2, search_rounds=1)
    logs = []
    for rep in range(2):
        out = str(tmp_path / f"run{rep}")
        logs.append(runner.run_reflection_search(cfg, ["a red cube left of a blue ball"], out, pipe, search.Shard(0, 1)))
        files = sorted(os.listdir(os.path.join(out, "00000", "samples")))               # the plain "round 0" pool
        mid = sorted(os.listdir(os.path.join(out, "00000", "midimg")))                  # round 1, as the reference names it
        assert len(files) == 2 and len(mid) == 2 and all(f.endswith(".png") for f in files + mid)
        assert Image.open(os.path.join(out, "00000", "midimg", mid[0])).size == (256, 256)
    assert logs[0] == logs[1]
