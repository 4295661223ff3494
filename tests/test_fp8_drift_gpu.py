"""cfg5 (fp8 weights + activations, e4m3, per-channel x per-token scales) accuracy evidence:

(a) 50-STEP DRIFT.  The same candidate denoised for 50 Euler steps three ways -- fp32 oracle (on the GPU, fp32 arithmetic),
    product bf16, product fp8 -- with the latents compared after EVERY step: rel-L2(fp8, fp32) and rel-L2(bf16, fp32) per step,
    on the 2 + 2 block hd128 model and on 2 + 2 blocks at FLUX.1-dev width (D = 3072, 24 heads).  The table is printed, written to
    gpurun_out/r03_fp8_drift.json (copied to profiles/) and bounded: the fp8 trajectory stays finite, its deviation grows no
    faster than the bound stated in the test, and it tracks the fp32 trajectory far better than an UNRELATED candidate does.
(b) PLUMBING, structurally: with the library's timing hook on, one fp8 block forward must launch exactly the expected number of
    fp8 GEMMs and quantisation passes and NO 256-tile bf16 GEMM for a stream that has no LoRA -- a stream left un-quantised or
    quantised twice changes these counts.
(c) PLUMBING, numerically, tighter than tests/test_w8_gpu.py: the emulation additionally rounds to bf16 the inputs the product
    stores as bf16 before quantising them (ATT, HID), so far fewer e4m3 codes flip between product and emulation.
The reference has no fp8 path: semantics are defined in include/rf_flux.h (rf_gemm_w8a8)."""
import json
import os

import pytest
import torch

from oracle import flux_oracle as O
from tests.test_model_gpu import BF, g, rel_l2, to_product

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from reflectionflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _trajectories(dev, om, St, gh, gw, T, seed):
    from reflectionflow_amd.flux.generate import generate
    cfgm = om.config
    gen = torch.Generator().manual_seed(seed)
    pe = torch.randn(1, St, cfgm["joint_attention_dim"], generator=gen)
    pooled = torch.randn(1, cfgm["pooled_projection_dim"], generator=gen)
    lat = O.get_noises([seed], gh * 16, gw * 16, dtype=torch.float32)[seed]
    pipe = to_product(om, dev)
    omg = om.to(dev)
    traj = {"fp32": [], "bf16": [], "fp8": []}
    O.denoise(omg, lat.to(dev), pe.to(dev), pooled.to(dev), T, image_hw=(gh, gw), conditioning_dtype=BF,
              callback=lambda i, t, x: traj["fp32"].append(x.float().cpu()))

    def run(key):
        def cb(p, i, t, kw):
            traj[key].append(kw["latents"].float().cpu())
            return {}
        generate(pipe, model_config={}, height=gh * 16, width=gw * 16, num_inference_steps=T, guidance_scale=3.5,
                 latents=g(lat, dev), prompt_embeds=g(pe, dev), pooled_prompt_embeds=g(pooled, dev), output_type="latent",
                 callback_on_step_end=cb)
    run("bf16")
    pipe.enable_fp8_weights(True)
    try:
        run("fp8")
    finally:
        pipe.enable_fp8_weights(False)
    # an unrelated candidate (other noise seed, same prompt): the yard-stick for "tracks the trajectory"
    lat2 = O.get_noises([seed + 1], gh * 16, gw * 16, dtype=torch.float32)[seed + 1]
    other = O.denoise(omg, lat2.to(dev), pe.to(dev), pooled.to(dev), T, image_hw=(gh, gw), conditioning_dtype=BF).float().cpu()
    return traj, other


@torch.no_grad()
@pytest.mark.parametrize("name", ["hd128_2+2", "flux_width_2+2"])
def test_fp8_fifty_step_drift(dev, name):
    T = 50
    if name == "hd128_2+2":
        from tests.golden_util import build
        om = build("hd128")
        St, gh, gw = 64, 8, 24                       # 64 + 192 = 256 joint tokens
    else:
        torch.manual_seed(0)
        om = O.FluxTransformer2DModel(num_layers=2, num_single_layers=2).float().eval()     # D = 3072, 24 heads, mlp 12288
        O.init_synthetic_(om, seed=2, std=0.02)
        St, gh, gw = 256, 16, 16                     # 256 + 256 = 512 joint tokens
    traj, other = _trajectories(dev, om, St, gh, gw, T, seed=11)
    assert len(traj["fp32"]) == len(traj["bf16"]) == len(traj["fp8"]) == T
    rows = []
    for i in range(T):
        rows.append({"step": i + 1, "fp8_vs_fp32": rel_l2(traj["fp8"][i], traj["fp32"][i]), "bf16_vs_fp32": rel_l2(traj["bf16"][i], traj["fp32"][i])})
    unrelated = rel_l2(other, traj["fp32"][-1])
    print(f"  {name}: 50-step drift (rel-L2 of the latents vs the fp32 trajectory); an unrelated candidate ends {unrelated:.3f} away")
    for r in rows:
        if r["step"] in (1, 2, 5, 10, 20, 30, 40, 50):
            print(f"    step {r['step']:2d}: fp8 {r['fp8_vs_fp32']:.3e}   bf16 {r['bf16_vs_fp32']:.3e}")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "r03_fp8_drift.json")
    blob = json.load(open(path)) if os.path.exists(path) else {}
    blob[name] = {"tokens": St + gh * gw, "steps": T, "unrelated_candidate_rel_l2": unrelated, "rows": rows}
    json.dump(blob, open(path, "w"), indent=1)
    assert all(torch.isfinite(x).all() for x in traj["fp8"])
    f8, b16 = rows[-1]["fp8_vs_fp32"], rows[-1]["bf16_vs_fp32"]
    # bounds (set from the measured table with ~2x margin, see profiles/r03_fp8_drift.md): the end-of-trajectory deviation of
    # W8A8-e4m3 stays an order of magnitude below an unrelated candidate and within a fixed multiple of bf16's own drift
    assert f8 < 0.25 * unrelated, (f8, unrelated)
    assert f8 < 30.0 * b16 + 0.02, (f8, b16)


@torch.no_grad()
def test_fp8_block_launch_census(dev):
    """One DoubleStream + one SingleStream block at FLUX width with dims.fp8: text and image streams (no LoRA) must run on the fp8
    GEMM (one grouped launch per stage) and be quantised exactly once per stage; the LoRA'd condition stream rides as a bf16 group
    of the same launches.  Counted with the library's timing hook."""
    from reflectionflow_amd import ops
    from reflectionflow_amd.flux.block import block_forward, single_block_forward
    from tests.golden_util import T as TT, build, load
    z = load("blocks_hd128")
    om = build("hd128", lora=True)
    pipe = to_product(om, dev)
    pm = pipe.transformer
    x, e, c, temb, ctemb = (TT(z[k]) for k in ("x", "e", "c", "temb", "ctemb"))
    rope = om.pos_embed(torch.cat([TT(z["txt_ids"]), TT(z["img_ids"])]))
    crope = om.pos_embed(TT(z["cond_ids"]))
    fg = lambda a: g(a, dev)  # noqa: E731
    cfg8 = {"union_cond_attn": True, "latent_lora": False, "fp8_weights": True}
    for use_c in (False, True):
        kw = dict(condition_latents=fg(c) if use_c else None, cond_rotary_emb=crope if use_c else None, cond_temb=fg(ctemb) if use_c else None)
        block_forward(pm.transformer_blocks[0], hidden_states=fg(x), encoder_hidden_states=fg(e), temb=fg(temb), image_rotary_emb=rope,
                      model_config=cfg8, **kw)                                     # warm (packing, workspaces)
        with ops.profile(256) as pr:
            block_forward(pm.transformer_blocks[0], hidden_states=fg(x), encoder_hidden_states=fg(e), temb=fg(temb), image_rotary_emb=rope,
                          model_config=cfg8, **kw)
        cl = pr.classes
        # 4 GEMM stages (qkv, out, ff1, ff2), each ONE mixed-precision launch
        assert cl["gemm_w8"]["launches"] == 4, cl
        # quantisation passes: LN+modulate (fp8 form) of txt and img for the attention and the FF halves = 4, quant_rows of ATT and of HID
        # for txt and img = 4
        assert cl["quant"]["launches"] == 8, cl
        assert "gemm_main" not in cl, cl                                            # no bf16 256-tile launch: every stage went through rf_gemm_w8a8
        # (gemm_small / rowop: the AdaLN linears of this per-op entry point -- SiLU + a skinny GEMM per stream -- and, with a
        #  condition, its bf16 LN+modulate and LoRA down-projections)
        n_small = cl.get("gemm_small", {"launches": 0})["launches"]
        assert n_small >= (2 if not use_c else 5), cl
        xs = torch.cat([e, x], 1)
        skw = dict(condition_latents=fg(c), cond_temb=fg(ctemb), cond_rotary_emb=crope) if use_c else {}
        single_block_forward(pm.single_transformer_blocks[0], fg(xs), fg(temb), image_rotary_emb=rope, model_config=cfg8, **skw)
        with ops.profile(256) as pr:
            single_block_forward(pm.single_transformer_blocks[0], fg(xs), fg(temb), image_rotary_emb=rope, model_config=cfg8, **skw)
        cl = pr.classes
        assert cl["gemm_w8"]["launches"] == 2 and cl["quant"]["launches"] == 2 and "gemm_main" not in cl, cl   # LN-mod fp8 + [ATT|HID] rows


@torch.no_grad()
def test_fp8_blocks_vs_storage_emulating_oracle(dev):
    """tests/test_w8_gpu.py bounds product-vs-emulation by the full quantisation noise because a few % of e4m3 codes flip between
    them.  Here the emulation also rounds to bf16 what the product stores as bf16 before quantising (ATT, HID): the flips that
    remain come from fp32 summation-order noise only, and the bound drops to 0.5 x cost_emu + 2 e_bf16 (measured ~0.25 x)."""
    import copy
    from reflectionflow_amd.flux.block import block_forward, single_block_forward
    from tests import w8_emulation as EM
    from tests.golden_util import T as TT, build, load
    from tests.test_model_gpu import bf16_oracle
    z = load("blocks_hd128")
    om = build("hd128", lora=True)
    ob = bf16_oracle(om)
    oe = EM.emulate_fp8(copy.deepcopy(om), 32, 64)
    pipe = to_product(om, dev)
    pm = pipe.transformer
    x, e, temb = (TT(z[k]) for k in ("x", "e", "temb"))
    rope = om.pos_embed(torch.cat([TT(z["txt_ids"]), TT(z["img_ids"])]))
    cfg = {"union_cond_attn": True, "latent_lora": False}
    cfg8 = dict(cfg, fp8_weights=True)
    a = lambda f: dict(hidden_states=f(x), encoder_hidden_states=f(e), temb=f(temb), image_rotary_emb=rope,  # noqa: E731
                       condition_latents=None, cond_temb=None)
    ref = O.block_forward(om.transformer_blocks[0], model_config=cfg, **a(lambda t: t))
    tb = O.block_forward(ob.transformer_blocks[0], model_config=cfg, **a(lambda t: t.to(BF)))
    hp = block_forward(pm.transformer_blocks[0], model_config=cfg8, **a(lambda t: g(t, dev)))
    res = {}
    for mode in (False, True):
        EM.FakeQuantLinear.emulate_storage = mode
        try:
            emu = O.block_forward(oe.transformer_blocks[0], model_config=cfg, **a(lambda t: t))
        finally:
            EM.FakeQuantLinear.emulate_storage = False
        for i, nm in enumerate(("txt", "img")):
            res[(mode, nm)] = (rel_l2(hp[i], emu[i]), rel_l2(emu[i], ref[i]), rel_l2(tb[i], ref[i]))
    for nm in ("txt", "img"):
        (e0, c0, t0), (e1, c1, t1) = res[(False, nm)], res[(True, nm)]
        print(f"  double/{nm}: product vs emulation {e0:.3e} -> {e1:.3e} with bf16-storage emulation (cost_emu {c1:.3e}, e_bf16 {t1:.3e})")
        assert e1 <= 0.5 * c1 + 2.0 * t1, (nm, e1, c1, t1)
