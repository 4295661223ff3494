"""Whole-model GPU parity at the BASELINE sizes: the FLUX.1-dev-shaped transformer (19 double + 38 single blocks,
11.9 B random-init parameters) with an r=32 FLUX-Corrector-shaped LoRA, through the product path vs the fp32 oracle
evaluated on the same GPU.

  cfg2  512 + 4096 tokens             forward: bitwise determinism, hipGraph replay, parity
  cfg4  512 + 4096 + 1024 cond        forward with LoRA on the condition rows (latent_lora False / True): inside the
                                      real 57-block sequence this exercises the split-K LoRA down-projections
                                      (K = 12288 / 15360) and the stream-K launches (264 / 792 / 1056 tiles)
  cfg5  512 + 16384 + 1024 cond       2-step generate() at 2048 x 2048 (S = 17920)

Tolerance (same calibrated rule as tests/test_model_gpu.py, stated here because the depth is 57 blocks):
    rel-L2(hip, fp32 oracle) <= 2 x rel-L2(oracle run in eager bf16, fp32 oracle) + 2e-3
with a hard ceiling of 6e-2 (the 3e-2 ceiling of the 2+2-block tests scaled for depth; the calibrated term is the
binding one -- both numbers are printed).  Conditioning scalars are exact in bf16 (forward tests) or formed in bf16
on both sides (`conditioning_dtype`, loop test) as the reference's bf16 pipeline does (transformer.py:95-98).
"""
import copy

import pytest
import torch

from oracle import flux_oracle as O
from tests.test_model_gpu import BF, rel_l2

pytestmark = pytest.mark.gpu
CEIL = 6e-2


def check_deep(hip, ref32, ref_bf16, what):
    assert torch.isfinite(hip.float()).all(), f"{what}: non-finite"
    e_hip, e_t = rel_l2(hip, ref32), rel_l2(ref_bf16, ref32)
    bound = min(2.0 * e_t + 2e-3, CEIL)
    print(f"  {what}: rel-L2 hip {e_hip:.3e}  eager-bf16 {e_t:.3e}  (bound {bound:.3e})")
    assert e_hip <= bound, f"{what}: rel-L2 hip {e_hip:.3e} vs torch-bf16 {e_t:.3e} (bound {bound:.3e})"
    return e_hip, e_t


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from reflectionflow_amd import _lib
    from reflectionflow_amd.flux.pipeline import synthetic_lora_state_dict
    _lib.load()
    dev = torch.device("cuda:0")
    pipe = bench.build_model(dev, {}, seed=0)
    with torch.no_grad():
        n = pipe.load_lora_weights(synthetic_lora_state_dict(pipe.transformer, r=32, seed=1))
        assert n == 1 + 19 * 6 + 38 * 6
        with torch.device(dev):
            om = O.FluxTransformer2DModel().float().eval()
        O.inject_lora(om, r=32, alpha=32.0)
        om = om.to(dev)
        missing, unexpected = om.load_state_dict({k: v.float() for k, v in pipe.transformer.state_dict().items()}, strict=False)
        assert not missing and not unexpected, (missing[:4], unexpected[:4])
        ob = copy.deepcopy(om).to(BF)
    yield dev, pipe, om, ob
    del om, ob, pipe
    torch.cuda.empty_cache()


def _inputs(dev, St, Si, Sc, seed=1):
    gen = torch.Generator().manual_seed(seed)
    pe = torch.randn(1, St, 4096, generator=gen).to(dev).to(BF)
    pooled = torch.randn(1, 768, generator=gen).to(dev).to(BF)
    lat = torch.randn(1, Si, 64, generator=gen).to(dev).to(BF)
    cond = torch.randn(1, Sc, 64, generator=gen).to(dev).to(BF) if Sc else None
    side = int(Si ** 0.5)
    return pe, pooled, lat, cond, O.prepare_latent_image_ids(side, side), torch.zeros(St, 3)


@torch.no_grad()
def test_cfg2_forward_deterministic_graph_replay_and_parity(full):
    """512 text + 4096 image tokens: HIP forward vs the fp32 oracle, bitwise run-to-run determinism of the kernel
    sequence (the test that exposes a missing LDS-DMA wait: such races only show with cold caches inside the real
    sequence), and hipGraph capture + replay reproducing the eager result bit for bit."""
    from reflectionflow_amd import engine as E
    dev, pipe, om, ob = full
    tr = pipe.transformer
    eng = E.engine_for(tr)
    St, Si = 512, 4096
    pe, pooled, lat, _, img_ids, txt_ids = _inputs(dev, St, Si, 0)
    t, gd = torch.tensor([0.5], device=dev), torch.tensor([4.0], device=dev)      # exact in bf16 after x1000
    temb = eng.temb(t.to(BF) * 1000, gd.to(BF) * 1000, pooled)
    mod = eng.mod_table(temb)[0].contiguous()
    cos, sin = eng.rope_tables(txt_ids, img_ids)
    outs = [eng.forward(lat[0], pe[0], mod, cos, sin).clone() for _ in range(6)]
    torch.cuda.synchronize()
    assert all(torch.isfinite(o.float()).all() for o in outs)
    assert all(torch.equal(o, outs[0]) for o in outs[1:]), "kernel sequence is not deterministic run-to-run"
    gout = torch.empty_like(lat[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.forward(lat[0], pe[0], mod, cos, sin, out=gout)     # warm the per-stream workspace outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        eng.forward(lat[0], pe[0], mod, cos, sin, out=gout)
    for _ in range(2):
        gout.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, outs[0]), "hipGraph replay of rf_flux_forward differs from the eager launch sequence"
    del graph
    kw = lambda f: dict(hidden_states=f(lat), encoder_hidden_states=f(pe), pooled_projections=f(pooled),  # noqa: E731
                        timestep=t, guidance=gd, img_ids=img_ids, txt_ids=txt_ids, return_dict=False)
    ref = O.tranformer_forward(om, None, None, None, model_config={}, **kw(lambda a: a.float()))[0][0]
    tb = O.tranformer_forward(ob, None, None, None, model_config={}, **kw(lambda a: a))[0][0]
    check_deep(outs[0], ref, tb, "cfg2 forward (57 blocks, S=4608)")


@torch.no_grad()
@pytest.mark.parametrize("latent_lora", [False, True])
def test_cfg4_forward_with_lora_on_condition_rows(full, latent_lora):
    from reflectionflow_amd import _lib
    from reflectionflow_amd.flux.transformer import tranformer_forward
    dev, pipe, om, ob = full
    St, Si, Sc = 512, 4096, 1024
    pe, pooled, lat, cond, img_ids, txt_ids = _inputs(dev, St, Si, Sc, seed=2)
    cond_ids = O.condition_ids_for(512)
    t, gd = torch.tensor([0.5], device=dev), torch.tensor([4.0], device=dev)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": latent_lora}
    kw = lambda f: dict(hidden_states=f(lat), encoder_hidden_states=f(pe), pooled_projections=f(pooled),  # noqa: E731
                        timestep=t, guidance=gd, img_ids=img_ids, txt_ids=txt_ids, return_dict=False)
    hp = tranformer_forward(pipe.transformer, cond, cond_ids.to(dev), None, model_config=cfg, **kw(lambda a: a))[0]
    hp2 = tranformer_forward(pipe.transformer, cond, cond_ids.to(dev), None, model_config=cfg, **kw(lambda a: a))[0]
    assert torch.equal(hp, hp2), "cfg4 forward (stream-K + split-K inside the sequence) is not bit-stable"
    ref = O.tranformer_forward(om, cond.float(), cond_ids, None, model_config=cfg, **kw(lambda a: a.float()))[0]
    tb = O.tranformer_forward(ob, cond, cond_ids, None, model_config=cfg, **kw(lambda a: a))[0]
    check_deep(hp, ref, tb, f"cfg4 forward (S=5632, r=32 LoRA, latent_lora={latent_lora})")
    # LoRA must matter on this model: the same forward without the condition stream's LoRA-carrying inputs differs
    if not latent_lora:
        base = tranformer_forward(pipe.transformer, None, None, None, model_config=cfg, **kw(lambda a: a))[0]
        assert rel_l2(hp, base) > 1e-3


@torch.no_grad()
@pytest.mark.parametrize("use_cond", [False, True])
def test_fast_denoise_is_bit_equal_to_the_per_step_path_at_50_steps(full, use_cond):
    """bench.py times the fast path (all T modulation tables up front, T steps in one C call / hipGraph); the
    parity suite pins the per-step path (tranformer_forward + scheduler.step per step, transformer.py:95-114 forming the time
    embedding at M = B).  Up to round 4 the two differed in the last bit of temb at T = 50 (PyTorch's linears at M = 50 vs M = 1) and
    by 6.7e-3 after 50 steps.  time_text_embed now runs on the row-invariant rf_gemm_bf16 in both: the 57-block FLUX-width model,
    T = 50, with and without a condition stream, must give IDENTICAL latents, and temb / modulation rows must not depend on M."""
    from reflectionflow_amd import engine as E
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    dev, pipe, om, ob = full
    eng = E.engine_for(pipe.transformer)
    St, Si, Sc, T_ = 512, 256, 64, 50                                             # 256 x 256 latents, 128 x 128 condition
    pe, pooled, lat, cond, _, _ = _inputs(dev, St, Si, Sc, seed=5)
    pipe.scheduler.set_timesteps(T_, device=dev, mu=0.8)
    ts = (pipe.scheduler.timesteps.to(dev).to(BF) / 1000).to(BF) * 1000
    gd = torch.full((T_,), 3.5, device=dev).to(BF) * 1000
    te_all = eng.temb(ts, gd, pooled.expand(T_, -1))
    te_one = torch.cat([eng.temb(ts[i:i + 1], gd[i:i + 1], pooled) for i in range(T_)])
    assert torch.equal(te_all, te_one), "time embedding depends on how many rows are evaluated at once"
    assert torch.equal(eng.mod_table(te_all)[7:8], eng.mod_table(te_all[7:8]))
    # ... and it is the reference's arithmetic: against the fp32 modules of the product model on the same inputs
    te32 = copy.deepcopy(pipe.transformer.time_text_embed).float()(ts.float(), gd.float(), pooled.float().expand(T_, -1))
    te_bf = pipe.transformer.time_text_embed(ts, gd, pooled.expand(T_, -1))
    e_hip, e_t = rel_l2(te_all, te32), rel_l2(te_bf, te32)
    print(f"  time_text_embed T=50 rows: rel-L2 hip {e_hip:.3e}  torch-bf16 modules {e_t:.3e}")
    assert e_hip <= 2.0 * e_t + 2e-3
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    conds = [Condition("cot", tokens=cond, ids=O.condition_ids_for(128).to(dev))] if use_cond else None
    common = dict(conditions=conds, model_config=cfg, default_lora=True, height=256, width=256, num_inference_steps=T_,
                  guidance_scale=3.5, prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent")
    fast = generate(pipe, latents=lat.clone(), **common).images
    n = []
    slow = generate(pipe, latents=lat.clone(), callback_on_step_end=lambda p, i, t, kw: n.append(i) or {}, **common).images
    assert len(n) == T_
    assert torch.isfinite(fast.float()).all()
    assert torch.equal(fast, slow), f"fast vs per-step path at T=50 differ: rel-L2 {rel_l2(fast, slow):.3e}"


@torch.no_grad()
def test_cfg5_generate_two_steps_2048(full):
    """BASELINE cfg5 geometry in bf16 (fp8 weights: tests/test_w8_gpu.py): 2048 x 2048 latents (16384 tokens) + 512^2
    condition (1024 tokens) + 512 text = 17920 joint tokens, 2 Euler steps through generate()."""
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    dev, pipe, om, ob = full
    St, Si, Sc = 512, 16384, 1024
    pe, pooled, lat, cond, _, _ = _inputs(dev, St, Si, Sc, seed=3)
    cond_ids = O.condition_ids_for(512)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    hp = generate(pipe, conditions=[Condition("cot", tokens=cond, ids=cond_ids.to(dev))], model_config=cfg,
                  default_lora=True, height=2048, width=2048, num_inference_steps=2, guidance_scale=3.5, latents=lat,
                  prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images
    assert hp.shape == (1, Si, 64)
    okw = dict(guidance_scale=3.5, condition_ids=cond_ids, model_config=cfg, image_hw=(128, 128))
    ref = O.denoise(om, lat.float(), pe.float(), pooled.float(), 2, condition_latents=cond.float(),
                    conditioning_dtype=BF, **okw)
    tb = O.denoise(ob, lat, pe, pooled, 2, condition_latents=cond, **okw)
    check_deep(hp, ref, tb, "cfg5 2-step generate (S=17920)")


@torch.no_grad()
def test_cfg5_generate_two_steps_2048_fp8_weights(full):
    """BASELINE cfg5 as named: 2048 x 2048 + condition with fp8 (e4m3) weights AND activations on the text / image
    streams' big GEMMs (rf_gemm_w8a8; the LoRA'd condition rows stay bf16), 2 Euler steps through generate().
    The reference has no fp8 semantics: the plumbing is checked against the fp32 oracle under the fp8 EMULATION of
    tests/w8_emulation.py (same quantisation points, fp32 arithmetic) with the statistical bounds explained in
    tests/test_w8_gpu.py; the kernels' arithmetic itself is pinned there on identical quantised operands."""
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from tests import w8_emulation as EM
    dev, pipe, om, ob = full
    St, Si, Sc = 512, 16384, 1024
    pe, pooled, lat, cond, _, _ = _inputs(dev, St, Si, Sc, seed=3)
    cond_ids = O.condition_ids_for(512)
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    gkw = dict(model_config=cfg, default_lora=True, height=2048, width=2048, num_inference_steps=2, guidance_scale=3.5,
               latents=lat, prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent")
    pipe.enable_fp8_weights(True)
    try:
        hp = generate(pipe, conditions=[Condition("cot", tokens=cond, ids=cond_ids.to(dev))], **gkw).images
    finally:
        pipe.enable_fp8_weights(False)
    hp16 = generate(pipe, conditions=[Condition("cot", tokens=cond, ids=cond_ids.to(dev))], **gkw).images
    okw = dict(guidance_scale=3.5, condition_ids=cond_ids, model_config=cfg, image_hw=(128, 128))
    ref = O.denoise(om, lat.float(), pe.float(), pooled.float(), 2, condition_latents=cond.float(), conditioning_dtype=BF, **okw)
    tb = O.denoise(ob, lat, pe, pooled, 2, condition_latents=cond, **okw)
    EM.emulate_fp8(om, St, Si)
    try:
        emu = O.denoise(om, lat.float(), pe.float(), pooled.float(), 2, condition_latents=cond.float(), conditioning_dtype=BF, **okw)
    finally:
        EM.remove_emulation(om)
    e_impl, e_t, e_cost, e16, cost_emu = rel_l2(hp, emu), rel_l2(tb, ref), rel_l2(hp, ref), rel_l2(hp16, ref), rel_l2(emu, ref)
    print(f"  cfg5 fp8 2-step generate: product vs emulated-fp8 oracle {e_impl:.3e}, product vs fp32 {e_cost:.3e}, emulation vs fp32 "
          f"{cost_emu:.3e}, eager-bf16 vs fp32 {e_t:.3e} (bf16 product vs fp32: {e16:.3e})")
    assert torch.isfinite(hp.float()).all()
    # bounds: tests/test_w8_gpu.py "What can be asserted here"
    assert e_impl <= 1.0 * cost_emu + 2.0 * e_t, f"product vs emulated fp8 oracle {e_impl:.3e} (cost_emu {cost_emu:.3e})"
    assert e_cost <= 1.5 * cost_emu + 2.0 * e_t, f"product deviates {e_cost:.3e} from fp32 (cost_emu {cost_emu:.3e})"
    assert not torch.equal(hp, hp16)


# ------------------------------------------------------------------------------------------------------------------------------
# The headline configurations end to end: the 50-step latent of the 57-block model vs the fp32 oracle.
# north_star: "latents match the reference PyTorch path on fixed seed / schedule within a stated fp tolerance" -- stated here:
#     rel-L2(hip latent, fp32 oracle latent) <= 2 x rel-L2(oracle in eager bf16, fp32 oracle) + 2e-3     at the final step,
# the same calibrated rule as every other level of the suite, with no separate ceiling: fifty Euler steps through 57 random-init
# blocks amplify rounding, the eager-bf16 trajectory measures by how much, and both per-step curves are printed and written to
# gpurun_out/r06_headline_parity.json.  The fp32 oracle runs on the same GPU with the conditioning scalars formed in bf16
# (`conditioning_dtype`, as the reference's bf16 pipeline forms them, transformer.py:95-98); seeds through the get_noises protocol.
def _fifty_step_table(what, hip_steps, ref_steps, tb_steps, extra=None):
    import json
    import os
    rows = [{"step": i + 1, "hip_vs_fp32": rel_l2(h, r), "eager_bf16_vs_fp32": rel_l2(b, r)} for i, (h, r, b) in enumerate(zip(hip_steps, ref_steps, tb_steps))]
    print(f"  {what}: per-step rel-L2 of the latents vs the fp32 oracle trajectory")
    for r in rows:
        if r["step"] in (1, 2, 5, 10, 20, 30, 40, 45, 50):
            print(f"    step {r['step']:2d}: hip {r['hip_vs_fp32']:.3e}   eager-bf16 {r['eager_bf16_vs_fp32']:.3e}")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, "r06_headline_parity.json")
    blob = json.load(open(path)) if os.path.exists(path) else {}
    blob[what] = dict(rows=rows, **(extra or {}))
    json.dump(blob, open(path, "w"), indent=1)
    return rows


def _check_final(what, hip, ref, tb):
    assert torch.isfinite(hip.float()).all(), f"{what}: non-finite"
    e_hip, e_t = rel_l2(hip, ref), rel_l2(tb, ref)
    bound = 2.0 * e_t + 2e-3
    print(f"  {what}: final latent rel-L2 hip {e_hip:.3e}  eager-bf16 {e_t:.3e}  (bound {bound:.3e})")
    assert e_hip <= bound, f"{what}: rel-L2 hip {e_hip:.3e} vs eager-bf16 {e_t:.3e} (bound {bound:.3e})"
    return e_hip, e_t


@torch.no_grad()
def test_cfg2_fifty_step_latent_vs_fp32_oracle(full):
    """BASELINE cfg2 as benched: 19 + 38 blocks, 1024 x 1024, T = 50, guidance 3.5, seeded noise from `get_noises`; the FAST path
    (`generate(output_type="latent")`: rf_flux_denoise / hipGraph -- what bench.py times) vs `O.denoise` in fp32."""
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.tts.utils import get_noises
    dev, pipe, om, ob = full
    T_, seed = 50, 1234
    pe, pooled, _, _, _, _ = _inputs(dev, 512, 4096, 0, seed=7)
    noise = get_noises(2 ** 31 - 1, 1, 1024, 1024, device=dev, dtype=BF, seeds=[seed])[seed]
    assert noise.shape == (1, 4096, 64)
    common = dict(conditions=None, model_config={}, height=1024, width=1024, num_inference_steps=T_, guidance_scale=3.5,
                  prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent")
    fast = generate(pipe, latents=noise.clone(), **common).images
    hip_steps = []
    slow = generate(pipe, latents=noise.clone(), callback_on_step_end=lambda p, i, t, kw: hip_steps.append(kw["latents"].clone()) or {},
                    callback_on_step_end_tensor_inputs=["latents"], **common).images
    assert len(hip_steps) == T_ and torch.equal(fast, slow), "the timed fast path and the per-step path differ at cfg2"
    ref_steps, tb_steps = [], []
    ref = O.denoise(om, noise.float(), pe.float(), pooled.float(), T_, guidance_scale=3.5, model_config={}, conditioning_dtype=BF,
                    callback=lambda i, t, x: ref_steps.append(x.clone()))
    tb = O.denoise(ob, noise, pe, pooled, T_, guidance_scale=3.5, model_config={}, callback=lambda i, t, x: tb_steps.append(x.clone()))
    other_noise = get_noises(2 ** 31 - 1, 1, 1024, 1024, device=dev, dtype=BF, seeds=[seed + 1])[seed + 1]
    other = generate(pipe, latents=other_noise, **common).images
    unrelated = rel_l2(other, ref)
    _fifty_step_table("cfg2 (57 blocks, S=4608, T=50)", hip_steps, ref_steps, tb_steps, {"unrelated_candidate_rel_l2": unrelated, "seed": seed})
    e_hip, _ = _check_final("cfg2 50-step latent (57 blocks, 1024^2)", fast, ref, tb)
    print(f"  an unrelated candidate (seed + 1) ends {unrelated:.3f} away")
    assert e_hip < 0.25 * unrelated


@torch.no_grad()
def test_cfg4_fifty_step_latent_vs_fp32_oracle_both_lora_forms(full):
    """BASELINE cfg4's denoise: 1024 x 1024 + 512^2 "cot" condition (S = 5632), r = 32 FLUX-Corrector-shaped LoRA gated to the
    condition rows, T = 50.  BOTH LoRA forms of the product: the K-segment form (base GEMM + rank-32 correction, the reference's
    rounding order; what training uses) and the merged form the search entry points default to (`runner.build_pipeline` ->
    `enable_merged_lora()`), each against the SAME fp32 oracle trajectory."""
    from reflectionflow_amd.flux.condition import Condition
    from reflectionflow_amd.flux.generate import generate
    from reflectionflow_amd.tts.utils import get_noises
    dev, pipe, om, ob = full
    T_, seed = 50, 4321
    pe, pooled, _, cond, _, _ = _inputs(dev, 512, 4096, 1024, seed=9)
    cond_ids = O.condition_ids_for(512)
    noise = get_noises(2 ** 31 - 1, 1, 1024, 1024, device=dev, dtype=BF, seeds=[seed])[seed]
    cfg = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
    common = dict(model_config=cfg, default_lora=True, height=1024, width=1024, num_inference_steps=T_, guidance_scale=3.5,
                  prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent")
    conds = lambda: [Condition("cot", tokens=cond, ids=cond_ids.to(dev))]       # noqa: E731
    okw = dict(guidance_scale=3.5, condition_ids=cond_ids, model_config=cfg, image_hw=(64, 64))
    ref_steps, tb_steps = [], []
    ref = O.denoise(om, noise.float(), pe.float(), pooled.float(), T_, condition_latents=cond.float(), conditioning_dtype=BF,
                    callback=lambda i, t, x: ref_steps.append(x.clone()), **okw)
    tb = O.denoise(ob, noise, pe, pooled, T_, condition_latents=cond, callback=lambda i, t, x: tb_steps.append(x.clone()), **okw)
    res = {}
    for form in ("k_segment", "merged"):
        pipe.enable_merged_lora(form == "merged")
        try:
            fast = generate(pipe, latents=noise.clone(), conditions=conds(), **common).images
            steps = []
            slow = generate(pipe, latents=noise.clone(), conditions=conds(), callback_on_step_end_tensor_inputs=["latents"],
                            callback_on_step_end=lambda p, i, t, kw: steps.append(kw["latents"].clone()) or {}, **common).images
        finally:
            pipe.enable_merged_lora(False)
        assert len(steps) == T_ and torch.equal(fast, slow), f"fast vs per-step path differ at cfg4 ({form})"
        _fifty_step_table(f"cfg4 {form} LoRA (57 blocks, S=5632, r=32, T=50)", steps, ref_steps, tb_steps, {"seed": seed})
        res[form] = (fast, _check_final(f"cfg4 50-step latent, {form} LoRA", fast, ref, tb))
    d = rel_l2(res["merged"][0], res["k_segment"][0])
    print(f"  merged vs K-segment form after 50 steps: {d:.3e}")
    base = generate(pipe, latents=noise.clone(), conditions=None, **dict(common, model_config={})).images
    assert rel_l2(res["k_segment"][0], base) > 5 * res["k_segment"][1][0], "the condition + LoRA do not move the 50-step latent"
