#!/usr/bin/env python3
"""BASELINE cfg4-shaped measurement on ONE GPU (not the driver's bench line): FLUX.1-dev shape, 1024^2 image +
512^2 condition (1024 condition tokens, S = 5632), FLUX-Corrector-shaped LoRA (r=32 on the config.yaml:53 modules,
active on the condition rows only: latent_lora=False), 50 steps (and the reference's default 28)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflectionflow_amd.flux.condition import Condition
from reflectionflow_amd.flux.generate import generate
from reflectionflow_amd.flux.pipeline import synthetic_lora_state_dict
from reflectionflow_amd.tts.utils import get_noises
dev = torch.device("cuda:0"); bf = torch.bfloat16
pipe = bench.build_model(dev, {}, seed=0)
pipe.load_lora_weights({k: v.to(dev) for k, v in synthetic_lora_state_dict(pipe.transformer, r=32, seed=1).items()}, adapter_name="reflection")
MERGED = "--merged-lora" in sys.argv
if MERGED:
    pipe.enable_merged_lora()     # W + s B A copies for the condition rows: no low-rank launches per step
g = torch.Generator().manual_seed(1)
pe = torch.randn(1, 512, 4096, generator=g).to(dev).to(bf); pooled = torch.randn(1, 768, generator=g).to(dev).to(bf)
cond_tokens = torch.randn(1, 1024, 64, generator=g).to(dev).to(bf)
ids = pipe._prepare_latent_image_ids(1, 32, 32, dev, bf)
mc = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
noises = get_noises(2**31 - 1, 3, 1024, 1024, device=dev, dtype=bf, seeds=[1, 2, 3])
def one(seed, T):
    cond = Condition("cot", tokens=cond_tokens, ids=ids, position_delta=[0, -32])
    return generate(pipe, conditions=[cond], model_config=mc, default_lora=True, height=1024, width=1024, num_inference_steps=T,
                    guidance_scale=3.5, latents=noises[seed], prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images
res = {}
for T in ((20,) if "--quick" in sys.argv else (50, 28)):
    o = one(1, T); torch.cuda.synchronize()
    t0 = time.perf_counter(); outs = [one(s, T) for s in (2, 3)]; torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    assert all(torch.isfinite(x.float()).all() for x in outs)
    flops = 94.95e12 * T
    res[f"T{T}"] = dict(s_per_latent=round(dt, 3), latents_per_s=round(1 / dt, 4), tflops=round(flops / dt / 1e12, 1), frac_of_2p5PF=round(flops / dt / 2.5e15, 4))
from reflectionflow_amd import ops
with ops.profile(4096) as pr:     # per-kernel-class split over 4 profiled forwards (the library's in-sequence hook)
    one(1, 4); torch.cuda.synchronize()
cl = {}
for k, v in pr.classes.items():
    cl[k] = {"launches_per_forward": v["launches"] / 4, "ms_per_forward": round(v["us"] / 4e3, 2)}
    if k not in ("rowop", "quant"):
        cl[k]["tflops"] = round(v["work"] / (v["us"] * 1e-6) / 1e12, 1)
res["classes_T4_profile"] = cl
print(json.dumps({"workload": "cfg4-shaped: 1024^2 + 512^2 condition, LoRA r=32 on condition rows, S=5632, 94.95 TFLOP/forward (SURVEY 8d)",
                  "lora": "merged per-group weight copies (pipe.enable_merged_lora())" if MERGED else "K-segments (x A^T as its own launches)", **res}))
