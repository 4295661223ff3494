#!/usr/bin/env python3
"""BASELINE cfg5-shaped measurement on ONE GPU (not the driver's bench line): FLUX.1-dev shape, 2048^2 image
(16384 tokens) + 512^2 condition (1024 tokens) + 512 text = 17920 joint tokens, FLUX-Corrector-shaped LoRA (r=32,
condition rows only), 456.3 TFLOP per forward (attention is 49 % of it).

    python tools/bench_cfg5.py [--steps T] [--w8]     # --w8: fp8 (e4m3) weights + activations on the big GEMMs

Prints one JSON line: s/latent for T steps, TFLOP/s, fraction of the bf16 MFMA peak, and the per-kernel-class split
(GEMM vs attention vs row kernels) from the library's in-sequence timing hook over 2 profiled forwards."""
import argparse, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflectionflow_amd import ops
from reflectionflow_amd.flux.condition import Condition
from reflectionflow_amd.flux.generate import generate
from reflectionflow_amd.flux.pipeline import synthetic_lora_state_dict
from reflectionflow_amd.tts.utils import get_noises

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10, help="Euler steps per timed latent (per-step cost is step-invariant)")
ap.add_argument("--res", type=int, default=2048)
ap.add_argument("--w8", action="store_true")
ap.add_argument("--merged-lora", action="store_true", help="pipe.enable_merged_lora(): LoRA folded into per-group weight copies")
ap.add_argument("--no-cond", action="store_true", help="no condition image (with --res 1024 this is cfg2's S = 4608)")
args = ap.parse_args()
dev = torch.device("cuda:0"); bf = torch.bfloat16
pipe = bench.build_model(dev, {}, seed=0)
pipe.load_lora_weights({k: v.to(dev) for k, v in synthetic_lora_state_dict(pipe.transformer, r=32, seed=1).items()}, adapter_name="reflection")
if args.merged_lora:
    pipe.enable_merged_lora()
if args.w8:
    pipe.enable_fp8_weights()
g = torch.Generator().manual_seed(1)
pe = torch.randn(1, 512, 4096, generator=g).to(dev).to(bf); pooled = torch.randn(1, 768, generator=g).to(dev).to(bf)
cond_tokens = torch.randn(1, 1024, 64, generator=g).to(dev).to(bf)
ids = pipe._prepare_latent_image_ids(1, 32, 32, dev, bf)
mc = {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}
noises = get_noises(2**31 - 1, 2, args.res, args.res, device=dev, dtype=bf, seeds=[1, 2])
S_img = (args.res // 16) ** 2
f_fwd, f_gemm, f_attn = bench.flops_per_forward(512, S_img, S_cond=1024)

def one(seed, T):
    cond = Condition("cot", tokens=cond_tokens, ids=ids, position_delta=[0, -32])
    return generate(pipe, conditions=None if args.no_cond else [cond], model_config=mc, default_lora=True, height=args.res, width=args.res,
                    num_inference_steps=T, guidance_scale=3.5, latents=noises[seed], prompt_embeds=pe,
                    pooled_prompt_embeds=pooled, output_type="latent").images

T = args.steps
one(1, 2); torch.cuda.synchronize()
t0 = time.perf_counter(); o = one(2, T); torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert torch.isfinite(o.float()).all()
with ops.profile(4096) as pr:
    one(1, 2); torch.cuda.synchronize()
cl = {}
for k, v in pr.classes.items():
    cl[k] = {"launches_per_forward": v["launches"] / 2, "ms_per_forward": round(v["us"] / 2e3, 2)}
    if k != "rowop" and k != "quant":
        cl[k]["tflops"] = round(v["work"] / (v["us"] * 1e-6) / 1e12, 1)
print(json.dumps({"workload": f"cfg5-shaped: {args.res}^2 + 512^2 condition, LoRA r=32 on condition rows, S={512 + S_img + 1024}, "
                              f"{f_fwd / 1e12:.1f} TFLOP/forward (GEMM {f_gemm / 1e12:.1f} + attention {f_attn / 1e12:.1f})",
                  "weights": "fp8 e4m3 (W8A8 on the block GEMMs)" if args.w8 else "bf16", "steps": T,
                  "s_per_step": round(dt / T, 4), "s_per_50_step_latent": round(dt / T * 50, 2),
                  "tflops": round(f_fwd * T / dt / 1e12, 1), "frac_of_2p5PF": round(f_fwd * T / dt / 2.5e15, 4),
                  "classes": cl}))
