"""Training-path bench: forward + backward of ONE DoubleStream + ONE SingleStream block at FLUX.1-dev width through
reflectionflow_amd.train (the flow-matching step, LoRA r = 32 on the FLUX-Corrector target list, condition tokens), timed with torch
events and broken down per kernel class with the library's in-sequence hook.   python tools/kb_train.py [--res 1024] [--cond 512]

FLOP accounting (2 M N K per GEMM, 4 S^2 D per attention forward; element-wise work excluded):
  forward   F = GEMM_f + ATT_f                      (one block pair)
  backward  B = GEMM_f (dX through every frozen weight) + 2.5 ATT_f (+ LoRA factor GEMMs, ~1 %, not counted)
  executed    = 2 F + B   (the backward RE-COMPUTES the block: train_flux/flux/transformer.py:139-157)
  model       = F + B     (what a step needs without recompute)"""
import argparse
import json
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--cond", type=int, default=512)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--rank", type=int, default=32)
ap.add_argument("--layers", type=int, default=1, help="DoubleStream blocks (FLUX.1-dev: 19)")
ap.add_argument("--single-layers", type=int, default=1, help="SingleStream blocks (FLUX.1-dev: 38)")
ap.add_argument("--optimizer", action="store_true", help="time an AdamW update of the LoRA factors inside the step (train/model.py:105-139)")
args = ap.parse_args()

from reflectionflow_amd import ops                                   # noqa: E402
from reflectionflow_amd.flux import modules as M                     # noqa: E402
from reflectionflow_amd.flux.pipeline import FluxPipeline, synthetic_lora_state_dict   # noqa: E402
from reflectionflow_amd.train.step import FluxTrainer, lora_parameters                 # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
old = torch.get_default_dtype()
torch.set_default_dtype(BF)
with torch.device(dev):
    tr = M.FluxTransformer2DModel(num_layers=args.layers, num_single_layers=args.single_layers)
torch.set_default_dtype(old)
M.init_synthetic_(tr, seed=0)
pipe = FluxPipeline(tr)
pipe.load_lora_weights(synthetic_lora_state_dict(tr, r=args.rank, seed=3), adapter_name="default")
D, mlp, H = 3072, 12288, 24
St, Si, Sc = 512, (args.res // 16) ** 2, (args.cond // 16) ** 2
S = St + Si + Sc
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g, device=dev).to(BF)   # noqa: E731
gh, gc = args.res // 16, args.cond // 16


def ids(n):
    return torch.stack([torch.zeros(n * n), torch.arange(n).repeat_interleave(n).float(), torch.arange(n).repeat(n).float()], 1).to(dev)


cond_ids = ids(gc)
cond_ids[:, 2] -= gc
batch = dict(x_0=r(1, Si, 64), img_ids=ids(gh), prompt_embeds=r(1, St, 4096), pooled_prompt_embeds=r(1, 768), text_ids=torch.zeros(St, 3, device=dev),
             condition_latents=r(1, Sc, 64), condition_ids=cond_ids, t=torch.tensor([0.5], device=dev), x_1=r(1, Si, 64))
trainer = FluxTrainer(tr, {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False})
params = lora_parameters(tr)
opt = torch.optim.AdamW(params, lr=1e-4, fused=True) if args.optimizer else None


def step():
    for p in params:
        p.grad = None
    loss = trainer.step(batch)
    loss.backward()
    if opt is not None:
        opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    loss = step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
with ops.profile(max_launches=min(24000, 2000 * (args.layers + args.single_layers))) as pr:
    step()
    torch.cuda.synchronize()
nd, ns = args.layers, args.single_layers
gemm_f = nd * S * (2 * D * 3 * D + 2 * D * D + 2 * 2 * D * mlp) + ns * S * (2 * D * (3 * D + mlp) + 2 * (D + mlp) * D)
att_f = (nd + ns) * 4 * S * S * D
F_, B_ = gemm_f + att_f, gemm_f + 2.5 * att_f
cl = pr.classes
res = {"workload": f"{nd} DoubleStream + {ns} SingleStream blocks, D=3072, S = {St} text + {Si} image + {Sc} condition = {S}, LoRA r = {args.rank} on the condition rows",
       "ms_per_step_fwd_bwd": round(ms, 3), "optimizer_in_step": "AdamW (fused) on the LoRA factors" if opt is not None else None,
       "lora_parameters": sum(p.numel() for p in params), "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2), "loss": float(loss),
       "tflop": {"forward": round(F_ / 1e12, 3), "backward": round(B_ / 1e12, 3), "executed_with_recompute": round((2 * F_ + B_) / 1e12, 3)},
       "tflops_executed": round((2 * F_ + B_) / ms / 1e9, 1), "tflops_model": round((F_ + B_) / ms / 1e9, 1),
       "frac_of_bf16_mfma_peak_executed": round((2 * F_ + B_) / ms / 1e9 / 2500.0, 4),
       "classes": {k: {"launches": v["launches"], "ms": round(v["us"] / 1e3, 3),
                       **({"tflops": round(v["work"] / v["us"] / 1e6, 1)} if k in ("gemm_main", "gemm_small", "attention", "attention_bwd") else
                          {"GBps": round(v["work"] / v["us"] / 1e3, 1)})} for k, v in cl.items()},
       "profiled_sum_ms": round(sum(v["us"] for v in cl.values()) / 1e3, 3), "dropped": pr.dropped}
print(json.dumps(res, indent=1))
