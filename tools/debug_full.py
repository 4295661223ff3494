#!/usr/bin/env python3
"""Full-size (FLUX.1-dev shape) forward: HIP path vs the fp32 oracle evaluated on the same GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import flux_oracle as O  # noqa: E402
from reflectionflow_amd.flux.transformer import tranformer_forward  # noqa: E402
from reflectionflow_amd.flux.generate import generate  # noqa: E402

dev = torch.device("cuda:0")
nd, ns = int(os.environ.get("ND", "19")), int(os.environ.get("NS", "38"))
pipe = bench.build_model(dev, dict(num_layers=nd, num_single_layers=ns), seed=0)
tr = pipe.transformer
BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
S_txt, S_img = 512, 4096
pe = torch.randn(1, S_txt, 4096, generator=g).to(dev).to(BF)
pooled = torch.randn(1, 768, generator=g).to(dev).to(BF)
lat = torch.randn(1, S_img, 64, generator=g).to(dev).to(BF)
img_ids = O.prepare_latent_image_ids(64, 64).to(dev)
txt_ids = torch.zeros(S_txt, 3, device=dev)
t = torch.tensor([0.5], device=dev)
gd = torch.tensor([4.0], device=dev)
with torch.no_grad():
    out = tranformer_forward(tr, None, None, None, model_config={}, hidden_states=lat, encoder_hidden_states=pe,
                             pooled_projections=pooled, timestep=t, guidance=gd, img_ids=img_ids, txt_ids=txt_ids,
                             return_dict=False)[0]
    torch.cuda.synchronize()
    print("hip forward finite:", bool(torch.isfinite(out.float()).all()), "absmean", float(out.float().abs().mean()), flush=True)
    if os.environ.get("ORACLE", "1") == "1":
        with torch.device(dev):
            om = O.FluxTransformer2DModel(num_layers=nd, num_single_layers=ns).float().eval()
        om.load_state_dict({k: v.float() for k, v in tr.state_dict().items()})
        ref = O.tranformer_forward(om, None, None, None, model_config={}, hidden_states=lat.float(),
                                   encoder_hidden_states=pe.float(), pooled_projections=pooled.float(), timestep=t,
                                   guidance=gd, img_ids=img_ids.cpu(), txt_ids=txt_ids.cpu(), return_dict=False)[0]
        print("oracle finite:", bool(torch.isfinite(ref).all()), "absmean", float(ref.abs().mean()))
        print("rel-L2 hip vs fp32 oracle:", float((out.float() - ref).norm() / ref.norm()), flush=True)
        del om
    # multi-step: where does it go non-finite?  is the loop deterministic run-to-run?
    T = int(os.environ.get("T", "50"))
    runs = []
    for rep in range(2):
        bad, traj = [], []
        def cb(p, i, tt, kw):
            x = kw["latents"]
            if not torch.isfinite(x.float()).all():
                bad.append(i)
            traj.append(x.clone())
            return {}
        res = generate(pipe, model_config={}, height=1024, width=1024, num_inference_steps=T, guidance_scale=3.5,
                       latents=lat.clone(), prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent",
                       callback_on_step_end=cb).images
        print(f"run {rep}: non-finite at steps {bad[:5]}  final absmax {float(res.float().abs().max())}", flush=True)
        runs.append(traj)
    first_diff = next((i for i, (a, b) in enumerate(zip(*runs)) if not torch.equal(a, b)), None)
    print("run-to-run first differing step:", first_diff)
    if first_diff is not None:
        a, b = runs[0][first_diff], runs[1][first_diff]
        d = (a.float() - b.float()).abs()
        idx = torch.nonzero(d > 0)
        print("  differing elements:", idx.shape[0], "max diff", float(d.max()), "rows", sorted(set(idx[:, 1].tolist()))[:20])
    for rep in range(2):
        res2 = generate(pipe, model_config={}, height=1024, width=1024, num_inference_steps=T, guidance_scale=3.5,
                        latents=lat.clone(), prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images
        print("fast loop finite:", bool(torch.isfinite(res2.float()).all()), flush=True)
