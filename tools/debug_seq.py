#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import flux_oracle as O
from reflectionflow_amd import engine as E
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
S_txt, S_img = 512, 4096
pe = torch.randn(S_txt, 4096, generator=g).to(dev).to(BF)
lat = torch.randn(S_img, 64, generator=g).to(dev).to(BF)
img_ids = O.prepare_latent_image_ids(64, 64).to(dev); txt_ids = torch.zeros(S_txt, 3, device=dev)
for nd, ns in [(1, 0), (0, 1), (3, 0), (0, 3), (19, 38)]:
    pipe = bench.build_model(dev, dict(num_layers=nd, num_single_layers=ns), seed=0)
    eng = E.engine_for(pipe.transformer)
    temb = torch.randn(1, 3072, generator=g).to(dev).to(BF)
    mod = eng.mod_table(temb)[0].contiguous()
    cos, sin = eng.rope_tables(txt_ids, img_ids)
    torch.cuda.synchronize()
    outs = []
    for rep in range(12):
        outs.append(eng.forward(lat, pe, mod, cos, sin).clone())
    torch.cuda.synchronize()
    nan = [int((~torch.isfinite(o.float())).sum()) for o in outs]
    nd_ = [int((o != outs[0]).sum()) for o in outs]
    print(f"nd={nd} ns={ns}: non-finite counts {nan}  differing vs run0 {nd_}", flush=True)
    del pipe, eng
    torch.cuda.empty_cache()
