"""Candidates of one rank on several HIP streams: does overlapping two 50-step denoise loops (each one hipGraph replay) fill the CUs
that a single loop's partial last rounds leave idle?  python tools/kb_streams2.py [--streams 1,2,3,4] [--cands 4]"""
import argparse, sys, time, json
import torch
sys.path.insert(0, ".")
import bench
from reflectionflow_amd.flux.generate import generate
from reflectionflow_amd.tts.utils import get_noises

ap = argparse.ArgumentParser()
ap.add_argument("--streams", default="1,2,4")
ap.add_argument("--cands", type=int, default=4)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--T", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
pipe = bench.build_model(dev, {}, seed=0)
g = torch.Generator().manual_seed(1)
pe = torch.randn(1, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
pooled = torch.randn(1, 768, generator=g).to(dev).to(torch.bfloat16)
seeds = [7919 * j + 13 for j in range(args.cands)]
noises = get_noises(2 ** 31 - 1, args.cands, args.res, args.res, device=dev, dtype=torch.bfloat16, seeds=seeds)


def one(seed):
    return generate(pipe, model_config={}, height=args.res, width=args.res, num_inference_steps=args.T, guidance_scale=3.5, latents=noises[seed],
                    prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images


ref = {}
for sd in seeds[:2]:
    ref[sd] = one(sd).clone()       # eager first call + graph capture on the default stream
torch.cuda.synchronize()
res = {}
for ns in [int(x) for x in args.streams.split(",")]:
    streams = [torch.cuda.Stream() for _ in range(ns)]
    outs = {}
    for rep in range(2):            # rep 0 warms (captures one graph per stream), rep 1 is timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, sd in enumerate(seeds):
            st = streams[i % ns]
            with torch.cuda.stream(st):
                outs[sd] = one(sd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ok = all(torch.equal(outs[sd], ref[sd]) for sd in ref)
    res[ns] = {"latents_per_s": round(args.cands / dt, 4), "s_per_latent": round(dt / args.cands, 3), "bit_equal_to_single_stream": ok}
    print(ns, "streams:", json.dumps(res[ns]), flush=True)
