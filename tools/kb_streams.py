"""Two candidates on two HIP streams vs back-to-back on one stream (whole 50-step denoises)."""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflectionflow_amd.flux.generate import generate
from reflectionflow_amd.tts.utils import get_noises
dev = torch.device("cuda:0")
T = int(os.environ.get("T", "50"))
pipe = bench.build_model(dev, {}, seed=0)
g = torch.Generator().manual_seed(1)
pe = torch.randn(1, 512, 4096, generator=g).to(dev).to(torch.bfloat16)
pooled = torch.randn(1, 768, generator=g).to(dev).to(torch.bfloat16)
noises = get_noises(2**31 - 1, 4, 1024, 1024, device=dev, dtype=torch.bfloat16, seeds=[1, 2, 3, 4])
def one(seed):
    return generate(pipe, model_config={}, height=1024, width=1024, num_inference_steps=T, guidance_scale=3.5,
                    latents=noises[seed], prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent").images
one(1); torch.cuda.synchronize()
t0 = time.perf_counter(); a = [one(s) for s in (1, 2, 3, 4)]; torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
print(f"sequential: 4 latents in {t_seq:.3f}s = {4/t_seq:.4f} latents/s", flush=True)
res = {}
def worker(seeds, stream):
    with torch.cuda.stream(stream):
        for s in seeds:
            res[s] = one(s)
for nstreams in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    parts = [[s for i, s in enumerate((1, 2, 3, 4)) if i % nstreams == k] for k in range(nstreams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(p, st)) for p, st in zip(parts, streams)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    same = all(torch.equal(res[s], a[i]) for i, s in enumerate((1, 2, 3, 4)))
    print(f"{nstreams} streams : 4 latents in {t_par:.3f}s = {4/t_par:.4f} latents/s  ({t_seq/t_par:.3f}x)  identical results: {same}", flush=True)
