"""what the VAE (PyTorch-ROCm / MIOpen modules, reflectionflow_amd/flux/vae.py) costs per candidate next
to the 3.1 s denoise: decode of a 1024^2 latent (16 x 128 x 128) and encode of a 512^2 condition image, bf16, contiguous vs
channels_last."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd.flux.vae import AutoencoderKL, init_synthetic_vae_
dev = torch.device("cuda:0")
vae = AutoencoderKL().to(dev).to(torch.bfloat16).eval()
init_synthetic_vae_(vae, seed=0)
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for fmt in ("contiguous", "channels_last"):
    if fmt == "channels_last":
        vae = vae.to(memory_format=torch.channels_last)
    z = torch.randn(1, 16, 128, 128, device=dev, dtype=torch.bfloat16)
    x = torch.randn(1, 3, 512, 512, device=dev, dtype=torch.bfloat16)
    if fmt == "channels_last":
        z, x = z.contiguous(memory_format=torch.channels_last), x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        td = timeit(lambda: vae.decode(z).sample)
        te = timeit(lambda: vae.encode(x).latent_dist.mean)
    print(f"{fmt:14s} decode 1024^2: {td*1e3:8.1f} ms   encode 512^2: {te*1e3:8.1f} ms", flush=True)
