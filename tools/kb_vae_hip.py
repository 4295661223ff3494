"""SURVEY 8(f) row 1 on the HIP path: decode of a 1024^2 candidate (16 x 128 x 128 latent) and encode of a 512^2 condition image
through librf_flux.so (rf_vae_decode / rf_vae_encode), with the library's in-sequence timing hook splitting the time into
GEMM (convolutions as 3-segment GEMMs, 1x1 shortcuts, attention projections), the one-head attention kernel and the row kernels
(GroupNorm statistics / apply, upsample, im2col).  `--torch` also times the PyTorch-ROCm / MIOpen modules (minutes of MIOpen
search on a fresh box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import ops
from reflectionflow_amd.flux.vae import AutoencoderKL, init_synthetic_vae_
from reflectionflow_amd.flux.vae_hip import HipVAE
dev = torch.device("cuda:0")
vae = init_synthetic_vae_(AutoencoderKL(), seed=0).to(dev).to(torch.bfloat16).eval()
hv = HipVAE(vae)
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
FL = {"decode": 0.0, "encode": 0.0}
for name, shape, fn in (("decode 1024^2", (1, 16, 128, 128), lambda t: hv.decode(t).sample),
                        ("encode  512^2", (1, 3, 512, 512), lambda t: hv.encode_moments(t)),
                        ("decode 2048^2", (1, 16, 256, 256), lambda t: hv.decode(t).sample)):
    if "2048" in name and "--big" not in sys.argv:
        continue
    x = torch.randn(*shape, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        t = timeit(lambda: fn(x))
        with ops.profile(4096) as pr:
            fn(x); torch.cuda.synchronize()
    cls = pr.classes
    parts = ", ".join(f"{k} {v['us']/1e3:.2f} ms/{v['launches']}" + (f" ({v['work']/v['us']/1e6:.0f} TF)" if k.startswith("gemm") or k == "attention" else f" ({v['work']/v['us']/1e3:.0f} GB/s)")
                      for k, v in cls.items())
    print(f"HIP {name}: {t*1e3:7.2f} ms   [{parts}]", flush=True)
if "--torch" in sys.argv:
    z = torch.randn(1, 16, 128, 128, device=dev, dtype=torch.bfloat16)
    x = torch.randn(1, 3, 512, 512, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        td = timeit(lambda: vae.decode(z).sample, 5)
        te = timeit(lambda: vae.encode(x).latent_dist.mean, 5)
    print(f"torch/MIOpen decode 1024^2: {td*1e3:8.1f} ms   encode 512^2: {te*1e3:8.1f} ms", flush=True)
