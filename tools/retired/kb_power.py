"""Sustained (6 s) GEMM loops with rocm-smi power/clock sampling: librf_flux kernel vs torch.matmul (hipBLASLt)."""
import json, subprocess, threading, time, sys
import torch
from reflectionflow_amd import _lib, ops
_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
dev = torch.device("cuda:0"); lib = _lib.load(); BF = torch.bfloat16

def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)["card0"]
            out.append((float(d["Current Socket Graphics Package Power (W)"]), int(d["sclk clock speed:"].strip("()Mhz"))))
        except Exception as e:  # noqa
            pass
        time.sleep(0.25)

def sustained(name, fn, flops, secs=6.0):
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, samples)); th.start()
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        fn(); n += 1
        if n % 50 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); dt = time.time() - t0
    stop.set(); th.join()
    tail = samples[len(samples) // 2:]
    pw = sum(s[0] for s in tail) / max(1, len(tail)); ck = sum(s[1] for s in tail) / max(1, len(tail))
    print(f"{name:34s} {flops * n / dt / 1e12:7.0f} TF sustained   {pw:6.0f} W  {ck:5.0f} MHz  ({len(tail)} samples)", flush=True)

for M, N, K in [(8192, 8192, 8192), (4608, 3072, 12288)]:
    x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    out = torch.empty(M, N, device=dev, dtype=BF)
    g = [ops.Group([ops.Seg(x, W)], out=out)]
    fl = 2.0 * M * N * K
    lib.rf_debug_force_gemm_sk(0)
    for tile in [int(a) for a in sys.argv[1:]] or [256]:
        lib.rf_debug_force_gemm_tile(tile)
        sustained(f"rf tile{tile} {M}x{N}x{K}", lambda: ops.gemm(g, N), fl)
    lib.rf_debug_force_gemm_tile(0)
    sustained(f"torch.matmul {M}x{N}x{K}", lambda: torch.matmul(x, W.t(), out=out), fl)
    xz = torch.zeros_like(x); Wz = torch.zeros_like(W); gz = [ops.Group([ops.Seg(xz, Wz)], out=out)]
    sustained(f"rf tile256 ZERO operands", lambda: ops.gemm(gz, N), fl, secs=4.0)
