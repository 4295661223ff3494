"""Knock-out timing of the one-wave-per-SIMD main loop (results are wrong by construction): which of LDS-DMA issue,
fragment reads, and the per-tile wait+barrier costs what on top of 64 MFMAs per K-tile."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 4608, 3072, 12288     # 216 tiles (one round), 192 K-tiles: main loop dominates
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
g = [ops.Group([ops.Seg(a, w)], out=y)]
NAMES = {0: "full", 1: "no DMA", 2: "no reads", 16: "barrier, no vmcnt wait", 32: "waits, no barrier", 4: "no wait, no barrier", 7: "MFMA only"}
lib.rf_debug_force_gemm_tile(258)
for rep in range(2):
    for k, name in NAMES.items():
        lib.rf_debug_gemm_w4_knock(k)
        t = ops.time_gemm(g, N, iters=20, splitk_ws=False)
        torch.cuda.synchronize()
        mhz, us = C.c_double(0), C.c_double(0)
        lib.rf_debug_clock_probe(0, C.byref(mhz), C.byref(us))
        cyc = us.value * mhz.value / 192
        print(f"{name:18s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF   {mhz.value:5.0f} MHz, block 0 main loop {us.value:6.1f} us = {cyc:6.0f} shader clocks per K-tile (64 MFMA = 2048)", flush=True)
lib.rf_debug_gemm_w4_knock(0)
lib.rf_debug_force_gemm_tile(0); lib.rf_debug_force_gemm_sk(0)
t = ops.time_gemm(g, N, iters=20, splitk_ws=False)
print(f"8-wave ping-pong   {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF", flush=True)
