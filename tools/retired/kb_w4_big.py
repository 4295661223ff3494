"""One-wave-per-SIMD ring kernel (tile 258) vs the 8-wave balanced ping-pong kernel on long-K / many-tile shapes."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (4608, 12288, 3072)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g = [ops.Group([ops.Seg(a, w)], out=y)]
    for rep in range(3):
        line = [f"{M}x{N}x{K}"]
        for name, tile in (("pp2", 256), ("w4-ring", 258)):
            lib.rf_debug_force_gemm_tile(tile); lib.rf_debug_force_gemm_sk(0); lib.rf_debug_gemm_w4_knock(0)
            t = ops.time_gemm(g, N, iters=20, splitk_ws=False)
            torch.cuda.synchronize()
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(0, C.byref(mhz), C.byref(us))
            line.append(f"{name} {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF @ {mhz.value:5.0f} MHz, block 0 loop {us.value:6.1f} us = {us.value*mhz.value/(K//64):5.0f} clk/K-tile")
        print(" | ".join(line), flush=True)
lib.rf_debug_force_gemm_tile(0); lib.rf_debug_force_gemm_sk(-1)
