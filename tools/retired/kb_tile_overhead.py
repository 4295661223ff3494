"""What a 256x256 output tile costs besides its K loop: the same launch grids at K = 64 (one K-tile) and K = 3072,
plain store vs gate-residual epilogue: time per round of 256 tiles."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0"); BF = torch.bfloat16
lib.rf_debug_force_gemm_sk(0)
for (M, N) in ((4608, 3072), (4608, 12288), (4608, 21504)):
    tiles = (M // 256) * (N // 256); rounds = -(-tiles // 256)
    for K in (64, 128, 3072):
        x = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        b = torch.randn(N, device=dev).to(BF); gate = torch.randn(N, device=dev).to(BF)
        y = torch.empty(M, N, device=dev, dtype=BF); res = torch.randn(M, N, device=dev).to(BF)
        line = [f"M={M} N={N} K={K:5d} tiles={tiles} rounds={rounds}"]
        for name, epi, kw in (("store", ops.RF_EPI_STORE, dict(bias=b, out=y)), ("gelu", ops.RF_EPI_GELU, dict(bias=b, out=y)),
                              ("gate_res", ops.RF_EPI_GATE_RES, dict(bias=b, out=res, residual=res, gate=gate))):
            t = min(ops.time_gemm([ops.Group([ops.Seg(x, W)], **kw)], N, epi, iters=20, splitk_ws=False) for _ in range(3))
            torch.cuda.synchronize()
            m0, u0, m2, u2 = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(0, C.byref(m0), C.byref(u0)); lib.rf_debug_clock_probe(2, C.byref(m2), C.byref(u2))
            line.append(f"{name} {t*1e6:7.1f} us = {t*1e6/rounds:6.1f} per round (block 0: start..loop end {u0.value:5.1f} us, epilogue {u2.value:5.1f} us)")
        print(" | ".join(line), flush=True)
lib.rf_debug_force_gemm_sk(-1)
