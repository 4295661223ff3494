"""GEMM main loop A/B (experiments library): four 16-MFMA phases per K-tile (gemm_mainloop_pp3_m16, shipped) vs two 32-MFMA phases
(gemm_mainloop_pp4_m16) on the launches of one FLUX forward; bit-equality of the outputs; interleaved repetitions."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib, ops
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_STORE
from tools.kbench import timeit
dev = torch.device("cuda:0"); BF = torch.bfloat16; lib = _lib.load()
lib.rf_debug_force_gemm_tile(256); lib.rf_debug_force_gemm_sk(0)
shapes = [("dbl_qkv", 4608, 9216, 3072, RF_EPI_STORE), ("dbl_out", 4608, 3072, 3072, RF_EPI_GATE_RES), ("dbl_ff1", 4608, 12288, 3072, RF_EPI_GELU),
          ("dbl_ff2", 4608, 3072, 12288, RF_EPI_GATE_RES), ("sgl_in", 4608, 21504, 3072, RF_EPI_STORE), ("sgl_out", 4608, 3072, 15360, RF_EPI_GATE_RES),
          ("ragged", 1000, 1300, 192, RF_EPI_STORE), ("sq8192", 8192, 8192, 8192, RF_EPI_STORE)]
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * 0.02).to(BF); b = torch.randn(N, device=dev).to(BF)
    res0 = torch.randn(M, N, device=dev).to(BF)
    outs = {}
    line = []
    for rep in range(3):
        for pp4 in (0, 1):
            lib.rf_debug_gemm_pp4(pp4)
            out = res0.clone()
            kw = dict(residual=out, gate=b) if epi == RF_EPI_GATE_RES else {}
            ops.linear(x, W, b, epilogue=epi, out=out, **kw)
            torch.cuda.synchronize()
            outs[pp4] = out.clone()
            o2 = torch.empty(M, N, device=dev, dtype=BF)
            kw2 = dict(residual=res0, gate=b) if epi == RF_EPI_GATE_RES else {}
            t = timeit(lambda: ops.linear(x, W, b, epilogue=epi, out=o2, **kw2), 8)
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(0, C.byref(mhz), C.byref(us))
            line.append(f"{'pp4' if pp4 else 'pp3'} {t*1e6:7.1f}us {2.0*M*N*K/t/1e12:6.0f}TF @{mhz.value:4.0f}")
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"{name:8s} {'==' if same else '!= max|d| %.3g' % float((outs[0].float()-outs[1].float()).abs().max())} | " + " | ".join(line), flush=True)
lib.rf_debug_gemm_pp4(0); lib.rf_debug_force_gemm_tile(0); lib.rf_debug_force_gemm_sk(-1)
