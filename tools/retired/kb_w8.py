"""rf_gemm_w8a8 vs rf_gemm_bf16 on the FLUX block shapes (isolated launches, random data): time and TFLOP/s.
M = 4608 (cfg2), 16896 (cfg5 text+image rows)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_STORE, Group, Seg
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = _lib.load()
def clk():
    torch.cuda.synchronize()
    mhz, us = C.c_double(0), C.c_double(0)
    lib.rf_debug_clock_probe(0, C.byref(mhz), C.byref(us))
    return mhz.value
lib.rf_debug_force_gemm_sk(0)   # one tile per block: the kernels that carry the clock probe
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)
for M in (4608, 16896):
    for name, N, K, epi in (("qkv-like", 9216, 3072, RF_EPI_STORE), ("out", 3072, 3072, RF_EPI_GATE_RES), ("ff1", 12288, 3072, RF_EPI_GELU),
                            ("ff2", 3072, 12288, RF_EPI_GATE_RES), ("sgl_in-like", 21504, 3072, RF_EPI_GELU), ("sgl_out", 3072, 15360, RF_EPI_GATE_RES)):
        x, W, b, gate = r(M, K), r(N, K, sc=0.02), r(N), r(N)
        out = torch.empty(M, N, device=dev, dtype=BF)
        kw = dict(bias=b, out=out)
        if epi == RF_EPI_GATE_RES:
            kw.update(residual=out, gate=gate)
        t16 = ops.time_gemm([Group([Seg(x, W)], **kw)], N, epi, iters=8)
        c16 = clk()
        A8, sa = ops.quant_rows_fp8(x)
        W8, sw = ops.quantize_weight_fp8(W)
        lib.rf_debug_gemm_mi16(0)
        t8o = ops.time_gemm([Group([Seg(A8, W8)], a_scale=sa, w_scale=sw, **kw)], N, epi, iters=8)
        c8o = clk()
        lib.rf_debug_gemm_mi16(1)
        t8 = ops.time_gemm([Group([Seg(A8, W8)], a_scale=sa, w_scale=sw, **kw)], N, epi, iters=8)
        c8 = clk()
        fl = 2.0 * M * N * K
        print(f"M={M:6d} {name:12s} N={N:6d} K={K:6d}: bf16 {t16*1e6:8.1f} us {fl/t16/1e12:7.1f} TF @{c16:5.0f} MHz | fp8 32x32x64 {t8o*1e6:8.1f} us {fl/t8o/1e12:7.1f} TF @{c8o:5.0f} | fp8 16x16x128 {t8*1e6:8.1f} us {fl/t8/1e12:7.1f} TF @{c8:5.0f} MHz | x{t16/t8:.2f}", flush=True)
