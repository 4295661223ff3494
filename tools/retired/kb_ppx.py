"""Ping-pong main loop experiments (rf_debug_force_gemm_tile(259), variant via rf_debug_gemm_w4_knock):
0 production order, 1 reads before DMA, 2 no DMA (timing only), 3 no reads (timing only), 4 neither (timing only)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
NAMES = {0: "round-1 loop", 1: "reads first", 2: "no DMA", 3: "no reads", 4: "MFMA+barriers", 5: "balanced 32x32", 6: "even 32x32",
         7: "shipped 16x16x32", 8: "16x16 no DMA", 9: "16x16 no reads", 10: "16x16 MFMA+barriers", 11: "even 16x16x32"}
if len(sys.argv) > 1:
    NAMES = {int(k): NAMES.get(int(k), f"var{k}") for k in sys.argv[1].split(",")}
for (M, N, K) in ((4608, 3072, 12288), (4608, 9216, 3072)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g = [ops.Group([ops.Seg(a, w)], out=y)]
    ref = a.float() @ w.float().t()
    lib.rf_debug_force_gemm_tile(259)
    for k in NAMES:
        if k in (2, 3, 4, 8, 9, 10):
            continue
        lib.rf_debug_gemm_w4_knock(k)
        ops.gemm(g, N, splitk_ws=False)
        err = ((y.float() - ref).norm() / ref.norm()).item()
        print(f"var {k} {M}x{N}x{K}: rel-L2 {err:.3e}", flush=True)
        assert err < 5e-3
    nkt = K // 64
    for rep in range(3):
        for k, name in NAMES.items():
            lib.rf_debug_gemm_w4_knock(k)
            t = ops.time_gemm(g, N, iters=20, splitk_ws=False)
            torch.cuda.synchronize()
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(0, C.byref(mhz), C.byref(us))
            print(f"{M}x{N}x{K} {name:16s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF   shader clock {mhz.value:6.0f} MHz over block 0's main loop ({us.value:.1f} us)"
                  f"  -> {2*M*N*K/t/1e12 / (2516.6 * mhz.value / 2400.0) * 100:5.1f} % of the MFMA rate at that clock", flush=True)
    lib.rf_debug_gemm_w4_knock(0)
    lib.rf_debug_force_gemm_tile(0)
