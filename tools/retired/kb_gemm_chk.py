"""Correctness of an experimental main-loop variant (rf_debug_force_gemm_tile(259) + rf_debug_gemm_w4_knock(VAR)) against
fp32 matmul next to the production kernel: K-tile counts 1, 2, 3, 5, 48, 240, ragged M / N, two K-segments."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else 259   # 258 = one-wave-per-SIMD ring kernel
def run(groups_fn, N, ref, tag):
    for tile, var in ((256, 0), (TILE, VAR)):
        lib.rf_debug_force_gemm_tile(tile); lib.rf_debug_force_gemm_sk(0); lib.rf_debug_gemm_w4_knock(var)
        y = groups_fn()
        torch.cuda.synchronize()
        err = ((y.float() - ref).norm() / ref.norm()).item()
        print(tag, "tile", tile, "var", var, f"rel-L2 {err:.3e}", flush=True)
        assert err < 5e-3
for (M, N, K) in ((512, 512, 64), (512, 512, 128), (512, 512, 192), (300, 520, 320), (4608, 3072, 3072), (1024, 1024, 15360)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    ref = a.float() @ w.float().t()
    def f():
        y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm([ops.Group([ops.Seg(a, w)], out=y)], N, splitk_ws=False)
        return y
    run(f, N, ref, f"{M}x{N}x{K}")
M, N = 1024, 3072
a0 = torch.randn(M, 3072, device=dev, dtype=torch.bfloat16); w0 = torch.randn(N, 3072, device=dev, dtype=torch.bfloat16) * 0.05
a1 = torch.randn(M, 64, device=dev, dtype=torch.bfloat16); w1 = torch.randn(N, 64, device=dev, dtype=torch.bfloat16) * 0.05
ref = a0.float() @ w0.float().t() + a1.float() @ w1.float().t()
def f2():
    y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Group([ops.Seg(a0, w0), ops.Seg(a1, w1)], out=y)], N, splitk_ws=False)
    return y
run(f2, N, ref, "2-seg")
lib.rf_debug_force_gemm_tile(0); lib.rf_debug_force_gemm_sk(-1); lib.rf_debug_gemm_w4_knock(0)
