"""A/B: 8-wave ping-pong main loop (one tile per block, rf_debug_force_gemm_sk(0)) vs the 4-wave one-wave-per-SIMD loop
(rf_debug_force_gemm_tile(258)) on the cfg2 launch shapes, plus a correctness check of the latter."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
import bench
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
# correctness
for (M, N, K) in ((512, 512, 256), (4608, 3072, 3072), (777, 1000, 192)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    lib.rf_debug_force_gemm_tile(258)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Group([ops.Seg(a, w)], out=y)], N)
    lib.rf_debug_force_gemm_tile(0)
    ref = a.float() @ w.float().t()
    err = ((y.float() - ref).norm() / ref.norm()).item()
    print(f"w4 {M}x{N}x{K}: rel-L2 {err:.3e}", flush=True)
    assert err < 5e-3
rows = []
for rep in range(3):
    for mode in ("pp", "w4", "default"):
        lib.rf_debug_gemm_w4_knock(0)
        lib.rf_debug_force_gemm_tile(258 if mode == "w4" else 0)
        lib.rf_debug_force_gemm_sk(0 if mode == "pp" else -1)
        rows.append((mode, bench.isolated_shapes(dev, 512, 4096, 3072, 12288, 24, 19, 38)))
lib.rf_debug_force_gemm_tile(0); lib.rf_debug_force_gemm_sk(-1)
for name in rows[0][1]:
    line = [f"{name:8s}"]
    for mode, r in rows:
        line.append(f"{mode} {r[name]['us']:7.1f} us {r[name]['tflops']:6.1f}")
    print(" | ".join(line), flush=True)
