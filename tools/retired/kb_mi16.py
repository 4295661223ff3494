"""A/B: 256x256 balanced ping-pong GEMM on v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 (rf_debug_gemm_mi16),
cfg2 launch shapes with their fused epilogues + bit-level agreement of the two on every epilogue."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
import bench
from reflectionflow_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib.rf_debug_force_gemm_sk(0)
# correctness vs fp32 on plain shapes (K-tile counts 1, 2, 3, 5, 48; ragged M / N; two segments)
for (M, N, K) in ((512, 512, 64), (512, 512, 128), (512, 512, 192), (300, 520, 320), (4608, 3072, 3072)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    ref = a.float() @ w.float().t()
    ys = []
    for mi in (0, 1):
        lib.rf_debug_gemm_mi16(mi)
        y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm([ops.Group([ops.Seg(a, w)], out=y)], N, splitk_ws=False)
        torch.cuda.synchronize()
        ys.append(y)
        err = ((y.float() - ref).norm() / ref.norm()).item()
        assert err < 5e-3, (M, N, K, mi, err)
    d = (ys[0].float() - ys[1].float()).abs().max().item()
    print(f"{M}x{N}x{K}: both within 5e-3 of fp32; max |32x32 - 16x16| = {d:.3e}", flush=True)
rows = []
for rep in range(3):
    for mi in (0, 1):
        lib.rf_debug_gemm_mi16(mi)
        rows.append((mi, bench.isolated_shapes(dev, 512, 4096, 3072, 12288, 24, 19, 38)))
lib.rf_debug_gemm_mi16(1); lib.rf_debug_force_gemm_sk(-1)
for name in rows[0][1]:
    line = [f"{name:8s}"]
    for mi, r in rows:
        line.append(f"{'16x16x32' if mi else '32x32x16'} {r[name]['us']:7.1f} us {r[name]['tflops']:6.1f} TF @{r[name].get('shader_mhz', 0):5d}")
    print(" | ".join(line), flush=True)
