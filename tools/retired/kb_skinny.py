"""LoRA down-projection x . lora_A^T (N = r_pad = 64, M = condition rows): skinny-N kernel vs the split-K route."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0"); BF = torch.bfloat16
for (M, N, K, K2) in ((1024, 64, 3072, 0), (1024, 64, 12288, 0), (1024, 64, 3072, 12288), (4096, 64, 3072, 0), (16384, 64, 3072, 0)):
    x = torch.randn(M, K, device=dev).to(BF); A = (torch.randn(N, K + K2, device=dev) * 0.05).to(BF)
    segs = [ops.Seg(x, A[:, :K])]
    if K2:
        x2 = torch.randn(M, K2, device=dev).to(BF); segs.append(ops.Seg(x2, A[:, K:]))
    y = torch.empty(M, 256, device=dev, dtype=BF)
    g = [ops.Group(segs, out=y[:, :N])]
    line = [f"M={M} N={N} K={K}+{K2}"]
    for sk in (0, 1):
        lib.rf_debug_gemm_skinny(sk)
        t = min(ops.time_gemm(g, N, iters=20) for _ in range(3))
        line.append(f"{'skinny' if sk else 'split-K'} {t*1e6:7.1f} us (path {lib.rf_debug_last_gemm_path()})")
    print(" | ".join(line), flush=True)
lib.rf_debug_gemm_skinny(0)
