"""A/B of the evenly loaded phases inside the stream-K kernel (cfg4 / cfg5 shapes that take stream-K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0"); BF = torch.bfloat16
G = ops.RF_EPI_GATE_RES
for name, rows, N, K, epi in (("cfg4 out", (512, 4096, 1024), 3072, 3072, G), ("cfg4 ff2", (512, 4096, 1024), 3072, 12288, G), ("cfg4 qkv", (512, 4096, 1024), 9216, 3072, ops.RF_EPI_STORE),
                              ("cfg4 sgl_out", (4608, 1024), 3072, 15360, G), ("cfg5 out", (512, 16384, 1024), 3072, 3072, G), ("cfg5 ff2", (512, 16384, 1024), 3072, 12288, G)):
    groups = []
    for M in rows:
        x = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * 0.02).to(BF); b = torch.randn(N, device=dev).to(BF)
        kw = dict(residual=torch.randn(M, N, device=dev).to(BF), gate=torch.randn(N, device=dev).to(BF)) if epi == G else {}
        groups.append(ops.Group([ops.Seg(x, W)], bias=b, out=torch.empty(M, N, device=dev, dtype=BF), **kw))
    line = [f"{name:14s}"]
    for rep in range(2):
        for ev in (0, 1):
            lib.rf_debug_gemm_even(ev)
            t = min(ops.time_gemm(groups, N, epi, iters=10) for _ in range(3))
            line.append(f"{'even' if ev else '8/4/8/4'} {t*1e6:7.1f} us (path {lib.rf_debug_last_gemm_path()})")
    print(" | ".join(line), flush=True)
lib.rf_debug_gemm_even(1)
