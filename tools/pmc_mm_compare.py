#!/usr/bin/env python3
"""hipBLASLt (torch.matmul) vs rf_gemm_bf16 on the same operands, one warm-up + one measured launch each, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE / MFMA busy): does the library kernel move fewer bytes over the fabric for the same product?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(bf)
for M, N, K in [(4608, 21504, 3072), (4608, 12288, 3072), (4608, 3072, 12288), (8192, 8192, 8192)]:
    x, W = r(M, K), r(N, K, sc=.02)
    out = torch.empty(M, N, device=dev, dtype=bf)
    g = [ops.Group([ops.Seg(x, W)], out=out)]
    for fn in (lambda: torch.matmul(x, W.t(), out=out), lambda: ops.gemm(g, N, schedule=L.RF_SCHED_TILE256, splitk_ws=False),
               lambda: ops.gemm(g, N, schedule=L.RF_SCHED_W4, splitk_ws=False)):
        fn(); torch.cuda.synchronize()
        fn(); torch.cuda.synchronize()
    del x, W, out
