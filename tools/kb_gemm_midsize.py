"""Mid-size GEMMs (fewer 256x256 tiles than CUs): 128x128 tiles vs 256x256 stream-K vs plain 256x256, per shape.
    python tools/kb_gemm_midsize.py      (profiles/r04_gemm_midsize.md)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as L, ops
L.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


D = 3072
shapes = [(M, N, K) for M in (2560, 1536, 1024, 3584) for (N, K) in ((D, D), (3 * D, D), (4 * D, D), (D, 4 * D), (7 * D, D), (D, 5 * D))]
for M, N, K in shapes:
    x, W, b = r(M, K), r(N, K, sc=0.02), r(N)
    y = torch.empty(M, N, device=dev, dtype=BF)
    g = [ops.Group([ops.Seg(x, W)], bias=b, out=y)]
    res = {}
    for name, sched in (("auto", 0), ("tile128", 1), ("tile256", 2), ("streamk", 3)):
        us = timeit(lambda: ops.gemm(g, N, ops.RF_EPI_STORE, schedule=sched))
        res[name] = round(us, 1)
    t256 = -(-M // 256) * -(-N // 256)
    best = min(res, key=res.get)
    print(json.dumps({"M": M, "N": N, "K": K, "tiles256": t256, "us": res, "best": best, "tflops_best": round(2.0 * M * N * K / res[best] / 1e6, 1)}), flush=True)
