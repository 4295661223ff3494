"""A/B of the 256 x 256 GEMM loops on the cfg2 block shapes (same box, interleaved): AUTO (8-wave ping-pong / stream-K), W4, W4B and
hipBLASLt through torch.matmul.  python tools/kb_w4b.py [--iters 20] [--reps 3]"""
import argparse
import json
import sys

import torch

sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--scheds", default="auto,w4,w4b")
args = ap.parse_args()
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa
SCHED = {"auto": L.RF_SCHED_AUTO, "t256": L.RF_SCHED_TILE256, "w4": L.RF_SCHED_W4, "w4b": L.RF_SCHED_W4B, "sk": L.RF_SCHED_STREAMK,
         "persist": L.RF_SCHED_PERSISTENT}
shapes = [("dbl_qkv", 4608, 9216, 3072), ("dbl_out", 4608, 3072, 3072), ("dbl_ff1", 4608, 12288, 3072), ("dbl_ff2", 4608, 3072, 12288),
          ("sgl_in", 4608, 21504, 3072), ("sgl_out", 4608, 3072, 15360), ("sq8192", 8192, 8192, 8192)]
res = {}
for name, M, N, K in shapes:
    x, W, b = r(M, K), r(N, K, sc=0.02), r(N)
    out = torch.empty(M, N, dtype=BF, device=dev)
    grp = [ops.Group([ops.Seg(x, W)], bias=b, out=out)]
    fl = 2.0 * M * N * K
    row = {}
    ref = None
    for rep in range(args.reps):
        for sname in args.scheds.split(","):
            t = ops.time_gemm(grp, N, iters=args.iters, schedule=SCHED[sname])
            row.setdefault(sname, []).append(fl / t / 1e12)
            if rep == 0:
                o = out.clone()
                if ref is None:
                    ref = o
                elif sname in ("w4", "w4b", "t256", "sk", "persist"):
                    row[sname + "_bit_equal_to_first"] = bool(torch.equal(o, ref))
        # hipBLASLt
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.matmul(x, W.t())
        e0.record()
        for _ in range(args.iters):
            torch.matmul(x, W.t())
        e1.record()
        torch.cuda.synchronize()
        row.setdefault("hipblaslt", []).append(fl / (e0.elapsed_time(e1) / args.iters * 1e-3) / 1e12)
    res[name] = {k: ([round(v, 1) for v in vs] if isinstance(vs, list) else vs) for k, vs in row.items()}
    print(name, json.dumps(res[name]), flush=True)
