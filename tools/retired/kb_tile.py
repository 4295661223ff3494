"""A/B of GEMM tile configurations through rf_debug_force_gemm_tile.  Usage: PYTHONPATH=. python tools/kb_tile.py 256 257"""
import sys, torch
from reflectionflow_amd import _lib, ops
_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
dev = torch.device("cuda:0"); lib = _lib.load(); BF = torch.bfloat16
tiles = [int(a) for a in sys.argv[1:]] or [256]
lib.rf_debug_force_gemm_sk(0)
G = ops.RF_EPI_GATE_RES
shapes = [("8192^3", (8192,), 8192, 8192, 0), ("qkv", (512, 4096), 9216, 3072, 0), ("out", (512, 4096), 3072, 3072, G),
          ("ff1", (512, 4096), 12288, 3072, ops.RF_EPI_GELU), ("ff2", (512, 4096), 3072, 12288, G), ("sgl_in", (4608,), 21504, 3072, 0),
          ("sgl_out", (4608,), 3072, 15360, G)]
for name, rows, N, K, epi in shapes:
    groups = []
    for M in rows:
        x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF) * 0.02
        kw = dict(residual=torch.randn(M, N, device=dev, dtype=BF), gate=torch.randn(N, device=dev, dtype=BF)) if epi == G else {}
        groups.append(ops.Group([ops.Seg(x, W)], bias=torch.randn(N, device=dev, dtype=BF), out=torch.empty(M, N, device=dev, dtype=BF), **kw))
    fl = 2.0 * sum(rows) * N * K
    outs, line = {}, f"{name:8s}"
    for t in tiles:
        lib.rf_debug_force_gemm_tile(t)
        s = min(ops.time_gemm(groups, N, epi, iters=20) for _ in range(3))
        outs[t] = [g.out.clone() for g in groups]
        line += f"  t{t}: {s*1e6:7.1f}us {fl/s/1e12:6.0f}TF"
    ok = all(torch.equal(a, b) for t in tiles[1:] for a, b in zip(outs[tiles[0]], outs[t]))
    print(line, " same" if ok else " DIFF", flush=True)
lib.rf_debug_force_gemm_tile(0)
