"""T5-XXL B = 1 (512 tokens) GEMM shapes under every schedule: which one streams the 9.4 GB of weights fastest?"""
import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)
names = {L.RF_SCHED_AUTO: "auto", L.RF_SCHED_TILE128: "tile128", L.RF_SCHED_TILE256: "tile256", L.RF_SCHED_STREAMK: "streamk", L.RF_SCHED_W4: "w4"}
tot = {k: 0.0 for k in names}
for tag, M, N, K in [("qk", 512, 8192, 4096), ("v^T", 4096, 512, 4096), ("o", 512, 4096, 4096), ("wi", 512, 20480, 4096), ("wo", 512, 4096, 10240)]:
    x, W = r(M, K), r(N, K, sc=0.02)
    out = torch.empty(M, N, device=dev, dtype=BF)
    g = [ops.Group([ops.Seg(x, W)], out=out)]
    line = f"{tag:4s} {M}x{N}x{K}  weights {N * K * 2 / 1e6 if tag != 'v^T' else M * K * 2 / 1e6:6.0f} MB "
    for sched, nm in names.items():
        try:
            sec = min(ops.time_gemm(g, N, iters=20, schedule=sched) for _ in range(3))
            tot[sched] += sec
            line += f" {nm} {sec * 1e6:6.1f}us"
        except Exception as e:
            line += f" {nm} n/a"
            tot[sched] += 1.0
    print(line, flush=True)
print("per layer:", {names[k]: round(v * 1e6, 1) for k, v in tot.items()}, " x24 ms:", {names[k]: round(v * 24e3, 2) for k, v in tot.items()})
