"""Timing of the 8-wave GEMM (TILE256) on three cfg2 shapes and of the attention forward at S = 4608 x 24 heads (knock-out A/B tool)."""
import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa
for rep in range(3):
    line = []
    for name, M, N, K in (("dbl_ff1", 4608, 12288, 3072), ("sgl_in", 4608, 21504, 3072), ("dbl_ff2", 4608, 3072, 12288)):
        x, W = r(M, K), r(N, K, sc=0.02)
        out = torch.empty(M, N, dtype=BF, device=dev)
        t = ops.time_gemm([ops.Group([ops.Seg(x, W)], out=out)], N, iters=20, schedule=L.RF_SCHED_TILE256)
        line.append(f"{name} {2.0 * M * N * K / t / 1e12:7.1f} TF")
    S, H = 4608, 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    out = torch.empty(S, H * 128, device=dev, dtype=BF)
    for nm, kern, bound in (("auto", None, 30.0), ("lagged16_mix", L.RF_ATTN_LAGGED16_MIX, 0.0), ("bounded16", L.RF_ATTN_BOUNDED16, 30.0)):
        f = lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=kern)   # noqa
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line.append(f"attn {nm} {us:6.1f} us ({4.0 * S * S * 128 * H / us / 1e6:6.1f} TF)")
    print(" | ".join(line), flush=True)
