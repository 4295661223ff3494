"""Epilogue cost: time one-round GEMMs (216 tiles = 4608 x 3072 outputs) at K = 64..512 and extrapolate to K = 0."""
import torch
from reflectionflow_amd import _lib, ops
_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
dev = torch.device("cuda:0"); lib = _lib.load(); BF = torch.bfloat16
lib.rf_debug_force_gemm_sk(0)
M, N = 4608, 3072
H = N // 3 // 128
for name, epi in [("STORE", ops.RF_EPI_STORE), ("GELU", ops.RF_EPI_GELU), ("GATE_RES", ops.RF_EPI_GATE_RES), ("QKV+norm+rope", ops.RF_EPI_QKV)]:
    ts = []
    for K in (64, 128, 256, 512, 1024):
        x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF) * 0.05
        b = torch.randn(N, device=dev, dtype=BF)
        kw, gkw = {}, {}
        if epi == ops.RF_EPI_GATE_RES:
            gkw = dict(residual=torch.randn(M, N, device=dev, dtype=BF), gate=torch.randn(N, device=dev, dtype=BF))
        if epi == ops.RF_EPI_QKV:
            q, k, vt, s_pad = ops.alloc_attn_operands(H, M, dev)
            cos, sin = torch.rand(M, 128, device=dev), torch.rand(M, 128, device=dev)
            nq = torch.ones(128, device=dev, dtype=BF)
            kw = dict(q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE)
            gkw = dict(norm_q=nq, norm_k=nq)
            g = [ops.Group([ops.Seg(x, W)], bias=b, **gkw)]
        else:
            g = [ops.Group([ops.Seg(x, W)], bias=b, out=torch.empty(M, N, device=dev, dtype=BF), **gkw)]
        t = min(ops.time_gemm(g, N, epi, iters=50, **kw) for _ in range(3))
        ts.append((K, t * 1e6))
    (k1, t1), (k2, t2) = ts[-2], ts[-1]
    slope = (t2 - t1) / (k2 - k1) * 64
    print(f"{name:14s} " + "  ".join(f"K={k}: {t:6.1f}us" for k, t in ts) + f"   per K-tile {slope:.2f}us  => K=0 intercept {t2 - slope * k2 / 64:.1f}us", flush=True)
