"""A/B timing of the 256x256 tile-per-block GEMM on the block shapes, one library per process:
    python tools/kb_gemm_epilogue_ab.py [--lib reflectionflow_amd/librf_flux_base.so] [--sched 2]
(run it alternately with and without --lib on the same box; profiles/r04_gemm_direct_epilogue.md)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--sched", type=int, default=2)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
from reflectionflow_amd import _lib as L
if a.lib:
    L.LIB_PATH = os.path.abspath(a.lib)
from reflectionflow_amd import ops
L.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)
S, D, H = 4608, 3072, 24


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters


out = {}
ws = ops.alloc_splitk_ws(dev) if hasattr(ops, "alloc_splitk_ws") else None
for name, M, N, K, epi in [("store_qkv", S, 3 * D, D, ops.RF_EPI_STORE), ("gate_res_out", S, D, D, ops.RF_EPI_GATE_RES), ("gelu_ff_up", S, 4 * D, D, ops.RF_EPI_GELU),
                           ("gate_res_ff_dn", S, D, 4 * D, ops.RF_EPI_GATE_RES), ("store_sgl_in", S, 7 * D, D, ops.RF_EPI_STORE), ("gate_res_sgl_out", S, D, 5 * D, ops.RF_EPI_GATE_RES)]:
    x, W, b = r(M, K), r(N, K, sc=0.02), r(N)
    y = torch.empty(M, N, device=dev, dtype=BF)
    kw = dict(residual=r(M, N), gate=r(N)) if epi == ops.RF_EPI_GATE_RES else {}
    g = [ops.Group([ops.Seg(x, W)], bias=b, out=y, **kw)]
    us = timeit(lambda: ops.gemm(g, N, epi, schedule=a.sched, splitk_ws=(False if a.sched in (2, 6) else None)))
    out[name] = (round(us, 1), round(2.0 * M * N * K / us / 1e6, 1))
# fused QKV + RMSNorm + RoPE epilogue
x, W, b = r(S, D), r(3 * D, D, sc=0.02), r(3 * D)
nq, nk = r(128), r(128)
cos, sin = torch.randn(S, 128, device=dev), torch.randn(S, 128, device=dev)
q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
g = [ops.Group([ops.Seg(x, W)], bias=b, tok_offset=0, norm_q=nq, norm_k=nk)]
us = timeit(lambda: ops.gemm(g, 3 * D, ops.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin), q_scale=ops.QK_PRESCALE, schedule=a.sched, splitk_ws=(False if a.sched in (2, 6) else None)))
out["qkv_rope"] = (round(us, 1), round(2.0 * S * 3 * D * D / us / 1e6, 1))
print(json.dumps({"lib": os.path.basename(L.LIB_PATH), "sched": a.sched, "us_tf": out}))
