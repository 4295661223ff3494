#!/usr/bin/env python3
"""One warm-up + ONE measured launch of each dominant-kernel shape (the 6 GEMM launches of a 1024^2 forward)
and of attention, in a fixed order, for `rocprofv3 --pmc` passes (tools/pmc_collect.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import ops
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_QKV, RF_EPI_QKV_GELU, Group, Seg
dev = torch.device("cuda:0"); bf = torch.bfloat16
S_txt, S_img, D, mlp, H = 512, 4096, 3072, 12288, 24
S = S_txt + S_img
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(bf)
xn, att, hid, x = r(S, D), r(S, D), r(S, mlp), r(S, D)
gate, b3, b1, bm, bfu = r(D), r(3 * D), r(D), r(mlp), r(3 * D + mlp)
Wq, Wq2, Wo, Wo2 = r(3 * D, D, sc=.02), r(3 * D, D, sc=.02), r(D, D, sc=.02), r(D, D, sc=.02)
W1, W1b, W2, W2b = r(mlp, D, sc=.02), r(mlp, D, sc=.02), r(D, mlp, sc=.02), r(D, mlp, sc=.02)
Wf, Ws = r(3 * D + mlp, D, sc=.02), r(D, D + mlp, sc=.02)
nw = [r(128) for _ in range(4)]
cos, sin = torch.rand(S, 128, device=dev), torch.rand(S, 128, device=dev)
q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
t, i = slice(0, S_txt), slice(S_txt, S)
def dbl_qkv(): ops.gemm([Group([Seg(xn[t], Wq2)], bias=b3, tok_offset=0, norm_q=nw[2], norm_k=nw[3]), Group([Seg(xn[i], Wq)], bias=b3, tok_offset=S_txt, norm_q=nw[0], norm_k=nw[1])], 3 * D, RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin))
def dbl_out(): ops.gemm([Group([Seg(att[t], Wo2)], bias=b1, gate=gate, out=x[t], residual=x[t]), Group([Seg(att[i], Wo)], bias=b1, gate=gate, out=x[i], residual=x[i])], D, RF_EPI_GATE_RES)
def dbl_ff1(): ops.gemm([Group([Seg(xn[t], W1b)], bias=bm, out=hid[t]), Group([Seg(xn[i], W1)], bias=bm, out=hid[i])], mlp, RF_EPI_GELU)
def dbl_ff2(): ops.gemm([Group([Seg(hid[t], W2b)], bias=b1, gate=gate, out=x[t], residual=x[t]), Group([Seg(hid[i], W2)], bias=b1, gate=gate, out=x[i], residual=x[i])], D, RF_EPI_GATE_RES)
def sgl_in(): ops.gemm([Group([Seg(xn, Wf)], bias=bfu, out=hid, tok_offset=0, norm_q=nw[0], norm_k=nw[1])], 3 * D + mlp, RF_EPI_QKV_GELU, n_split=3 * D, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin))
def sgl_out(): ops.gemm([Group([Seg(att, Ws[:, :D]), Seg(hid, Ws[:, D:])], bias=b1, gate=gate, out=x, residual=x)], D, RF_EPI_GATE_RES)
for fn in (dbl_qkv, dbl_out, dbl_ff1, dbl_ff2, sgl_in, sgl_out):
    fn(); torch.cuda.synchronize()      # warm-up (dispatch 1 of the pair)
    fn(); torch.cuda.synchronize()      # measured (dispatch 2 of the pair)
# attention as the engine calls it: prescaled q, proven score bound -> bounded-score kernel (v4); fresh random operands
# (the QKV launches above overwrote q / k / vt)
q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
BOUND = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
for _ in range(2):
    ops.attention(q, k, vt, S, out=att, q_prescaled=True, score_bound=BOUND); torch.cuda.synchronize()
# round 3: the lagged-max form (no bound: what a checkpoint with qk_bound > 100 runs)
for _ in range(2):
    ops.attention(q, k, vt, S, out=att, q_prescaled=True, score_bound=0.0); torch.cuda.synchronize()
print("done")
