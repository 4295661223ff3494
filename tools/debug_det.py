#!/usr/bin/env python3
"""Run-to-run bitwise determinism + finiteness of each kernel at the full FLUX shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import ops
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_QKV, RF_EPI_QKV_GELU, RF_EPI_STORE, Group, Seg
dev = torch.device("cuda:0"); bf = torch.bfloat16
S_txt, S_img, D, mlp, H = 512, 4096, 3072, 12288, 24
S = S_txt + S_img
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(bf)
xn, att, hid, x0 = r(S, D), r(S, D), r(S, mlp), r(S, D)
gate, b3, b1, bm, bfu = r(D), r(3 * D), r(D), r(mlp), r(3 * D + mlp)
Wq, Wq2, Wo, Wo2 = r(3 * D, D, sc=.02), r(3 * D, D, sc=.02), r(D, D, sc=.02), r(D, D, sc=.02)
W1, W1b, W2, W2b = r(mlp, D, sc=.02), r(mlp, D, sc=.02), r(D, mlp, sc=.02), r(D, mlp, sc=.02)
Wf, Ws = r(3 * D + mlp, D, sc=.02), r(D, D + mlp, sc=.02)
nw = [r(128) for _ in range(4)]
cos, sin = torch.rand(S, 128, device=dev), torch.rand(S, 128, device=dev)
t, i = slice(0, S_txt), slice(S_txt, S)

def run_qkv():
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    ops.gemm([Group([Seg(xn[t], Wq2)], bias=b3, tok_offset=0, norm_q=nw[2], norm_k=nw[3]),
              Group([Seg(xn[i], Wq)], bias=b3, tok_offset=S_txt, norm_q=nw[0], norm_k=nw[1])], 3 * D, RF_EPI_QKV,
             q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin))
    return torch.cat([q.flatten(), k.flatten(), vt.flatten()])
def run_out():
    x = x0.clone()
    ops.gemm([Group([Seg(att[t], Wo2)], bias=b1, gate=gate, out=x[t], residual=x[t]),
              Group([Seg(att[i], Wo)], bias=b1, gate=gate, out=x[i], residual=x[i])], D, RF_EPI_GATE_RES)
    return x
def run_ff1():
    h = torch.empty_like(hid)
    ops.gemm([Group([Seg(xn[t], W1b)], bias=bm, out=h[t]), Group([Seg(xn[i], W1)], bias=bm, out=h[i])], mlp, RF_EPI_GELU)
    return h
def run_ff2():
    x = x0.clone()
    ops.gemm([Group([Seg(hid[t], W2b)], bias=b1, gate=gate, out=x[t], residual=x[t]),
              Group([Seg(hid[i], W2)], bias=b1, gate=gate, out=x[i], residual=x[i])], D, RF_EPI_GATE_RES)
    return x
def run_sgl_in():
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    h = torch.empty_like(hid)
    ops.gemm([Group([Seg(xn, Wf)], bias=bfu, out=h, tok_offset=0, norm_q=nw[0], norm_k=nw[1])], 3 * D + mlp, RF_EPI_QKV_GELU,
             n_split=3 * D, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin))
    return torch.cat([q.flatten(), k.flatten(), vt.flatten(), h.flatten()])
def run_sgl_out():
    x = x0.clone()
    ops.gemm([Group([Seg(att, Ws[:, :D]), Seg(hid, Ws[:, D:])], bias=b1, gate=gate, out=x, residual=x)], D, RF_EPI_GATE_RES)
    return x
qa, ka, vta, _ = ops.alloc_attn_operands(H, S, dev)
qa.normal_(); ka.normal_(); vta.normal_()
def run_attn():
    return ops.attention(qa, ka, vta, S)
sc_, sh_ = r(D), r(D)
def run_ln():
    return ops.layernorm_modulate(x0, sc_, sh_)
def run_small():
    lat = r(S_img, 64); Wx = r(D, 64, sc=.1); bx = r(D)
    a = ops.linear(lat, Wx, bx)
    Wp = r(64, D, sc=.02); bp = r(64)
    b = ops.linear(xn[i], Wp, bp)
    s1 = r(50, D); Wm = r(6 * D, D, sc=.02); bmm = r(6 * D)
    tab = torch.empty(50, 8 * D, device=dev, dtype=bf)
    ops.linear(s1, Wm, bmm, out=tab[:, D:7 * D])
    return torch.cat([a.flatten(), b.flatten(), tab[:, D:7 * D].flatten()])
for name, fn in [("qkv", run_qkv), ("out", run_out), ("ff1", run_ff1), ("ff2", run_ff2), ("sgl_in", run_sgl_in),
                 ("sgl_out", run_sgl_out), ("attn", run_attn), ("ln", run_ln), ("small", run_small)]:
    torch.manual_seed(0)
    outs = []
    for rep in range(6):
        torch.manual_seed(123)
        outs.append(fn().clone())
        torch.cuda.synchronize()
    nd = [int((o != outs[0]).sum()) for o in outs[1:]]
    fin = all(bool(torch.isfinite(o.float()).all()) for o in outs)
    print(f"{name:8s} finite={fin} differing elements vs run0: {nd}", flush=True)
