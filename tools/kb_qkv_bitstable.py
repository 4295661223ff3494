"""Round 5: the fused QKV GEMM (RMSNorm + RoPE epilogue) at 512 text + 256 image rows: is it bit-reproducible, per schedule?"""
import sys
import torch
sys.path.insert(0, ".")
from reflectionflow_amd import ops, _lib as L
from oracle import flux_oracle as O
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa
D, H = 3072, 24
lib = L.load()
for St, Si in ((512, 256), (512, 400), (0, 256), (256, 256), (512, 128)):
    S = St + Si
    s_pad = (S + 63) // 64 * 64
    xt, xi = r(max(St, 1), D), r(Si, D)
    Wt, Wi = r(3 * D, D, sc=0.02), r(3 * D, D, sc=0.02)
    bt, bi = r(3 * D), r(3 * D)
    nq, nk = (1 + 0.02 * torch.randn(128, device=dev)).to(BF), (1 + 0.02 * torch.randn(128, device=dev)).to(BF)
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 32, torch.arange(S) % 32], 1).to(dev)
    cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
    for sched, nm in ((L.RF_SCHED_AUTO, "AUTO"), (L.RF_SCHED_TILE128, "TILE128"), (L.RF_SCHED_TILE256, "TILE256")):
        for fused in (True, False):
            outs = []
            for _ in range(12):
                q = torch.zeros(H, s_pad, 128, dtype=BF, device=dev)
                k, vt = torch.zeros_like(q), torch.zeros_like(q)
                groups = []
                if St:
                    groups.append(ops.Group([ops.Seg(xt, Wt)], bias=bt, norm_q=nq, norm_k=nk))
                groups.append(ops.Group([ops.Seg(xi, Wi)], bias=bi, norm_q=nq, norm_k=nk, tok_offset=St))
                d = ops.build_gemm_desc(groups, 3 * D, L.RF_EPI_QKV, q=q, k=k, vt=vt, heads=H, s_pad=s_pad, rope=(cos, sin) if fused else None,
                                        q_scale=ops.QK_PRESCALE, splitk_ws=ops.splitk_scratch(dev), schedule=sched)
                L.check(lib.rf_gemm_bf16(__import__("ctypes").byref(d), ops.stream_ptr()), "gemm")
                torch.cuda.synchronize()
                outs.append((q, k, vt))
            bad = [i for i in range(1, 12) if not all(torch.equal(a, b) for a, b in zip(outs[i], outs[0]))]
            nq_ = sum(int((outs[i][0] != outs[0][0]).sum()) for i in bad)
            print(f"S={St}+{Si} {nm:8s} fused_rope={fused}: path {lib.rf_debug_last_gemm_path()} differing runs {bad if bad else 'none'} (q elems {nq_})", flush=True)
