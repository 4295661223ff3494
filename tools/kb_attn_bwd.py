"""Timing of the attention backward's two kernels per form at the training shapes (A/B tool).  python tools/kb_attn_bwd.py"""
import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
from reflectionflow_amd.train import kernels as K
from oracle import flux_oracle as O
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for H, S in ((24, 5632), (24, 2560)):
    raw = (torch.randn(S, 3 * H * 128, generator=g, device=dev) * 1.5).to(BF)
    w1 = torch.ones(128, device=dev).to(BF)
    ids = torch.stack([torch.zeros(S), torch.arange(S) // 64, torch.arange(S) % 64], 1).to(dev)
    cos, sin = (t.contiguous() for t in O.FluxPosEmbed(10000, (16, 56, 56))(ids))
    a = K.qkv_train_fwd(raw, H, 0, (w1, w1, None, None), cos, sin)
    lse = torch.empty(H, a.s_pad, dtype=torch.float32, device=dev)
    out = ops.attention(a.q, a.k, a.vt, S, q_prescaled=True, lse=lse)
    dout = torch.randn(S, H * 128, generator=g, device=dev).to(BF)
    line = [f"H={H} S={S}"]
    for nm, kern in (("auto", 0), ("dq256+dkv128", L.RF_ATTN_BWD_DQ_256 | L.RF_ATTN_BWD_DKV_128), ("dq128+dkv192", L.RF_ATTN_BWD_DQ_128 | L.RF_ATTN_BWD_DKV_192),
                     ("dq192+dkv128x2", L.RF_ATTN_BWD_DQ_192 | L.RF_ATTN_BWD_DKV_128X2)):
        f = lambda: K.attention_bwd(a, out, dout, lse=lse.clone(), kernel=kern)   # noqa
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): f()
        e1.record(); torch.cuda.synchronize()
        line.append(f"{nm} {e0.elapsed_time(e1) / 8 * 1e3:7.1f} us")
    print(" | ".join(line), flush=True)
