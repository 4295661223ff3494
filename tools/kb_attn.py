import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = _lib.load()
for S in (4608, 5632):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_()
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    for rep in range(3):
        line = []
        outs = []
        for v in (0, 1):
            lib.rf_debug_attn_v2(v)
            t = timeit(lambda: ops.attention(q, k, vt, S, out=out), 10)
            outs.append(out.clone())
            line.append(f"v{v+1}: {t*1e6:7.1f} us {4.0*S*S*H*128/t/1e12:6.1f} TF")
        print(f"S={S}", " | ".join(line), " maxdiff", float((outs[0].float()-outs[1].float()).abs().max()), flush=True)
lib.rf_debug_attn_v2(-1)
