"""Round-3 attention A/B on one GPU (experiments library): the shipped bounded / lagged kernels vs (a) fragment reads two groups
ahead, (b) the old 8-byte epilogue stores, (c) the mixed-size launch (256- and 192-query workgroups), at the BASELINE lengths.
Interleaved repetitions; shader clock of block 0's main loop from the kernels' own probe."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib as L, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = L.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1)
for S in (4608, 5632, 17920):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    ref = None
    for rep in range(3):
        line = []
        for name, kern, knock in (("bounded", L.RF_ATTN_BOUNDED16, 0), ("ahead2", L.RF_ATTN_BOUNDED16, 4), ("narrow-st", L.RF_ATTN_BOUNDED16, 8),
                                  ("ahead2+narrow", L.RF_ATTN_BOUNDED16, 12), ("mix", L.RF_ATTN_BOUNDED16_MIX, 0),
                                  ("lagged", L.RF_ATTN_LAGGED16, 0), ("lag-mix", L.RF_ATTN_LAGGED16_MIX, 0), ("split", L.RF_ATTN_BOUNDED16_SPLIT, 0)):
            lib.rf_debug_attn_knock(knock)
            out.zero_()
            f = lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=kern)
            t = timeit(f, 10 if S < 10000 else 4)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            d = float((out.float() - ref.float()).abs().max())
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(1, C.byref(mhz), C.byref(us))
            line.append(f"{name} {t*1e6:7.1f}us {4.0*S*S*H*128/t/1e12:6.0f}TF @{mhz.value:4.0f} d={d:.0e}")
        print(f"S={S} | " + " | ".join(line), flush=True)
lib.rf_debug_attn_knock(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1)
