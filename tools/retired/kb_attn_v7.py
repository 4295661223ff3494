"""v7 (ping-pong schedule) vs v5 of the 16x16x32 attention kernels (experiments library): bounded / lagged, plain and mixed-size
grids, interleaved repetitions; bit-equality of the outputs; shader clock of block 0's loop."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib as L, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = L.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1)
for S in (4608, 5632, 2048):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    ref = {}
    for rep in range(3):
        line = []
        for name, kern in (("bounded", L.RF_ATTN_BOUNDED16), ("bounded-mix", L.RF_ATTN_BOUNDED16_MIX), ("lagged", L.RF_ATTN_LAGGED16), ("lagged-mix", L.RF_ATTN_LAGGED16_MIX)):
            for v7 in (0, 1):
                lib.rf_debug_attn_v7(v7)
                out.zero_()
                f = lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=kern)
                t = timeit(f, 10)
                torch.cuda.synchronize()
                key = name.split("-")[0]
                if key not in ref:
                    ref[key] = out.clone()
                same = bool(torch.equal(out, ref[key]))
                mhz, us = C.c_double(0), C.c_double(0)
                lib.rf_debug_clock_probe(1, C.byref(mhz), C.byref(us))
                line.append(f"{name} v{7 if v7 else 5} {t*1e6:6.1f}us {4.0*S*S*H*128/t/1e12:5.0f}TF @{mhz.value:4.0f} {'==' if same else '!='}")
        print(f"S={S} | " + " | ".join(line), flush=True)
lib.rf_debug_attn_v7(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1)
