"""Phase stamps of the ping-pong attention schedule (v7, experiments library): s_memtime of block 0's eight waves at the nine phase
boundaries of key tile 40 (bounded kernel, S = 4608).  Phases: O4' M1 O1 M2 O2 M3 O3 M4; group B (waves 4-7) runs one phase behind."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); lib = L.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1); lib.rf_debug_attn_v7(1)
S, H = 4608, 24
q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
names = ["bar", "O4'", "bar+M1", "bar+O1", "bar+M2", "bar+O2", "bar+M3", "bar+O3"]
for label, kn in (("v7", 16), ("v7, no DMA after tile 3 (knock-out)", 48), ("v7, no exp2 (knock-out)", 80)):
    lib.rf_debug_attn_knock(kn)
    for _ in range(5):
        ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=L.RF_ATTN_BOUNDED16)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    lib.rf_debug_attn_stamps7(buf)
    base = buf[0]
    print(label + ": clocks between consecutive stamps of key tile 40 (stamp i sits right after barrier i; an interval = the phase + its closing barrier wait):")
    print("          " + "  ".join(f"{n:>7s}" for n in names) + "   | total | start vs wave 0")
    for w in range(8):
        st = [buf[w * 16 + i] for i in range(9)]
        d = [int(st[i + 1]) - int(st[i]) for i in range(8)]
        x = [int(buf[w * 16 + i]) for i in range(14)]
        print(f"  wave {w}: " + "  ".join(f"{v:7d}" for v in d) + f"   | {int(st[8]) - int(st[0]):5d} | {int(st[0]) - int(base):+d}"
              f"   || M2 issue {x[9] - x[4]:4d} + barrier {x[5] - x[9]:4d} | O2: reads {x[10] - x[5]:4d}, exp2+sums {x[11] - x[10]:4d}, tail {x[12] - x[11]:4d}, barrier {x[6] - x[12]:4d}"
              f" | M3 issue {x[13] - x[6]:4d} + barrier {x[7] - x[13]:4d}", flush=True)
lib.rf_debug_attn_knock(0); lib.rf_debug_attn_v7(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1)
