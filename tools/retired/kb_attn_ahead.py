"""Fragment-read lookahead x rotated waves (experiments library), bounded kernel, S = 4608 / 17920: timing only."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib as L, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = L.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1)
for S in (4608, 17920):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=L.RF_ATTN_BOUNDED16)
    ref = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_ONLINE256).float()
    for rep in range(3):
        line = []
        for name, kn in (("rot a1", 4), ("rot a2", 0), ("rot a3", 32), ("rot a4", 36), ("norot a1", 68), ("norot a2", 64), ("norot a3", 96), ("norot a4", 100)):
            lib.rf_debug_attn_knock(kn)
            t = timeit(f, 10 if S < 10000 else 4)
            torch.cuda.synchronize()
            d = float((out.float() - ref).abs().max())
            line.append(f"{name} {t*1e6:7.1f} ({d:.0e})")
        print(f"S={S} | " + " | ".join(line), flush=True)
lib.rf_debug_attn_knock(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1)
