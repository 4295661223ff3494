"""How many 192-query workgroups per head should the mixed-size attention launch use?  Product library, bounded kernel, S = 4608 and
5632, 24 heads: mix_small = 0 (the library's plan) and explicit values, interleaved repetitions."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as L, ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


dev = torch.device("cuda:0"); lib = L.load()
for S, cands in ((4608, (0, 4, 8, 12, 16, 20, 24)), (5632, (0, 8, 16, 20, 24, 28))):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    for rep in range(5):
        line = []
        t = timeit(lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=30.0, kernel=L.RF_ATTN_BOUNDED16), 10)
        line.append(f"plain {t*1e6:6.1f}")
        for b in cands:
            t = timeit(lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=30.0, kernel=L.RF_ATTN_BOUNDED16_MIX, mix_small=b), 10)
            line.append(f"b={b:2d} {t*1e6:6.1f}")
        print(f"S={S} | " + " | ".join(line), flush=True)
