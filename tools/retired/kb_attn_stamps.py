"""Where a key tile's clocks go (experiments library): s_memtime stamps of block 0's eight waves around the barrier and the two
halves of key tile 40 of the bounded-score kernel at S = 4608, with fragment reads one / two groups ahead; plus the timing of
reads three groups ahead."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()
from reflectionflow_amd import _lib as L, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = L.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1)
S, H = 4608, 24
q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
f = lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, kernel=L.RF_ATTN_BOUNDED16)
ref = None
for rep in range(7):
    for name, kn in (("shipped (prio 2, rotated)", 256), ("row sums on the matrix pipe", 2304)):
        lib.rf_debug_attn_knock(kn)
        t = timeit(f, 10)
        torch.cuda.synchronize()
        if ref is None:
            ref = ops.attention(q, k, vt, S, q_prescaled=True, kernel=L.RF_ATTN_ONLINE256).float()
        print(f"{name:44s} {t*1e6:7.1f} us   max|d| vs online-softmax kernel {float((out.float() - ref).abs().max()):.1e}", flush=True)
for name, kn in (("shipped + stamps", 272), ("row sums on the matrix pipe + stamps", 2320)):
    lib.rf_debug_attn_knock(kn)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.rf_debug_attn_stamps(buf)
    print(name, ": per wave, clocks of key tile 40 of block 0: [wait + barrier | first half | second half] (F = PV + exp, G = QK + pack; rotated waves 4-7 run G then F) ; start offset vs wave 0")
    base = buf[0]
    for w in range(8):
        s0, s1, s2, s3 = (buf[w * 8 + i] for i in range(4))
        print(f"  wave {w}: {s1 - s0:6d} | {s2 - s1:6d} | {s3 - s2:6d}   total {s3 - s0:6d}   start {int(s0) - int(base):+d}")
lib.rf_debug_attn_knock(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1)
