"""Knock-outs of the bounded-score attention kernel v5 (timing only, results wrong): which of fragment reads and softmax
VALU work costs what on top of 64 MFMAs per tile and wave (1024 cycles; two waves per SIMD -> 2048 per tile time)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = _lib.load()
lib.rf_debug_attn_v2(1); lib.rf_debug_attn_sk(0); lib.rf_debug_attn_v5(1); lib.rf_debug_attn_mix(0)
for S in (4608, 17920):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_(); q.mul_(ops.QK_PRESCALE)
    bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    for rep in range(2):
        for kn, name in ((0, "full"), (1, "no fragment reads"), (2, "no exp2 / sums / packing"), (3, "MFMA + DMA + barriers")):
            lib.rf_debug_attn_knock(kn)
            t = timeit(lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound, scratch=False), 10 if S < 10000 else 4)
            torch.cuda.synchronize()
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(1, C.byref(mhz), C.byref(us))
            print(f"S={S} {name:28s} {t*1e6:8.1f} us {4.0*S*S*H*128/t/1e12:7.1f} TF @{mhz.value:5.0f} MHz  block 0 loop {us.value:7.1f} us = {us.value*mhz.value/(S//64):6.0f} clocks per key tile", flush=True)
lib.rf_debug_attn_knock(0); lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v5(-1); lib.rf_debug_attn_mix(-1)
