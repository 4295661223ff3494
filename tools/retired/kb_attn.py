"""Attention kernels A/B on one GPU: v1 (4 waves), v2 (8 waves, online softmax), v4 (8 waves, bounded score, 32x32x16
MFMAs), v5 (the same on 16x16x32) at the BASELINE sequence lengths, realistic score scale (q, k ~ RMS-normalised rows,
prescaled q), interleaved repetitions; shader clock from the kernels' own probe."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); lib = _lib.load()
lib.rf_debug_attn_mix(0)   # the plain one-size grids (the mixed-size launch is tools/kb_attn_mix.py's subject)
for S in (4608, 5632, 17920):
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_(); vt.normal_()
    q.mul_(ops.QK_PRESCALE)
    bound = float(q.float().norm(dim=-1).max() * k.float().norm(dim=-1).max()) * 1.01
    out = torch.empty(S, H * 128, device=dev, dtype=torch.bfloat16)
    for rep in range(3):
        line, outs = [], []
        for name, v2, v4, v5, sk, v6 in (("v1", 0, 0, 0, 0, 0), ("v2", 1, 0, 0, 0, 0), ("v4", 1, 1, 0, 0, 0), ("v5", 1, 1, 1, 0, 0), ("v5 split", 1, 1, 1, 1, 0),
                                        ("v6", 1, 1, 1, 0, 1)):
            lib.rf_debug_attn_v2(v2); lib.rf_debug_attn_v4(v4); lib.rf_debug_attn_v5(v5); lib.rf_debug_attn_sk(sk); lib.rf_debug_attn_v6(v6)
            t = timeit(lambda: ops.attention(q, k, vt, S, out=out, q_prescaled=True, score_bound=bound), 10 if S < 10000 else 4)
            torch.cuda.synchronize()
            outs.append(out.clone())
            mhz, us = C.c_double(0), C.c_double(0)
            lib.rf_debug_clock_probe(1, C.byref(mhz), C.byref(us))
            line.append(f"{name}: {t*1e6:8.1f} us {4.0*S*S*H*128/t/1e12:6.1f} TF" + (f" @{mhz.value:5.0f} MHz" if v4 else ""))
        print(f"S={S} bound={bound:.1f}", " | ".join(line), " max|v4-v2|", float((outs[2].float()-outs[1].float()).abs().max()),
              " max|v5-v4|", float((outs[3].float()-outs[2].float()).abs().max()),
              " max|split-v5|", float((outs[4].float()-outs[3].float()).abs().max()),
              " max|v6-v5|", float((outs[5].float()-outs[3].float()).abs().max()), "path", lib.rf_debug_last_attn_path(), flush=True)
lib.rf_debug_attn_v2(-1); lib.rf_debug_attn_v4(1); lib.rf_debug_attn_v5(-1); lib.rf_debug_attn_sk(-1); lib.rf_debug_attn_v6(0)
