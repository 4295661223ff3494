"""A/B on the cfg2 launch shapes: 8/4/8/4 (gemm_mainloop_pp2_m16) vs evenly loaded 6/6/6/6 phases (gemm_mainloop_pp3_m16),
both on 16x16x32 MFMAs, one tile per block; bit-equality of the two."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
import bench
from reflectionflow_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
x = torch.randn(4608, 3072, device=dev).to(torch.bfloat16); W = (torch.randn(3072, 3072, device=dev) * 0.05).to(torch.bfloat16)
ys = []
for ev in (0, 1):
    lib.rf_debug_gemm_even(ev)
    y = torch.empty(4608, 3072, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Group([ops.Seg(x, W)], out=y)], 3072, splitk_ws=False); torch.cuda.synchronize(); ys.append(y)
print("bit-identical:", bool(torch.equal(ys[0], ys[1])))
rows = []
for rep in range(3):
    for ev in (0, 1):
        lib.rf_debug_gemm_even(ev)
        rows.append((ev, bench.isolated_shapes(dev, 512, 4096, 3072, 12288, 24, 19, 38)))
lib.rf_debug_gemm_even(1)
for name in rows[0][1]:
    print(" | ".join([f"{name:8s}"] + [f"{'6/6/6/6' if ev else '8/4/8/4'} {r[name]['us']:7.1f} us {r[name]['tflops']:6.1f}" for ev, r in rows]), flush=True)
