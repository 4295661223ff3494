"""A/B: plain vs non-temporal output stores in the GEMM epilogue (rf_debug_gemm_nt_store) on the cfg2 launch shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
import bench
from reflectionflow_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
rows = []
for rep in range(3):
    for nt in (0, 1):
        lib.rf_debug_gemm_nt_store(nt)
        rows.append((nt, bench.isolated_shapes(dev, 512, 4096, 3072, 12288, 24, 19, 38)))
lib.rf_debug_gemm_nt_store(0)
for name in rows[0][1]:
    print(" | ".join([f"{name:8s}"] + [f"{'nt' if nt else 'plain'} {r[name]['us']:7.1f} us {r[name]['tflops']:6.1f}" for nt, r in rows]), flush=True)
