"""A/B on the six cfg2 launch shapes (fused epilogues as in the model): one 256x256 tile per block (round-1 schedule)
vs ONE persistent launch walking whole tiles (rf_debug_force_gemm_sk(2)): a tile's stores drain under the next tile's
main loop and there is no block dispatch between rounds.  Interleaved repetitions in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
import bench
from reflectionflow_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for S_img in (4096, 16384):
    rows = []
    for rep in range(2):
        for mode in (0, 2):
            lib.rf_debug_force_gemm_sk(mode)
            rows.append((mode, bench.isolated_shapes(dev, 512, S_img, 3072, 12288, 24, 19, 38)))
    lib.rf_debug_force_gemm_sk(-1)
    print(f"==== S = 512 + {S_img} ====")
    for name in rows[0][1]:
        line = [f"{name:8s}"]
        for mode, r in rows:
            line.append(f"{'tile/block' if mode == 0 else 'persistent'} {r[name]['us']:8.1f} us {r[name]['tflops']:7.1f} TF")
        print(" | ".join(line), flush=True)
