import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import bench_gemm
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_STORE
tiles = tuple(int(x) for x in os.environ.get("RF_TILES", "256,257").split(","))
for name, M, N, K, epi in [("qkv", 4608, 9216, 3072, RF_EPI_STORE), ("out", 4608, 3072, 3072, RF_EPI_GATE_RES), ("ff_up", 4608, 12288, 3072, RF_EPI_GELU),
                           ("ff_dn", 4608, 3072, 12288, RF_EPI_GATE_RES), ("sgl_in", 4608, 21504, 3072, RF_EPI_STORE), ("sgl_out", 4608, 3072, 15360, RF_EPI_GATE_RES),
                           ("sq8192", 8192, 8192, 8192, RF_EPI_STORE)]:
    for rep in range(2):
        print(name, " ".join(f"{t}:{bench_gemm(M, N, K, t, epi, iters=8)[1]:7.1f}" for t in tiles), flush=True)
