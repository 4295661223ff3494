"""bench.py's isolated-shape table alone (real epilogues, token groups), 3 repetitions: A/B tool for GEMM dispatch changes."""
import sys, json, torch
sys.path.insert(0, ".")
import bench
from reflectionflow_amd import _lib
_lib.load()
dev = torch.device("cuda:0")
for rep in range(3):
    t = bench.isolated_shapes(dev, 512, 4096, 3072, 12288, 24, 19, 38)
    print(json.dumps({k: v["tflops"] for k, v in t.items()}), flush=True)
