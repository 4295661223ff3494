"""Library reference point: torch.matmul (hipBLASLt / rocBLAS) on the same box and shapes, to separate
"kernel design" from "what this silicon sustains under its power cap".  Not part of the product path."""
import torch, time
dev = torch.device("cuda:0")
BF = torch.bfloat16

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best

for name, M, N, K in [("8192^3", 8192, 8192, 8192), ("4096^3", 4096, 4096, 4096), ("qkv", 4608, 9216, 3072), ("out", 4608, 3072, 3072),
                      ("ff1", 4608, 12288, 3072), ("ff2", 4608, 3072, 12288), ("sgl_in", 4608, 21504, 3072), ("sgl_out", 4608, 3072, 15360)]:
    x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    b = torch.randn(N, device=dev, dtype=BF)
    s1 = t(lambda: torch.matmul(x, W.t()))
    s2 = t(lambda: torch.nn.functional.linear(x, W, b))
    fl = 2.0 * M * N * K
    print(f"{name:8s} matmul {s1*1e6:8.1f} us {fl/s1/1e12:7.0f} TF   linear+bias {s2*1e6:8.1f} us {fl/s2/1e12:7.0f} TF", flush=True)
# SDPA reference
import torch.nn.functional as F
q = torch.randn(1, 24, 4608, 128, device=dev, dtype=BF); k = torch.randn_like(q); v = torch.randn_like(q)
for be in ("flash", "default"):
    try:
        s = t(lambda: F.scaled_dot_product_attention(q, k, v), iters=10)
        print(f"sdpa S=4608 H=24: {s*1e6:.1f} us {4*24*4608*4608*128/s/1e12:.0f} TF")
        break
    except Exception as e:
        print("sdpa failed", e)
