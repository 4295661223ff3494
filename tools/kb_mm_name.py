"""Which hipBLASLt/Tensile kernel does torch.matmul pick for the FLUX shapes?  (name only -- library reference point)"""
import torch
dev = torch.device("cuda:0"); BF = torch.bfloat16
for M, N, K in [(8192, 8192, 8192), (4608, 9216, 3072), (4608, 3072, 12288)]:
    x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF)
    for _ in range(3): torch.matmul(x, W.t())
torch.cuda.synchronize()
