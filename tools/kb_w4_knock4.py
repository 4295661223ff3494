import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)
for M, N, K in [(8192, 8192, 8192), (4608, 21504, 3072), (4608, 3072, 12288)]:
    x, W = r(M, K), r(N, K, sc=0.02)
    out = torch.empty(M, N, device=dev, dtype=BF)
    g = [ops.Group([ops.Seg(x, W)], out=out)]
    fl = 2.0 * M * N * K
    for name, sched in [("8-wave tile256", L.RF_SCHED_TILE256), ("w4", 6), ("w4 no-DMA", 7), ("w4 no-sync", 8), ("w4 no-DMA no-sync", 9), ("w4 no-reads", 10), ("w4 MFMA only", 13)]:
        ts = []
        for rep in range(3):
            sec = ops.time_gemm(g, N, iters=30, schedule=sched, splitk_ws=False)
            ts.append(sec)
        sec = min(ts)
        import bench as B
        mhz, loop_us = B.clock_probe(0)
        print(f"{M}x{N}x{K} {name:22s} {sec*1e6:8.1f} us {fl/sec/1e12:7.1f} TF  clk {mhz and round(mhz)}  block-0 main loop {loop_us and round(loop_us, 1)} us; tiles/CU {-(-(-(-M // 256) * -(-N // 256)) // 256)} -> {sec * 1e6 / -(-(-(-M // 256) * -(-N // 256)) // 256):.1f} us per tile round", flush=True)
