"""Warm vs HBM-cold launches of the 256 x 256 GEMM loops: between two timed launches 2 GiB of unrelated data goes through the memory
system (what a 57-block forward does to a launch's weight panels).  python tools/kb_cold.py"""
import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa
flush_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
flush_dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
SCHED = {"t256": L.RF_SCHED_TILE256, "w4b": L.RF_SCHED_W4B}
for name, M, N, K in (("sgl_out", 4608, 3072, 15360), ("dbl_ff2", 4608, 3072, 12288), ("sgl_in", 4608, 21504, 3072), ("dbl_ff1", 4608, 12288, 3072)):
    x, W, b = r(M, K), r(N, K, sc=0.02), r(N)
    out = torch.empty(M, N, dtype=BF, device=dev)
    grp = [ops.Group([ops.Seg(x, W)], bias=b, out=out)]
    fl = 2.0 * M * N * K
    line = [name]
    for sname, sc in SCHED.items():
        d = ops.build_gemm_desc(grp, N, L.RF_EPI_STORE, schedule=sc, splitk_ws=ops.splitk_scratch(dev))
        import ctypes as C
        lib = L.load()
        for mode in ("warm", "cold_weights", "cold_all"):
            ts = []
            for rep in range(6):
                if mode != "warm":
                    flush_dst.copy_(flush_src)                          # 2 GiB of traffic: evicts L2 and the Infinity Cache
                    if mode == "cold_weights":
                        x.add_(0)                                       # activations were just written by the previous kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.rf_gemm_bf16(C.byref(d), ops.stream_ptr()), "gemm")
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3)
            t = sorted(ts)[len(ts) // 2]
            line.append(f"{sname} {mode} {fl / t / 1e12:7.1f}")
    print(" | ".join(line), flush=True)
