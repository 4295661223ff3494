"""Sustained MFMA-only rate under the 1.4 kW cap, random bf16 operands: 32x32x16 vs 16x16x32 (tools/ubench/mfma_power.hip)."""
import ctypes as C, os, subprocess, time, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libmfmapower.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_power.hip"), "-o", so])
lib = C.CDLL(so)
lib.mfma_power_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(512, device=dev); clk = torch.zeros(2, dtype=torch.int64, device=dev)
ITERS = 20000
for rep in range(2):
    for kind, name, per_iter, peak in ((0, "bf16 32x32x16", 8 * 2 * 32 * 32 * 16, 2516.6), (1, "bf16 16x16x32", 16 * 2 * 16 * 16 * 32, 2516.6),
                                        (2, "fp8 32x32x64 (MX, unit scales)", 8 * 2 * 32 * 32 * 64, 5033.2), (3, "fp8 16x16x128 (MX, unit scales)", 16 * 2 * 16 * 16 * 128, 5033.2)):
        for _ in range(3):
            lib.mfma_power_run(kind, 256, ITERS, sink.data_ptr(), clk.data_ptr(), None)
        torch.cuda.synchronize(); t0 = time.time(); n = 0
        while time.time() - t0 < 2.5:
            lib.mfma_power_run(kind, 256, ITERS, sink.data_ptr(), clk.data_ptr(), None); n += 1
            if n % 8 == 0: torch.cuda.synchronize()
        torch.cuda.synchronize(); dt = time.time() - t0
        c = clk.cpu().tolist()
        mhz = c[0] / (c[1] / 100.0)
        tf = n * 256 * 8 * ITERS * per_iter / dt / 1e12
        print(f"{name:32s} {tf:7.1f} TF sustained over {dt:.1f} s   shader clock {mhz:5.0f} MHz   -> {tf / (peak * mhz / 2400) * 100:5.1f} % of the rate at that clock", flush=True)
