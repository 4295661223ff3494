"""Run the gfx950 issue-cost micro-benchmarks (tools/ubench/ub.hip).  Prints cycles per operation per wave."""
import ctypes as C, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libub.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "ub.hip"), "-o", so])
lib = C.CDLL(so)
lib.ub_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
src = torch.randn(8 << 20, device=dev).to(torch.bfloat16)      # 16 MiB source for the DMA tests (L2 / MALL resident)
out = torch.zeros(16, dtype=torch.int64, device=dev); sink = torch.zeros(512, device=dev)
REP = 64
def run(test, nw, blocks=1):
    out.zero_()
    for _ in range(2):
        lib.ub_run(test, nw, blocks, src.data_ptr(), out.data_ptr(), sink.data_ptr(), None)
    torch.cuda.synchronize()
    return out.view(8, 2).cpu().tolist()
def show(name, rows, per, nw=8):
    vals = [f"{r[0] / per:.1f}" + (f"(+{r[1]})" if r[1] else "") for r in rows[:nw]]
    print(f"{name:58s} " + " ".join(vals), flush=True)
for blocks in (1, 256):
    print(f"---- {blocks} workgroup(s) ----")
    show("MFMA 32x32x16, 4 waves (1/SIMD): cycles per MFMA", run(0, 4, blocks), REP * 8, 4)
    show("MFMA 32x32x16, 8 waves (2/SIMD): cycles per MFMA per wave", run(0, 8, blocks), REP * 8)
    show("s_barrier, 8 waves: cycles per barrier", run(1, 8, blocks), REP * 8)
    show("buffer_load lds 1 KiB, 1 wave issuing: cycles per issue (+drain)", run(2, 1, blocks), REP * 8, 1)
    show("buffer_load lds 1 KiB, 8 waves issuing: cycles per issue (+drain)", run(2, 8, blocks), REP * 8)
    show("8 MFMA + 2 DMA interleaved, 4 waves: cycles per MFMA", run(3, 4, blocks), REP * 8, 4)
    show("8 MFMA + 2 DMA interleaved, 8 waves: cycles per MFMA per wave", run(3, 8, blocks), REP * 8)
    show("8 x ds_read_b128 + wait, 4 waves: cycles per group of 8", run(4, 4, blocks), REP, 4)
    show("8 x ds_read_b128 + wait, 8 waves: cycles per group of 8", run(4, 8, blocks), REP)
    show("ping-pong skeleton (8 MFMA, 2 barriers): cycles per phase", run(5, 8, blocks), REP)

print("---- LDS port sharing (1 workgroup; 8 x 1 KiB per iteration per wave; cycles per KiB per wave) ----")
for blocks in (1, 256):
    for name, t in (("ds_read_b128 x4 waves alone", 6), ("LDS-DMA x4 waves alone", 7), ("ds_read x4 + LDS-DMA x4 together", 8),
                    ("ds_write_b128 x4 waves alone", 9), ("ds_read x4 + ds_write x4 together", 10)):
        show(f"[{blocks} wg] {name}", run(t, 8, blocks), REP * 4 * 8)
