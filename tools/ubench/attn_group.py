"""Clocks per attention 'group' (4 x 16x16x32 MFMA + the VALU / LDS work beside them) with one and two waves per SIMD
(tools/ubench/attn_group.hip).  4 MFMAs = 64 clocks of the matrix pipe."""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libattngroup.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "attn_group.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "attn_group.hip"), "-o", so])
lib = C.CDLL(so)
lib.attn_group_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(256 * 512, device=dev); clk = torch.zeros(2, dtype=torch.int64, device=dev)
ITERS = 2000
names = {0: "4 MFMA", 1: "+ 4 exp2", 3: "+ 4 exp2 + 4 adds (= F without reads)", 4: "+ 2 cvt_pk (= G without reads)", 8: "+ 2 ds_read_b128",
         9: "+ 2 reads + 4 exp2", 11: "+ 2 reads + 4 exp2 + 4 adds (= F)", 12: "+ 2 reads + 2 cvt_pk (= G)",
         17: "NO MFMA: 4 exp2 alone", 19: "NO MFMA: 4 exp2 + 4 adds", 20: "NO MFMA: 2 cvt_pk"}
for blocks in (1,):
    for threads in (256, 512):
        for var in (0, 1, 3, 4, 8, 9, 11, 12, 17, 19, 20):
            for _ in range(2):
                assert lib.attn_group_run(var, threads, blocks, ITERS, sink.data_ptr(), clk.data_ptr(), None) == 0
            torch.cuda.synchronize()
            c = clk.cpu().tolist()[0]
            per_group = c / (ITERS * 8)
            print(f"blocks {blocks:3d}  waves/SIMD {threads // 256}  {names[var]:44s} {per_group:7.1f} clocks per group per wave  ({per_group / (threads // 256):6.1f} per SIMD-group)", flush=True)
