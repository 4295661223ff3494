"""s_memtime timeline of the GEMM main loop (rf_debug_gemm_timeline): where do a wave's cycles go per K-tile?"""
import ctypes as C, torch
from reflectionflow_amd import _lib, ops
_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
dev = torch.device("cuda:0"); lib = _lib.load(); BF = torch.bfloat16
for name, M, N, K in [("8192^3", 8192, 8192, 8192), ("out 4608x3072x3072", 4608, 3072, 3072), ("sgl_in 4608x21504x3072", 4608, 21504, 3072)]:
    x = torch.randn(M, K, device=dev, dtype=BF); W = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    out = torch.empty(M, N, device=dev, dtype=BF)
    d = ops.build_gemm_desc([ops.Group([ops.Seg(x, W)], out=out)], N)
    tl = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        _lib.check(lib.rf_debug_gemm_timeline(C.byref(d), tl.data_ptr(), ops.stream_ptr()), "timeline")
    torch.cuda.synchronize()
    t = tl.view(16, 8, 8).cpu()
    nk = int(t[0, 0, 4])
    print(f"== {name}: nk={nk}; per K-tile cycles (s_memtime ticks), blocks 0..15 x 8 waves: mean [min..max]")
    for i, lab in enumerate(["drain(vmcnt0)", "barrier", "body", "total"]):
        v = t[:, :, i].float() / nk
        print(f"   {lab:14s} {v.mean():8.0f}  [{v.min():.0f} .. {v.max():.0f}]   per wave of block 0: {[int(a) for a in v[0]]}")
