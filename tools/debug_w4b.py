import sys, torch
sys.path.insert(0, ".")
from reflectionflow_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in ((256, 256, 128), (256, 256, 192), (256, 256, 256), (256, 256, 512), (512, 512, 512)):
    x = (torch.randn(M, K, generator=g, device=dev)).to(BF); W = (torch.randn(N, K, generator=g, device=dev) * 0.05).to(BF)
    outs = {}
    for nm, sc in (("w4", L.RF_SCHED_W4), ("w4b", L.RF_SCHED_W4B)):
        with ops.gemm_schedule(sc):
            outs[nm] = ops.linear(x, W, None, splitk_ws=False)
    torch.cuda.synchronize()
    d = (outs["w4"].float() - outs["w4b"].float())
    # per K-tile contribution check: which K tiles are wrong?  compare against partial sums
    ref_tiles = [x[:, k:k + 64].float() @ W[:, k:k + 64].float().t() for k in range(0, K, 64)]
    full = sum(ref_tiles)
    err_b = outs["w4b"].float() - full
    # least squares: express err as combination of tile contributions
    A = torch.stack([t.flatten() for t in ref_tiles], 1)
    coef = torch.linalg.lstsq(A, err_b.flatten().unsqueeze(1)).solution.flatten()
    rows = (d.abs().amax(1) > 0).nonzero().flatten()
    cols = (d.abs().amax(0) > 0).nonzero().flatten()
    t = ops.time_gemm([ops.Group([ops.Seg(x, W)], out=outs["w4b"])], N, iters=5, schedule=L.RF_SCHED_W4B)
    print(f"{M}x{N}x{K}: max|w4-w4b| {float(d.abs().max()):.3f}; rows wrong {rows.numel()} cols wrong {cols.numel()}; err ~ sum coef*tile: {[round(float(c),2) for c in coef]}  time {t*1e6:.1f} us")
