#!/usr/bin/env python3
"""Kernel micro-benchmarks on the FLUX.1-dev shapes (run on the GPU box).
Random data (never zeros: DVFS inflates zero-filled numbers, MI355X_MICROARCH.md)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflectionflow_amd import _lib as _rf_lib; _rf_lib.load_experiments()   # A/B switches live in librf_flux_exp.so (make -C reflectionflow_amd/csrc EXPERIMENTS=1)
from reflectionflow_amd import _lib, ops  # noqa: E402
from reflectionflow_amd.ops import RF_EPI_GATE_RES, RF_EPI_GELU, RF_EPI_QKV_GELU, RF_EPI_STORE, Group, Seg  # noqa: E402

dev = torch.device("cuda:0")
TILES = tuple(int(x) for x in os.environ.get("RF_TILES", "128,256").split(","))
ONLY_GEMM = os.environ.get("RF_ONLY_GEMM", "0") == "1"
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm(M, N, K, tile, epi=RF_EPI_STORE, iters=10):
    lib = _lib.load()
    lib.rf_debug_force_gemm_tile(tile)
    x = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    b = torch.randn(N, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    kw = {}
    if epi == RF_EPI_GATE_RES:
        kw = dict(residual=out, gate=b)
    t = timeit(lambda: ops.linear(x, W, b, epilogue=epi, out=out, **kw), iters)
    lib.rf_debug_force_gemm_tile(0)
    return t, 2.0 * M * N * K / t / 1e12


def main():
    res = {}
    S = 4608
    shapes = [("qkv_dbl", S, 9216, 3072, RF_EPI_STORE), ("out_proj", S, 3072, 3072, RF_EPI_GATE_RES),
              ("ff_up", S, 12288, 3072, RF_EPI_GELU), ("ff_down", S, 3072, 12288, RF_EPI_GATE_RES),
              ("sgl_in", S, 21504, 3072, RF_EPI_STORE), ("sgl_out", S, 3072, 15360, RF_EPI_GATE_RES),
              ("sq4096", 4096, 4096, 4096, RF_EPI_STORE), ("sq8192", 8192, 8192, 8192, RF_EPI_STORE)]
    for name, M, N, K, epi in shapes:
        for tile in TILES:
            t, tf = bench_gemm(M, N, K, tile, epi)
            res[f"gemm_{name}_t{tile}"] = dict(ms=t * 1e3, tflops=tf)
            print(f"gemm {name:9s} {M}x{N}x{K} tile{tile}: {t*1e3:8.3f} ms  {tf:7.1f} TF/s", flush=True)
    if ONLY_GEMM:
        return
    # attention, FLUX joint sequence
    for S in (4608, 5632):
        H = 24
        q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
        q.normal_(); k.normal_(); vt.normal_()
        out = torch.empty(S, H * 128, device=dev, dtype=BF)
        t = timeit(lambda: ops.attention(q, k, vt, S, out=out), 10)
        tf = 4.0 * S * S * H * 128 / t / 1e12
        res[f"attn_S{S}"] = dict(ms=t * 1e3, tflops=tf)
        print(f"attention S={S} H={H}: {t*1e3:8.3f} ms  {tf:7.1f} TF/s", flush=True)
    # bandwidth kernels
    S, D = 4608, 3072
    x = torch.randn(S, D, device=dev).to(BF)
    sc, sh = torch.randn(D, device=dev).to(BF), torch.randn(D, device=dev).to(BF)
    o = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm_modulate(x, sc, sh, out=o), 20)
    res["ln_mod"] = dict(us=t * 1e6, gbps=2 * S * D * 2 / t / 1e9)
    print(f"ln_mod {S}x{D}: {t*1e6:8.1f} us  {2*S*D*2/t/1e9:7.0f} GB/s", flush=True)
    H = 24
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.normal_(); k.normal_()
    w = torch.ones(128, device=dev, dtype=BF)
    cos = torch.rand(S, 128, device=dev); sin = torch.rand(S, 128, device=dev)
    t = timeit(lambda: ops.qk_rmsnorm_rope(q, k, S, 512, w, w, w, w, cos, sin), 20)
    by = 2 * 2 * H * S * 128 * 2
    res["qk_norm_rope"] = dict(us=t * 1e6, gbps=by / t / 1e9)
    print(f"qk_norm_rope H={H} S={S}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kbench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
