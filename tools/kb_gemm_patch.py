#!/usr/bin/env python3
"""Micro-variants of the shipped 8-wave GEMM loop (`gemm_mainloop_pp3_m16`) as TEXTUAL patches of a temporary copy of csrc/ (the
product sources stay clean), each built into its own librf_flux.so and timed on the six cfg2 shapes, sustained (20 back-to-back
launches), interleaved over repetitions, with a bit-equality check against the unpatched loop.

    python tools/kb_gemm_patch.py [--iters 20] [--reps 3] [--variants base,prio_mfma,...]
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reflectionflow_amd import _lib as L, ops   # noqa: E402


def sub(s, old, new, count=None):
    n = s.count(old)
    assert n >= 1 and (count is None or n == count), f"patch anchor found {n}x (want {count}): {old[:80]!r}"
    return s.replace(old, new)


def body(s):
    i = s.index("__device__ __forceinline__ void gemm_mainloop_pp3_m16(")
    j = s.index("// launches on the 16x16 MFMA shapes (gemm_mainloop_pp2_m16)", i)
    return i, j


def in_pp3(fn):
    def wrap(s):
        i, j = body(s)
        return s[:i] + fn(s[i:j]) + s[j:]
    return wrap


@in_pp3
def p_prio_mfma(b):     # s_setprio 1 while a wave multiplies, 0 while it reads / stages
    for rb, cb, P, e in (("0", "0", "P", "e0"), ("2", "0", "Q", "e0"), ("2", "4", "Q", "e1"), ("0", "4", "P", "e1")):
        old = f"    mma16s({rb}, {cb}, {P}, {e}, m);\n"
        b = sub(b, old, "    __builtin_amdgcn_s_setprio(1);\n" + old + "    __builtin_amdgcn_s_setprio(0);\n", 1)
    return b


@in_pp3
def p_prio_load(b):     # the opposite: priority to the wave that reads / stages
    for rb, cb, P, e in (("0", "0", "P", "e0"), ("2", "0", "Q", "e0"), ("2", "4", "Q", "e1"), ("0", "4", "P", "e1")):
        old = f"    mma16s({rb}, {cb}, {P}, {e}, m);\n"
        b = sub(b, old, "    __builtin_amdgcn_s_setprio(0);\n" + old + "    __builtin_amdgcn_s_setprio(1);\n", 1)
    return b


def _static(g, prio=1):
    @in_pp3
    def f(b):
        b = sub(b, "  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs half a phase behind group 0\n",
                f"  if (grp == {g}) __builtin_amdgcn_s_setprio({prio});\n  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs half a phase behind group 0\n", 1)
        return sub(b, "  if (grp == 0) __builtin_amdgcn_s_barrier();  // match group 1's extra barrier\n",
                   "  __builtin_amdgcn_s_setprio(0);\n  if (grp == 0) __builtin_amdgcn_s_barrier();  // match group 1's extra barrier\n", 1)
    return f


@in_pp3
def p_mfma_rt_outer(b):  # MFMA order inside a phase: row tile outermost (consecutive MFMAs share the A fragment instead of the W fragment)
    return sub(b, "    for (int ks = 0; ks < 2; ++ks)\n#pragma unroll\n      for (int ct = 0; ct < 4; ++ct)\n#pragma unroll\n        for (int rt = 0; rt < 2; ++rt)\n",
               "    for (int ks = 0; ks < 2; ++ks)\n#pragma unroll\n      for (int rt = 0; rt < 2; ++rt)\n#pragma unroll\n        for (int ct = 0; ct < 4; ++ct)\n", 1)


def p_gw(n):   # raster: width of the column bands a XCD's chunk of tiles walks (tile_coords)
    return lambda s: sub(s, "  constexpr int GW = 8;\n", f"  constexpr int GW = {n};\n", 1)


VARIANTS = {"gw4": [p_gw(4)], "gw6": [p_gw(6)], "gw12": [p_gw(12)], "gw16": [p_gw(16)], "gw2": [p_gw(2)],
            "base": [], "prio_mfma": [p_prio_mfma], "prio_load": [p_prio_load], "prio_static_group1": [_static(1)], "prio_static_group0": [_static(0)],
            "prio_static_group0_p2": [_static(0, 2)], "prio_static_group0_p3": [_static(0, 3)], "base_again": [],
            "mfma_rt_outer": [p_mfma_rt_outer]}
SHAPES = [("dbl_qkv", 4608, 9216, 3072), ("dbl_out", 4608, 3072, 3072), ("dbl_ff1", 4608, 12288, 3072), ("dbl_ff2", 4608, 3072, 12288),
          ("sgl_in", 4608, 21504, 3072), ("sgl_out", 4608, 3072, 15360)]


def build(name, patches, work):
    d = os.path.join(work, name)
    shutil.copytree(os.path.join(ROOT, "reflectionflow_amd", "csrc"), os.path.join(d, "reflectionflow_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    src = os.path.join(d, "reflectionflow_amd", "csrc", "gemm_bf16.hip")
    s = open(src).read()
    for p in patches:
        s = p(s)
    open(src, "w").write(s)
    r = subprocess.run(["make", "-C", os.path.dirname(src), "-j8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return os.path.join(d, "reflectionflow_amd", "librf_flux.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    names = [n for n in args.variants.split(",") if n]
    work = tempfile.mkdtemp(prefix="rf_gemm_patch_")
    try:
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max(1, min(6, (os.cpu_count() or 8) // 8))) as ex:
            sos = dict(zip(names, ex.map(lambda n: build(n, VARIANTS[n], work), names)))
        print(f"built {len(sos)} variants", flush=True)
        if args.build_only:
            return
        L.load()
        libs = {}
        for n, so in sos.items():
            lib = C.CDLL(so)
            lib.rf_gemm_bf16.restype, lib.rf_gemm_bf16.argtypes = C.c_int, [C.POINTER(L.rf_gemm_desc), C.c_void_p]
            libs[n] = lib
        dev = torch.device("cuda:0")
        BF = torch.bfloat16
        g = torch.Generator(device=dev).manual_seed(0)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa: E731
        st = ops.stream_ptr()
        table = {}
        for sname, M, N, K in SHAPES:
            x, W, b = r(M, K), r(N, K, sc=0.02), r(N)
            out = torch.empty(M, N, dtype=BF, device=dev)
            d = ops.build_gemm_desc([ops.Group([ops.Seg(x, W)], bias=b, out=out)], N, L.RF_EPI_STORE, schedule=L.RF_SCHED_TILE256,
                                    splitk_ws=ops.splitk_scratch(dev))
            fl = 2.0 * M * N * K
            row, ref = {n: [] for n in names}, None
            for rep in range(args.reps):
                for n in names:
                    lib = libs[n]
                    for _ in range(3):
                        assert lib.rf_gemm_bf16(C.byref(d), st) == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        lib.rf_gemm_bf16(C.byref(d), st)
                    e1.record()
                    torch.cuda.synchronize()
                    row[n].append(fl / (e0.elapsed_time(e1) / args.iters * 1e-3) / 1e12)
                    if rep == 0:
                        o = out.clone()
                        if ref is None:
                            ref = o
                        else:
                            assert torch.equal(o, ref), f"{n} is not bit-equal to {names[0]} on {sname}"
            table[sname] = {n: round(sorted(v)[len(v) // 2], 1) for n, v in row.items()}
            print(sname, json.dumps(table[sname]), flush=True)
        print("geomean vs base:", {n: round(float(torch.tensor([table[s][n] / table[s][names[0]] for s in table]).log().mean().exp()), 4) for n in names})
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
