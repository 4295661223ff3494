#!/usr/bin/env python3
"""Knock-out study of the attention forward's key-tile loop ("no softmax VALU, no DMA, no V-path ... and the measured
ceiling each implies").  The product sources stay clean: every variant is a TEXTUAL patch of a temporary copy of csrc/ (asserted to
apply), built into its own librf_flux.so and timed through the same C entry point (rf_attention, bounded-score mixed-size launch = what
the bench's forward runs at S = 4608, 24 heads).  Knocked-out variants compute garbage on purpose -- only their time means something.

    python tools/kb_attn_knock.py [--S 4608] [--heads 24] [--iters 50] [--reps 3] [--variants full,no_exp,...]
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


def sub(s, old, new, count=1):
    assert s.count(old) >= count, f"patch anchor not found ({count}x): {old[:70]!r}"
    return s.replace(old, new)


def p_exp(s):   # no transcendental: P = S (the v_mov / register traffic stays)
    s = sub(s, "e4[j] = __builtin_amdgcn_exp2f(s_cur[G][j]);", "e4[j] = s_cur[G][j];")
    return sub(s, "float e0 = __builtin_amdgcn_exp2f(s_cur[G >> 1][(G & 1) * 2]), e1 = __builtin_amdgcn_exp2f(s_cur[G >> 1][(G & 1) * 2 + 1]);",
               "float e0 = s_cur[G >> 1][(G & 1) * 2], e1 = s_cur[G >> 1][(G & 1) * 2 + 1];")


def p_pack(s):  # no fp32 -> bf16 conversion of P (the operand words are the fp32 bit patterns)
    s = sub(s, "uint32_t w0 = pack2(s_cur[ti][0], s_cur[ti][1]);", "uint32_t w0 = __builtin_bit_cast(uint32_t, s_cur[ti][0]);")
    return sub(s, "uint32_t w1 = pack2(s_cur[ti][2], s_cur[ti][3]);", "uint32_t w1 = __builtin_bit_cast(uint32_t, s_cur[ti][2]);")


def p_dma(s):   # no LDS-DMA inside the loop (the prologue's tiles stay in the ring)
    s = sub(s, "    if (t + 3 < nt) issue_k(t + 3, (TS + 3) % 4);\n", "")
    return sub(s, "    if (t + 1 < nt) issue_v(t + 1, (TS + 1) % 4);\n", "")


def p_reads(s):  # no fragment reads inside the loop (the MFMAs multiply whatever the first two reads left in the registers)
    return sub(s, "      if constexpr (P >= 16) {\n      } else if constexpr (G < 8) {", "      if constexpr (P >= 2) {\n      } else if constexpr (G < 8) {")


def p_pv(s):    # no P V MFMAs (and no row-sum MFMAs)
    s = sub(s, "          oacc[G][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[P % NFR][b], pf[b * NQT + qt], oacc[G][qt], 0, 0, 0);",
            "          asm volatile(\"\" :: \"v\"(fr[P % NFR][b]), \"v\"(pf[b * NQT + qt]));")
    return sub(s, "        for (int qt = 0; qt < NQT; ++qt) lsum[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[(G / 4) * NQT + qt], lsum[qt], 0, 0, 0);",
               "        for (int qt = 0; qt < NQT; ++qt) asm volatile(\"\" :: \"v\"(pf[(G / 4) * NQT + qt]));")


def p_qk(s):    # no Q K^T MFMAs
    return sub(s, "          s_nxt[bt * NQT + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[P % NFR][e], qf[qt][dsb + e], s_nxt[bt * NQT + qt], 0, 0, 0);\n        }",
               "          asm volatile(\"\" : \"+v\"(s_nxt[bt * NQT + qt]) : \"v\"(fr[P % NFR][e]), \"v\"(qf[qt][dsb + e]));\n        }")


def p_bar(s):   # no workgroup barriers in the loop (the counted vmcnt waits stay)
    s = sub(s, "    else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");                        \\\n    __builtin_amdgcn_s_barrier();                                                \\\n",
            "    else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");                        \\\n")
    s = sub(s, "      else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n      __builtin_amdgcn_s_barrier();\n    }\n  };",
            "      else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    }\n  };")
    return sub(s, "    if constexpr (NPC == 0) {\n      __builtin_amdgcn_s_barrier();\n    } else if constexpr (NPC == 2) {", "    if constexpr (NPC == 0) {\n    } else if constexpr (NPC == 2) {")


def p_prio(s):  # no s_setprio scheme
    return sub(s, "constexpr int ATT5_VAR = 256 | 2048;", "constexpr int ATT5_VAR = 2048;")


def p_summ(s):  # row sums back on the VALU (what the kernel did before round 3's SUMM)
    return sub(s, "constexpr int ATT5_VAR = 256 | 2048;", "constexpr int ATT5_VAR = 256;")


def p_var(bits):   # another compile-time schedule option set of attn5_body (every one computes the same result)
    return lambda s: sub(s, "constexpr int ATT5_VAR = 256 | 2048;", f"constexpr int ATT5_VAR = {bits};")


def p_small_cost(c):   # the mixed-size launch's cost model: a 192-query workgroup's time relative to a 256-query one's
    return lambda s: sub(s, "constexpr float ATT5_SMALL_COST = 0.9f;", f"constexpr float ATT5_SMALL_COST = {c}f;")


TUNING = {
    "full_again": [],
    "prio_scheme1": [p_var("128 | 2048")], "prio_scheme3": [p_var("384 | 2048")], "prio_scheme6": [p_var("768 | 2048")],
    "reads_1_group_ahead": [p_var("256 | 2048 | 4")], "reads_3_groups_ahead": [p_var("256 | 2048 | 32")],
    "small_cost_0.80": [p_small_cost("0.80")], "small_cost_0.85": [p_small_cost("0.85")], "small_cost_0.95": [p_small_cost("0.95")],
}
VARIANTS = {
    "full": [],
    "no_exp": [p_exp],
    "no_pack": [p_pack],
    "no_softmax_valu": [p_exp, p_pack],
    "no_dma": [p_dma],
    "no_frag_reads": [p_reads],
    "no_lds_traffic": [p_dma, p_reads],
    "no_barriers": [p_bar],
    "no_pv": [p_pv],
    "no_qk": [p_qk],
    "no_setprio": [p_prio],
    "rowsum_on_valu": [p_summ],
    "mfma_only": [p_exp, p_pack, p_dma, p_reads, p_bar],
}
KNOCKOUTS = list(VARIANTS)
VARIANTS.update(TUNING)


def build(name, patches, work):
    d = os.path.join(work, name)
    shutil.copytree(os.path.join(ROOT, "reflectionflow_amd", "csrc"), os.path.join(d, "reflectionflow_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    src = os.path.join(d, "reflectionflow_amd", "csrc", "attention.hip")
    s = open(src).read()
    for p in patches:
        s = p(s)
    open(src, "w").write(s)
    r = subprocess.run(["make", "-C", os.path.dirname(src), "-j8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    so = os.path.join(d, "reflectionflow_amd", "librf_flux.so")
    assert os.path.exists(so), so
    return so


def load(so):
    lib = C.CDLL(so)
    lib.rf_attention.restype, lib.rf_attention.argtypes = C.c_int, [C.POINTER(L.rf_attn_desc), C.c_void_p]
    lib.rf_debug_last_attn_path.restype = C.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=4608)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default=",".join(KNOCKOUTS), help="default: the knock-outs; `--variants full," + ",".join(TUNING) + "` = the tuning sweep")
    ap.add_argument("--kernel", type=int, default=L.RF_ATTN_AUTO if hasattr(L, "RF_ATTN_AUTO") else 0)
    args = ap.parse_args()
    L.load()
    dev = torch.device("cuda:0")
    H, S = args.heads, args.S
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q.copy_((torch.randn(q.shape, generator=g, device=dev) * 0.3).to(torch.bfloat16))
    k.copy_((torch.randn(k.shape, generator=g, device=dev) * 0.3).to(torch.bfloat16))
    vt.copy_(torch.randn(vt.shape, generator=g, device=dev).to(torch.bfloat16))
    out = torch.empty(S, H * 128, dtype=torch.bfloat16, device=dev)
    ws = ops.attn_scratch(dev)
    d = L.rf_attn_desc()
    d.q, d.k, d.vt, d.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.heads, d.S, d.s_pad, d.n_main, d.ldo = H, S, s_pad, S, out.stride(0)
    d.mode, d.q_prescaled, d.cross_bias, d.scale = 0, 1, 0.0, 1.0
    d.score_bound, d.lag_thresh, d.kernel, d.mix_small = 20.0, 0.0, args.kernel, 0
    d.ws, d.ws_bytes, d.lse = ws.data_ptr(), ws.numel() * 4, None
    names = [n for n in args.variants.split(",") if n]
    work = tempfile.mkdtemp(prefix="rf_attn_knock_")
    libs = {}
    try:
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max(1, min(6, (os.cpu_count() or 8) // 8))) as ex:
            sos = dict(zip(names, ex.map(lambda n: build(n, VARIANTS[n], work), names)))
        for n in names:
            libs[n] = load(sos[n])
        print(f"built {len(names)} variants", flush=True)
        flops = 4.0 * S * S * 128 * H
        res = {n: [] for n in names}
        st = ops.stream_ptr()
        for rep in range(args.reps):
            for n in names:
                lib = libs[n]
                for _ in range(5):
                    assert lib.rf_attention(C.byref(d), st) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    lib.rf_attention(C.byref(d), st)
                e1.record()
                torch.cuda.synchronize()
                res[n].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        path = libs[names[0]].rf_debug_last_attn_path()
        base = sorted(res["full"])[len(res["full"]) // 2] if "full" in res else None
        print(f"S = {S}, heads = {H}, launch path {path} (10 = bounded mixed-size), {args.iters} launches x {args.reps} interleaved repetitions; us per launch (median), TF = 4 S^2 128 heads / t")
        table = {}
        for n in names:
            t = sorted(res[n])[len(res[n]) // 2]
            table[n] = {"us": round(t, 1), "us_all": [round(v, 1) for v in res[n]], "tflops_equiv": round(flops / t / 1e6, 1),
                        "vs_full": round(t / base, 3) if base else None}
            print(f"  {n:18s} {t:7.1f} us  {flops / t / 1e6:7.1f} TF-equiv  x{t / base if base else 0:.3f}   {[round(v, 1) for v in res[n]]}", flush=True)
        print(json.dumps(table))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
