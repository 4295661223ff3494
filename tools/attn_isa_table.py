#!/usr/bin/env python3
"""Static instruction table of the attention forward's key-tile loop: compiles csrc/attention.hip to gfx950 ISA
(device only), finds every back-edge loop of `attn_fwd_kernel_v5<false>` (the plain launch of the bounded kernel: waves 0-3 and the
rotated waves 4-7 have their own loop; each iteration = 4 key tiles, the ring unroll) and counts instructions by class per KEY TILE and
wave.  CPU only (hipcc cross-compiles).    python tools/attn_isa_table.py [--kernel SUBSTR] [--md OUT]"""
import argparse
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="attn_fwd_kernel_v5ILb0E")
ap.add_argument("--src", default=os.path.join(ROOT, "reflectionflow_amd", "csrc", "attention.hip"))
ap.add_argument("--tiles-per-iteration", type=int, default=4)
ap.add_argument("--md", default=None)
args = ap.parse_args()
csrc = os.path.dirname(args.src)
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "a.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{csrc}", "-Wno-unused-value",
                    "-Wno-comment", "--cuda-device-only", "-S", args.src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(args.kernel) + r"\w*:", l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def klass(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_exp"):
        return "v_exp_f32"
    if op.startswith("v_cvt_pk_bf16") or op.startswith("v_cvt_pk"):
        return "v_cvt_pk (pack P)"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read_b128" if "b128" in op else "ds_read (other)"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ds_write"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "LDS-DMA / vmem load"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return "vmem store"
    if op == "s_barrier":
        return "s_barrier"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_setprio":
        return "s_setprio"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("v_accvgpr") or op.startswith("v_mov") or op.startswith("v_pk_mov"):
        return "v_mov / accvgpr moves"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith("s_"):
        return "SALU (m0, addresses, loop)"
    return "other"


rows = []
for (a, b) in loops:
    ops = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", ".")) and l.split()]
    c = collections.Counter(klass(o) for o in ops)
    if c["MFMA"] < 32:
        continue
    rows.append((a, b, len(ops), c))
order = ["MFMA", "v_exp_f32", "v_cvt_pk (pack P)", "other VALU", "v_mov / accvgpr moves", "ds_read_b128", "ds_read (other)", "ds_write",
         "LDS-DMA / vmem load", "s_waitcnt", "s_barrier", "s_setprio", "s_nop", "SALU (m0, addresses, loop)", "other"]
T = args.tiles_per_iteration
md = [f"| per key tile and wave | " + " | ".join(f"loop {k} (lines {a}-{b})" for k, (a, b, _n, _c) in enumerate(rows)) + " |",
      "|---|" + "---|" * len(rows)]
for name in order:
    if any(c[name] for *_x, c in rows):
        md.append(f"| {name} | " + " | ".join(f"{c[name] / T:.1f}" for *_x, c in rows) + " |")
md.append("| all instructions | " + " | ".join(f"{n / T:.1f}" for _a, _b, n, _c in rows) + " |")
md.append("| MFMA pipe cycles (16 per 16x16x32) | " + " | ".join(f"{16 * c['MFMA'] / T:.0f}" for *_x, c in rows) + " |")
txt = "\n".join(md)
print(txt)
if args.md:
    open(args.md, "w").write(txt + "\n")
