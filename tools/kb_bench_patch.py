#!/usr/bin/env python3
"""Same-box, in-sequence A/B of a csrc/ patch through bench.py itself: every variant is a TEXTUAL patch of a temporary copy of csrc/
built into its own librf_flux.so; bench.py runs in a fresh process per measurement with `_lib.LIB_PATH` pointed at it (A B A B ...).
Reports latents/s, the in-sequence GEMM / attention class times and the sha-256 of the timed latent (is the variant bit-identical?).

    python tools/kb_bench_patch.py --variants base,sk_from_two_rounds [--rounds 3] [--steps 3]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sub(s, old, new):
    assert s.count(old) == 1, f"patch anchor found {s.count(old)}x: {old[:80]!r}"
    return s.replace(old, new)


def gemm(fn):
    return ("gemm_bf16.hip", fn)


PATCHES = {
    "base": [],
    # AUTO takes stream-K below 83 % round fill; cfg2's 648- and 864-tile launches (2.53 / 3.375 rounds) sit at 84.4 %: let launches of
    # at least two rounds in at < 86 % (their stream-K region is the last partial round only; the one-round 216-tile shapes stay out)
    "sk_from_two_rounds": [gemm(lambda s: sub(s, "  if (force_sk < 0 && !persistent_only && (double)T / ((double)rounds * P) >= 0.83) return 0;",
                                              "  if (force_sk < 0 && !persistent_only && (double)T / ((double)rounds * P) >= (rounds >= 3 ? 0.86 : 0.83)) return 0;"))],
    "sk_from_two_rounds_b": [gemm(lambda s: sub(sub(s, "  if (force_sk < 0 && !persistent_only && (double)T / ((double)rounds * P) >= 0.83) return 0;",
                                                     "  if (force_sk < 0 && !persistent_only && (double)T / ((double)rounds * P) >= (rounds >= 3 ? 0.86 : 0.83)) return 0;"),
                                                "                                                 (double)T / ((double)rounds * P) >= 0.83);",
                                                "                                                 (double)T / ((double)rounds * P) >= (rounds >= 3 ? 0.86 : 0.83));"))],
}
CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
from reflectionflow_amd import _lib
_lib.LIB_PATH = %(so)r
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-train", "--no-vae", "--no-text", "--no-attention-table", "--no-isolated-shapes",
            "--steps", "%(steps)d", "--warmup", "1"]
import bench
bench.main()
"""


def build(name, patches, work):
    d = os.path.join(work, name)
    shutil.copytree(os.path.join(ROOT, "reflectionflow_amd", "csrc"), os.path.join(d, "reflectionflow_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    for fn, p in patches:
        src = os.path.join(d, "reflectionflow_amd", "csrc", fn)
        text = p(open(src).read())
        open(src, "w").write(text)
    if not patches:
        os.utime(os.path.join(d, "reflectionflow_amd", "csrc", "capi.hip"))
    r = subprocess.run(["make", "-C", os.path.join(d, "reflectionflow_amd", "csrc"), "-j16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return os.path.join(d, "reflectionflow_amd", "librf_flux.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base,sk_from_two_rounds")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    names = args.variants.split(",")
    work = tempfile.mkdtemp(prefix="rf_bench_patch_")
    try:
        sos = {n: build(n, PATCHES[n], work) for n in names}
        print("built", names, flush=True)
        if args.build_only:
            return
        res = {n: [] for n in names}
        for rnd in range(args.rounds):
            for n in names:
                c = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": sos[n], "steps": args.steps}], capture_output=True, text=True, cwd=ROOT)
                line = [l for l in c.stdout.splitlines() if l.startswith('{"metric"')]
                assert line, c.stderr[-2000:]
                d = json.loads(line[-1])
                cl = d["roofline"]["classes"]
                rec = {"value": d["value"], "gemm_ms": cl["gemm_main"]["ms_per_forward"], "attn_ms": cl["attention"]["ms_per_forward"],
                       "sha": (d.get("timed_latent_parity") or {}).get("sha256_timed")}
                res[n].append(rec)
                print(rnd, n, rec, flush=True)
        for n in names:
            v = sorted(r["value"] for r in res[n])
            g = sorted(r["gemm_ms"] for r in res[n])
            print(f"{n:28s} latents/s median {v[len(v) // 2]:.5f} (all {v})   gemm_main ms/forward median {g[len(g) // 2]:.3f}   sha {res[n][0]['sha']}")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
