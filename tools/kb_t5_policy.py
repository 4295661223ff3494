#!/usr/bin/env python3
"""Which GEMM schedule policy runs a WHOLE T5-XXL encode fastest (weights HBM-cold: 9.4 GB pass between two uses of a panel)?
`tools/kb_t5_shapes.py` times each shape warm and alone; this tool times the real 24-layer encode under a patched `text_gemm` policy.
Every variant is a textual patch of a temporary copy of csrc/ built into its own librf_flux.so and run in its own process.

    python tools/kb_t5_policy.py [--reps 20]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ANCHOR = "    if (t256 >= 16 && t256 < 200 && K / 64 >= 16) d.schedule = RF_SCHED_STREAMK;\n"
VARIANTS = {
    "streamk_16_200 (shipped through round 5)": ANCHOR,
    "auto": "    (void)t256;\n",
    "auto_but_tile256_from_128_tiles": "    if (t256 >= 128 && t256 < 256) d.schedule = RF_SCHED_TILE256;\n",
    "streamk_only_under_48_tiles": "    if (t256 >= 16 && t256 < 48 && K / 64 >= 16) d.schedule = RF_SCHED_STREAMK;\n",
    "streamk_48_to_200": "    if (t256 >= 48 && t256 < 200 && K / 64 >= 16) d.schedule = RF_SCHED_STREAMK;\n",
}
CHILD = r"""
import json, sys, time, torch
sys.path.insert(0, %(root)r)
from reflectionflow_amd import _lib
_lib.LIB_PATH = %(so)r
import bench
from reflectionflow_amd import ops
from reflectionflow_amd.flux.text_hip import HipT5Encoder
dev = torch.device("cuda:0")
sd = bench.synthetic_t5_xxl_state(dev)
t5 = HipT5Encoder(sd, 64, dev)
out = {}
for B in (1, 4):
    ids = torch.randint(0, 32128, (B, 512), device=dev, generator=torch.Generator(device=dev).manual_seed(B))
    for _ in range(3):
        y = t5.encode(ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(%(reps)d):
        y = t5.encode(ids)
    e1.record(); torch.cuda.synchronize()
    out["ms_B%%d" %% B] = round(e0.elapsed_time(e1) / %(reps)d, 3)
    out["sum_B%%d" %% B] = float(y.float().abs().sum())
with ops.profile(4096) as pr:
    t5.encode(torch.randint(0, 32128, (1, 512), device=dev)); torch.cuda.synchronize()
out["classes_B1"] = {k: {"launches": v["launches"], "ms": round(v["us"] / 1e3, 3)} for k, v in pr.classes.items()}
print("RESULT " + json.dumps(out), flush=True)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    work = tempfile.mkdtemp(prefix="rf_t5_policy_")
    try:
        res = {}
        for name, repl in VARIANTS.items():
            d = os.path.join(work, str(len(res)))
            shutil.copytree(os.path.join(ROOT, "reflectionflow_amd", "csrc"), os.path.join(d, "reflectionflow_amd", "csrc"))
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
            src = os.path.join(d, "reflectionflow_amd", "csrc", "text.hip")
            s = open(src).read()
            cur = ANCHOR if ANCHOR in s else None
            assert cur is not None or name == "auto", "patch anchor not found in csrc/text.hip (the policy changed: update ANCHOR)"
            if cur is not None:
                s = s.replace(cur, repl)
            open(src, "w").write(s)
            r = subprocess.run(["make", "-C", os.path.dirname(src), "-j16"], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            so = os.path.join(d, "reflectionflow_amd", "librf_flux.so")
            c = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": so, "reps": args.reps}], capture_output=True, text=True, cwd=ROOT)
            line = [l for l in c.stdout.splitlines() if l.startswith("RESULT ")]
            assert line, c.stderr[-2000:]
            res[name] = json.loads(line[-1][7:])
            print(f"{name:45s} B=1 {res[name]['ms_B1']:7.3f} ms   B=4 {res[name]['ms_B4']:7.3f} ms ({res[name]['ms_B4'] / 4:6.3f} per prompt)   "
                  f"{res[name]['classes_B1']}", flush=True)
        print(json.dumps(res))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
