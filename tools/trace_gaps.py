"""Idle time inside a training step, from a rocprofv3 --kernel-trace CSV of `bench.py --train-only`:
  python tools/trace_gaps.py <..._kernel_trace.csv> [out.md]
A step = the kernels between two `prodigy_apply_kernel` (or `adamw_kernel`) dispatches.  For the LAST complete step: busy time, idle
time, the idle time by the kernel in front of the gap and by gap size -- a step whose gaps sit behind runs of short kernels is bound
by the host's launch rate there, and a faster kernel buys nothing in those stretches."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "")
    for pre in ("rf::", "at::native::(anonymous namespace)::", "at::native::"):
        n = n.replace(pre, "")
    return n.split("(")[0][:70]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "prodigy_apply_kernel" in r[2] or "adamw_kernel" in r[2]]
assert len(marks) >= 2, "need two optimizer launches to delimit a step"
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
t0, t1 = step[0][0], step[-1][1]
busy = sum(e - s for s, e, _ in step)
gaps, by_kernel, hist = 0, defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
edges = [2, 5, 10, 20, 50, 100, 1000, 10 ** 9]
end = step[0][1]
for i in range(1, len(step)):
    s, e, n = step[i]
    g = s - end
    if g > 0:
        gaps += g
        k = by_kernel[short(step[i - 1][2])]
        k[0] += g
        k[1] += 1
        for ed in edges:
            if g / 1e3 <= ed:
                hist[ed][0] += g
                hist[ed][1] += 1
                break
    end = max(end, e)
big = []
end = step[0][1]
for i in range(1, len(step)):
    s_, e_, n_ = step[i]
    if s_ - end > 300_000:
        big.append((s_ - end, i))
    end = max(end, e_)
out = []
out.append(f"step window {1e-6 * (t1 - t0):.2f} ms, {len(step)} kernels: busy {1e-6 * busy:.2f} ms (sum of kernel durations), idle {1e-6 * gaps:.2f} ms "
           f"= {100.0 * gaps / (t1 - t0):.1f} % of the window")
out.append("")
out.append("| gap size | gaps | idle ms |")
out.append("|---|---|---|")
lo = 0
for ed in edges:
    out.append(f"| {lo}-{ed if ed < 10 ** 9 else 'inf'} us | {hist[ed][1]} | {1e-6 * hist[ed][0]:.2f} |")
    lo = ed
out.append("")
out.append("| kernel in front of the gap | gaps | idle ms | mean gap us |")
out.append("|---|---|---|---|")
for k, (g, c) in sorted(by_kernel.items(), key=lambda kv: -kv[1][0])[:25]:
    out.append(f"| `{k}` | {c} | {1e-6 * g:.2f} | {1e-3 * g / c:.1f} |")
out.append("")
out.append("gaps above 0.3 ms, with the three kernels on either side (index in the step / of the step's kernels):")
for g, i in big:
    out.append(f"* {1e-6 * g:.2f} ms at kernel {i} / {len(step)}: ... " + " -> ".join(f"`{short(step[j][2])[:40]}`" for j in range(max(0, i - 3), i))
               + "  **|gap|**  " + " -> ".join(f"`{short(step[j][2])[:40]}`" for j in range(i, min(len(step), i + 3))))
durs = defaultdict(lambda: [0, 0])
for s, e, n in step:
    d = durs[short(n)]
    d[0] += e - s
    d[1] += 1
out.append("")
out.append("| kernel | launches | ms | mean us |")
out.append("|---|---|---|---|")
for k, (g, c) in sorted(durs.items(), key=lambda kv: -kv[1][0])[:40]:
    out.append(f"| `{k}` | {c} | {1e-6 * g:.2f} | {1e-3 * g / c:.1f} |")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
