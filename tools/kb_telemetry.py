"""Power / clock / throttle attribution of the two MFMA kernels ("settle the GEMM ceiling with instruments").

For each workload a sustained loop (default 5 s) runs between two `amd-smi metric` snapshots; the firmware's own accumulators say
which limiter held the clock and the energy counter gives joules:

    TF/s | avg W (energy counter) | J per TFLOP | in-kernel shader clock (s_memtime / s_memrealtime) | PPT-limited share of the time |
    per-XCD "gfx clock below host limit" shares (power / thermal / total)

Workloads: a bare v_mfma_f32_16x16x32_bf16 loop (tools/ubench, no memory traffic) when the ubench binary is built, torch.matmul
(hipBLASLt) and rf_gemm_bf16 at 8192^3 and at the cfg2 launch shapes, rf_attention at S = 4608 and 17920.
    python tools/kb_telemetry.py [--secs 5] > profiles/r04_telemetry.json"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                     # noqa: E402
from reflectionflow_amd import _lib, ops              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--secs", type=float, default=5.0)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--sched", type=int, default=-1, help="rf_gemm_schedule to ALSO time rf_gemm_bf16 with (e.g. 6 = RF_SCHED_W4)")
args = ap.parse_args()
dev = torch.device("cuda:0")
BF = torch.bfloat16
_lib.load()


def sustained(name, fn, flops, secs):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    time.sleep(1.0)                                                   # let the previous workload's power state decay
    s0 = B.amdsmi_snapshot(0)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = B.amdsmi_snapshot(0)
    mhz, _ = B.clock_probe(1 if name.startswith("attention") else 0)
    d = B.amdsmi_delta(s0, s1) or {}
    tf = flops * n / dt / 1e12
    row = {"workload": name, "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500, 4), "launches": n, "seconds": round(dt, 2),
           "in_kernel_shader_mhz": round(mhz) if mhz else None, **d}
    if d.get("energy_j"):
        row["joule_per_tflop"] = round(d["energy_j"] / (flops * n / 1e12), 4)
    print(json.dumps(row), flush=True)
    return row


rows = []
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)   # noqa: E731
want = lambda k: not args.only or args.only in k                  # noqa: E731
for M, N, K, tag in [(8192, 8192, 8192, "8192^3"), (4608, 21504, 3072, "sgl_in"), (4608, 12288, 3072, "dbl_ff1"), (4608, 3072, 12288, "dbl_ff2"),
                     (4608, 9216, 3072, "dbl_qkv")]:
    x, W = r(M, K), r(N, K, sc=0.02)
    out = torch.empty(M, N, device=dev, dtype=BF)
    g = [ops.Group([ops.Seg(x, W)], out=out)]
    fl = 2.0 * M * N * K
    if want("rf_gemm " + tag):
        rows.append(sustained(f"rf_gemm_bf16 {tag} ({M}x{N}x{K})", lambda: ops.gemm(g, N), fl, args.secs))
    if args.sched >= 0 and want("rf_gemm " + tag):
        with ops.gemm_schedule(args.sched):
            rows.append(sustained(f"rf_gemm_bf16[sched {args.sched}] {tag} ({M}x{N}x{K})", lambda: ops.gemm(g, N), fl, args.secs))
    if want("torch " + tag):
        rows.append(sustained(f"torch.matmul (hipBLASLt) {tag}", lambda: torch.matmul(x, W.t(), out=out), fl, args.secs))
    del x, W, out
for S, H in [(4608, 24), (17920, 24)]:
    if not want("attention"):
        continue
    q, k, vt, s_pad = ops.alloc_attn_operands(H, S, dev)
    q[:, :S] = (torch.randn(H, S, 128, device=dev) * ops.QK_PRESCALE).to(BF)
    k[:, :S] = torch.randn(H, S, 128, device=dev).to(BF)
    vt.copy_(torch.randn(vt.shape, device=dev).to(BF))
    o = torch.empty(S, H * 128, dtype=BF, device=dev)
    rows.append(sustained(f"attention S={S} x {H} heads (auto kernel, bound 25)", lambda: ops.attention(q, k, vt, S, out=o, q_prescaled=True, score_bound=25.0),
                          4.0 * S * S * 128 * H, args.secs))
    del q, k, vt, o
print(json.dumps({"rows": rows}))
