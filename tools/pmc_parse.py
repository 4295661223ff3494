#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/pmc_collect.sh (gpurun_out/pmc/) into profiles/r01_pmc_kernels.{json,md}.

Conventions (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"; calibrated on these very runs):
  * FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced
    reads (16 B/lane global_load and LDS-DMA alike) -> doubled.  Infinity-Cache hits are counted, so this is
    memory-side fabric traffic, an upper bound on HBM traffic.
  * SQ_VALU_MFMA_BUSY_CYCLES = 32 x (number of v_mfma_f32_32x32x16_bf16) summed over the chip's 1024 SIMDs --
    verified: it equals 32 * flops / (2*16384) to 4 digits for every launch below.
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs -> kernel cycles = GRBM_GUI_ACTIVE / 8; clock = cycles / duration.
"""
import collections, csv, glob, json, os, sys
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"     # output prefix under profiles/
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = collections.OrderedDict()
names = ['dbl_qkv', 'dbl_out', 'dbl_ff1', 'dbl_ff2', 'sgl_in', 'sgl_out', 'attn', 'attn_lag']
for pth in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out/pmc/p*/p*_counter_collection.csv'))):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(pth)):
        kn = r['Kernel_Name']
        if not any(t in kn for t in ('rf::gemm_bf16_kernel<256', 'rf::gemm_bf16_pp_kernel', 'rf::gemm_bf16_pp16_kernel', 'rf::gemm_bf16_pp16e_kernel', 'rf::gemm_bf16_sk_kernel', 'rf::attn_fwd')):
            continue
        d = per.setdefault(int(r['Dispatch_Id']), {'name': kn, 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    disp = [per[k] for k in sorted(per)]
    assert len(disp) == 16, (pth, len(disp))
    tag = pth.split('/')[-2]
    for i, n in enumerate(names):
        d = disp[2 * i + 1]                       # second launch of each pair = measured
        e = rows.setdefault(n, {'kernel': d['name'].split('(')[0].replace('void ', '')})
        for k, v in d.items():
            if k != 'name':
                e[k if k != 'ns' else 'ns_' + tag] = v
S, D, mlp = 4608, 3072, 12288
alg = {'dbl_qkv': (S*D*2 + 2*3*D*D*2 + S*3*D*2, 2.0*S*3*D*D, 19), 'dbl_out': (S*D*2 + 2*D*D*2 + 2*S*D*2, 2.0*S*D*D, 19),
       'dbl_ff1': (S*D*2 + 2*mlp*D*2 + S*mlp*2, 2.0*S*mlp*D, 19), 'dbl_ff2': (S*mlp*2 + 2*D*mlp*2 + 2*S*D*2, 2.0*S*D*mlp, 19),
       'sgl_in': (S*D*2 + (3*D+mlp)*D*2 + S*(3*D+mlp)*2, 2.0*S*(3*D+mlp)*D, 38),
       'sgl_out': (S*(D+mlp)*2 + D*(D+mlp)*2 + 2*S*D*2, 2.0*S*D*(D+mlp), 38), 'attn': (4*S*D*2, 4.0*S*S*D, 57),
       'attn_lag': (4*S*D*2, 4.0*S*S*D, 0)}
out = collections.OrderedDict()
md = [f"# {ROUND} -- PMC counters of the dominant kernels (MI355X, rocprofv3 --pmc, one measured launch each)", "",
      "Command: `bash tools/pmc_collect.sh` (5 separate `rocprofv3 --kernel-trace --pmc ...` passes over `tools/pmc_kernels.py`), parsed by `tools/pmc_parse.py`.",
      "Shapes are the launches of one 1024^2 FLUX.1-dev forward (512 text + 4096 image tokens). Profiled runs are ~10-15 % slower than un-profiled ones.", "",
      "| launch (per forward) | kernel | us | fabric read MB (FETCH_SIZE x2) | write MB | algorithmic MB | traffic / algorithmic | MFMA busy % | clock GHz | LDS bank-conflict % |",
      "|---|---|---|---|---|---|---|---|---|---|"]
for n, e in rows.items():
    us = e['ns_p3'] / 1e3
    fetch, write = 2 * e['FETCH_SIZE'] * 1024, e['WRITE_SIZE'] * 1024
    ab, fl, cnt = alg[n]
    cyc = e['GRBM_GUI_ACTIVE'] / 8
    util = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100
    dev_ = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * fl / 32768) - 1
    assert abs(dev_) < (0.10 if n.startswith('attn') else 0.03), (n, dev_)   # attention: the mixed-size launch multiplies padded query rows (+4..6 %)
    clk = cyc / e['ns_p4']
    ldsc = e['SQ_LDS_BANK_CONFLICT'] / max(e['SQ_LDS_IDX_ACTIVE'], 1) * 100
    out[n] = dict(kernel=e['kernel'], launches_per_forward=cnt, us_profiled=round(us, 1), fabric_read_bytes=fetch, write_bytes=write,
                  algorithmic_bytes=ab, flops=fl, traffic_over_algorithmic=round((fetch + write) / ab, 2), mfma_busy_pct=round(util, 1),
                  clock_ghz=round(clk, 2), lds_bank_conflict_pct=round(ldsc, 3))
    md.append(f"| {n} (x{cnt}) | `{e['kernel'][:48]}` | {us:.1f} | {fetch/1e6:.0f} | {write/1e6:.0f} | {ab/1e6:.0f} | {(fetch+write)/ab:.2f} | {util:.1f} | {clk:.2f} | {ldsc:.3f} |")
gem = [v for k, v in out.items() if not k.startswith('attn')]
tot_n = sum(v['launches_per_forward'] for v in gem)
summary = dict(gemm_traffic_bytes_per_launch_avg=sum(v['launches_per_forward'] * (v['fabric_read_bytes'] + v['write_bytes']) for v in gem) / tot_n,
               gemm_algorithmic_bytes_per_launch_avg=sum(v['launches_per_forward'] * v['algorithmic_bytes'] for v in gem) / tot_n,
               gemm_mfma_busy_pct_weighted=sum(v['launches_per_forward'] * v['us_profiled'] * v['mfma_busy_pct'] for v in gem) /
               sum(v['launches_per_forward'] * v['us_profiled'] for v in gem))
# stamp: which kernel source these passes measured (bench.py refuses `roofline.traffic` when csrc/gemm_bf16.hip has changed since)
import hashlib, subprocess
try:
    git_sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or os.environ.get("RF_GIT_SHA", "")
except Exception:
    git_sha = os.environ.get("RF_GIT_SHA", "")
summary["measured_at"] = {"git_sha": git_sha or None,
                          "gemm_bf16_hip_sha256_16": hashlib.sha256(open(os.path.join(ROOT, "reflectionflow_amd/csrc/gemm_bf16.hip"), "rb").read()).hexdigest()[:16]}
out['_summary'] = summary
md += ["", f"GEMM kernel, forward-weighted: traffic {summary['gemm_traffic_bytes_per_launch_avg']/1e6:.0f} MB/launch vs algorithmic "
           f"{summary['gemm_algorithmic_bytes_per_launch_avg']/1e6:.0f} MB/launch; MFMA pipe busy {summary['gemm_mfma_busy_pct_weighted']:.1f} % of SIMD-cycles "
           "at the clock the chip actually sustains (1.8-2.07 GHz, not 2.4).",
       "Reading: the >2x traffic ratio is operand panels re-read by several XCDs (private L2s) and served by the Infinity Cache -- the kernels are far from "
       "the fabric limit (<= 2 TB/s of ~6) and are bound by MFMA issue efficiency, not by memory. Zero LDS bank conflicts confirms the XOR/padded layouts."]
json.dump(out, open(os.path.join(ROOT, f'profiles/{ROUND}_pmc_kernels.json'), 'w'), indent=1)
open(os.path.join(ROOT, f'profiles/{ROUND}_pmc_kernels.md'), 'w').write("\n".join(md) + "\n")
print("\n".join(md))
