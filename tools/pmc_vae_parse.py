#!/usr/bin/env python3
"""Aggregate the rocprofv3 PMC passes over tools/kb_vae_hip.py (gpurun_out/pmc_vae/p*/..counter_collection.csv) by kernel:
fabric read / write bytes per decode-sized launch class, MFMA busy.  Conventions as tools/pmc_parse.py (FETCH_SIZE x2 in KiB)."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for pth in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out/pmc_vae/p*/p*_counter_collection.csv'))):
    for r in csv.DictReader(open(pth)):
        kn = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        if not kn.startswith('rf::'):
            continue
        agg[kn][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[kn][r['Counter_Name']] += 1
        if r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'FETCH_SIZE'):
            agg[kn]['ns_' + r['Counter_Name']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
rows = []
for kn, a in agg.items():
    n = max(cnt[kn].values())
    fetch = 2 * a.get('FETCH_SIZE', 0) * 1024 / max(cnt[kn].get('FETCH_SIZE', 1), 1)
    write = a.get('WRITE_SIZE', 0) * 1024 / max(cnt[kn].get('WRITE_SIZE', 1), 1)
    cyc = a.get('GRBM_GUI_ACTIVE', 0) / 8
    busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc) * 100 if cyc else 0.0
    us = a.get('ns_GRBM_GUI_ACTIVE', 0) / max(cnt[kn].get('GRBM_GUI_ACTIVE', 1), 1) / 1e3
    clk = cyc / a['ns_GRBM_GUI_ACTIVE'] if a.get('ns_GRBM_GUI_ACTIVE') else 0
    rows.append(dict(kernel=kn, launches=n, avg_us=round(us, 1), fabric_read_MB_per_launch=round(fetch / 1e6, 1), write_MB_per_launch=round(write / 1e6, 1),
                     mfma_busy_pct=round(busy, 1), clock_ghz=round(clk, 2)))
rows.sort(key=lambda r: -r['avg_us'] * r['launches'])
md = ["# r03 -- PMC counters of the VAE path (tools/kb_vae_hip.py: decode 1024^2 + encode 512^2, all launches of the run, averaged per kernel)", "",
      "`bash tools/r03_final.sh` -> separate `rocprofv3 --kernel-trace --pmc ...` passes; FETCH_SIZE doubled (gfx950), fabric bytes incl. Infinity-Cache hits.", "",
      "| kernel | launches | avg us (profiled) | fabric read MB / launch | write MB / launch | MFMA busy % | clock GHz |", "|---|---|---|---|---|---|---|"]
for r in rows:
    md.append(f"| `{r['kernel'][:70]}` | {r['launches']} | {r['avg_us']} | {r['fabric_read_MB_per_launch']} | {r['write_MB_per_launch']} | {r['mfma_busy_pct']} | {r['clock_ghz']} |")
open(os.path.join(ROOT, 'profiles/r03_pmc_vae.md'), 'w').write("\n".join(md) + "\n")
json.dump(rows, open(os.path.join(ROOT, 'profiles/r03_pmc_vae.json'), 'w'), indent=1)
print("\n".join(md))
