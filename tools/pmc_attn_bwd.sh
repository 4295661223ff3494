#!/bin/bash
# rocprofv3 --pmc passes (one counter group each) over tools/pmc_attn_bwd.py; table -> gpurun_out/pmc_attn_bwd.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ab
mkdir -p $OUT
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p$i -- python tools/pmc_attn_bwd.py > $OUT/p$i.log 2>&1 </dev/null
  echo "pass $i ($ctr): rc=$?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for pth in sorted(glob.glob('gpurun_out/pmc_ab/p*/p*_counter_collection.csv')):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(pth)):
        kn = r['Kernel_Name']
        if 'attn' not in kn: continue
        d = per.setdefault(int(r['Dispatch_Id']), {'name': kn, 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    disp = [per[k] for k in sorted(per)]
    # per shape: 2 iterations x (fwd, prep, dq, dkv); keep the second iteration of each shape
    for si in range(2):
        blk = disp[si * 8 + 4: si * 8 + 8]
        for d in blk:
            short = 'fwd' if 'fwd' in d['name'] else 'prep' if 'prep' in d['name'] else 'dq' if '_dq_' in d['name'] else 'dkv'
            e = agg.setdefault((si, short), {})
            for k, v in d.items():
                if k != 'name': e[k] = v
shapes = [(24, 5632), (24, 2560)]
lines = ["| shape | kernel | us (profiled) | MFMA busy / CU-busy cycles | LDS bank-conflict / LDS active cycles | fabric read MB (FETCH x2 x 1 KiB) | write MB |", "|---|---|---|---|---|---|---|"]
for (si, short), e in agg.items():
    H, S = shapes[si]
    mf = e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(e.get('SQ_BUSY_CU_CYCLES', 1), 1)
    lds = e.get('SQ_LDS_BANK_CONFLICT', 0) / max(e.get('SQ_LDS_IDX_ACTIVE', 1), 1)
    lines.append(f"| {H} x {S} | {short} | {e.get('ns', 0) / 1e3:.0f} | {mf:.3f} | {lds:.3f} | {2 * e.get('FETCH_SIZE', 0) * 1024 / 1e6:.0f} | {e.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f} |")
open('gpurun_out/pmc_attn_bwd.md', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/p*/
