#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mm
mkdir -p $OUT
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p$i -- python tools/pmc_mm_compare.py > $OUT/p$i.log 2>&1 </dev/null
  echo "pass $i ($ctr): rc=$?"
done
python - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for pth in sorted(glob.glob('gpurun_out/pmc_mm/p*/p*_counter_collection.csv')):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(pth)):
        kn = r['Kernel_Name']
        if not ('Cijk' in kn or 'rf::gemm' in kn): continue
        d = per.setdefault(int(r['Dispatch_Id']), {'name': kn[:60], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    disp = [per[k] for k in sorted(per)]
    for i, d in enumerate(disp):
        if i % 2 == 1:
            e = rows.setdefault(i // 2, {'name': d['name']})
            e.update({k: v for k, v in d.items() if k != 'name'})
shapes = [(4608,21504,3072),(4608,12288,3072),(4608,3072,12288),(8192,8192,8192)]
for i, e in rows.items():
    M,N,K = shapes[i // 3]
    alg = (M*K + N*K + M*N) * 2
    f = 2*e.get('FETCH_SIZE',0)*1024; w = e.get('WRITE_SIZE',0)*1024
    print(f"{M}x{N}x{K} {e['name'][:44]:44s} us {e['ns']/1e3:7.1f} fetch {f/1e6:7.0f} MB write {w/1e6:6.0f} MB  traffic/alg {(f+w)/alg:5.2f}  mfma_busy {e.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6:8.1f}M lds_idx {e.get('SQ_LDS_IDX_ACTIVE',0)/1e6:8.1f}M conf {e.get('SQ_LDS_BANK_CONFLICT',0)/1e6:6.1f}M tcp_req {e.get('TCP_TCC_READ_REQ_sum',0)/1e6:7.1f}M tcc_hit {e.get('TCC_HIT_sum',0)/1e6:7.1f}M miss {e.get('TCC_MISS_sum',0)/1e6:7.1f}M")
PY
