#!/bin/bash
# rocprofv3 PMC passes for the dominant kernels (run on the GPU box via gpurun). Counters in their own passes,
# only with --kernel-trace (MI355X_MICROARCH.md "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE cannot share a pass.
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p $OUT
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p$i -- python tools/pmc_kernels.py > $OUT/p$i.log 2>&1 </dev/null
  echo "pass $i ($ctr): rc=$?"
done
find $OUT -name "*.csv" | head -20
for f in $(find $OUT -name "*counter_collection.csv"); do echo "== $f"; head -3 "$f" | cut -c1-300; grep -c . "$f"; done
