#!/bin/bash
# second PMC series: where do the waves of the two dominant kernels spend their cycles?
export TMPDIR=/tmp
OUT=gpurun_out/pmc2
mkdir -p $OUT
i=0
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/p$i -o p$i -- python tools/pmc_kernels.py > $OUT/p$i.log 2>&1 </dev/null
  echo "pass $i: rc=$?"
done
