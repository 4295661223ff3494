#!/bin/bash
# Round-3 measurement session on the GPU box (one gpurun call): the bench line, rocprofv3 kernel trace + stats of the bench command,
# PMC passes of the dominant kernels and of the VAE path, the VAE call-site trace (no MIOpen kernel), the cfg4 side bench.
export TMPDIR=/tmp
OUT=gpurun_out/final4
mkdir -p $OUT
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated-shapes --no-graph > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_prof.err )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_vae_sites -o sites -- python $GRAFT_REPO_ROOT/tools/trace_vae_callsites.py > $GRAFT_REPO_ROOT/$OUT/vae_sites.log 2>&1 )
bash tools/pmc_collect.sh > $OUT/pmc_collect.log 2>&1
python tools/pmc_parse.py r03 > $OUT/pmc_parse.log 2>&1
mkdir -p gpurun_out/pmc_vae
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_vae/p$i -o p$i -- python tools/kb_vae_hip.py > gpurun_out/pmc_vae/p$i.log 2>&1 </dev/null
  echo "vae pmc pass $i ($ctr): rc=$?"
done
python tools/pmc_vae_parse.py > $OUT/pmc_vae_parse.log 2>&1
python tools/bench_cfg4.py > $OUT/cfg4.json 2> $OUT/cfg4.err
find $OUT/prof -name "*kernel_stats.csv" -exec head -14 {} \;
find $OUT/prof_vae_sites -name "*kernel_stats.csv" -exec head -30 {} \;
cut -c1-400 $OUT/bench.json; tail -12 $OUT/pmc_parse.log; tail -14 $OUT/pmc_vae_parse.log; cat $OUT/cfg4.json | cut -c1-300; tail -2 $OUT/vae_sites.log
