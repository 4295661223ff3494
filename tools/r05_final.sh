#!/bin/bash
# Round-5 closing session on the GPU box (one gpurun call): the GPU suite, the bench line of the final build and the rocprofv3 kernel trace +
# stats of the bench command (one profiled candidate, eager launches so every kernel is a traced dispatch).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_r05
mkdir -p $OUT
timeout 1500 python -m pytest tests/ -q -m gpu > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated-shapes --no-graph --no-train > $OUT/bench_prof.json 2> $OUT/bench_prof.err </dev/null
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_bench.csv; rm -rf $OUT/prof
head -8 $OUT/kernel_stats_bench.csv | cut -c1-170
cut -c1-400 $OUT/bench.json
