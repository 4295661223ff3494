#!/bin/bash
# Round-4 closing session on the GPU box (one gpurun call): the bench line of the final build and the rocprofv3 kernel trace + stats
# of the bench command (same flags as round 3's: one profiled candidate, eager launches so every kernel is a traced dispatch).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_r04
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated-shapes --no-graph > $OUT/bench_prof.json 2> $OUT/bench_prof.err </dev/null
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_bench.csv; rm -rf $OUT/prof
head -12 $OUT/kernel_stats_bench.csv | cut -c1-170
cut -c1-300 $OUT/bench.json
