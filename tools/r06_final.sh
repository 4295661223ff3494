#!/bin/bash
# Round-6 closing session on the GPU box (one gpurun call): the GPU suite on the final tree, smoke(), the default bench line, the rocprofv3
# kernel trace + stats of the bench command (eager launches so that every kernel is a traced dispatch) and of a training step, and the five
# PMC passes over the dominant kernels (tools/pmc_collect.sh; parsed in the build container by tools/pmc_parse.py r06).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_r06
mkdir -p $OUT
timeout 1800 python -m pytest tests/ -q -m gpu > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated-shapes --no-graph --no-train > $OUT/bench_prof.json 2> $OUT/bench_prof.err </dev/null
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_bench.csv; rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python bench.py --train-only --steps 2 --warmup 1 > $OUT/bench_train_prof.json 2> $OUT/bench_train_prof.err </dev/null
cp $(find $OUT/prof_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv; rm -rf $OUT/prof_train
bash tools/pmc_collect.sh > $OUT/pmc_collect.log 2>&1
head -8 $OUT/kernel_stats_bench.csv | cut -c1-170
cut -c1-300 $OUT/bench.json
echo done
