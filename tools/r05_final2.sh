#!/bin/bash
# Round-5 second closing session on the GPU box (one gpurun call), after the training-step work: the GPU suite on the final tree, smoke(),
# the default bench line, the training step at the reference's batch size, and the rocprofv3 kernel trace of a training step
# (per-kernel table + the idle-time analysis of tools/trace_gaps.py).  Everything lands in gpurun_out/final5/.
OUT=gpurun_out/final6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err </dev/null; tail -c 400 $OUT/bench_default.json
timeout 600 python bench.py --train-only --steps 2 --warmup 1 --train-batch 8 > $OUT/bench_train_batch8.json 2> $OUT/bench_train_batch8.err </dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python bench.py --train-only --steps 2 --warmup 1 > $OUT/bench_train_prof.json 2> $OUT/bench_train_prof.err </dev/null
f=$(find $OUT/prof_train -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f $OUT/train_step_gaps.md > /dev/null 2>&1
cp $(find $OUT/prof_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
rm -rf $OUT/prof_train
echo done
