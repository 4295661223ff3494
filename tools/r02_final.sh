#!/bin/bash
# Round-2 measurement session on the GPU box (one gpurun call): rocprofv3 kernel trace + stats of the bench command,
# PMC passes of the dominant kernels, the cfg4 side bench.
export TMPDIR=/tmp
OUT=gpurun_out/final
mkdir -p $OUT
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated-shapes > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_prof.err )
bash tools/pmc_collect.sh > $OUT/pmc_collect.log 2>&1
python tools/pmc_parse.py r02 > $OUT/pmc_parse.log 2>&1
python tools/bench_cfg4.py > $OUT/cfg4.json 2> $OUT/cfg4.err
find $OUT/prof -name "*stats*" | head; find $OUT/prof -name "*kernel_stats.csv" -exec head -12 {} \;
cut -c1-300 $OUT/bench.json; tail -12 $OUT/pmc_parse.log; cat $OUT/cfg4.json
