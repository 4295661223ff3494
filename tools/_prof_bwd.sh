cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PYTHONPATH=. timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -- python tools/kb_train.py --res 1024 --cond 512 --iters 3 </dev/null > gpurun_out/prof_bwd.log 2>&1
f=$(find gpurun_out/prof_bwd -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200
