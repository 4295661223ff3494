cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PYTHONPATH=. timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_full -- python tools/kb_train.py --layers 19 --single-layers 38 --res 512 --cond 512 --iters 2 --optimizer </dev/null > gpurun_out/prof_train_full.log 2>&1
f=$(find gpurun_out/prof_train_full -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/train_full_512_kernel_stats.csv; rm -rf gpurun_out/prof_train_full; head -40 gpurun_out/train_full_512_kernel_stats.csv | cut -c1-160
tail -25 gpurun_out/prof_train_full.log | head -40
