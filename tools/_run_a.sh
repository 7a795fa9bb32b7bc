cd "$GRAFT_REPO_ROOT"
bash tools/tail_traffic.sh > /dev/null 2>&1; head -16 gpurun_out/r04_tail_traffic.txt | cut -c1-130
bash tools/prof_train.sh --graph > /dev/null 2>&1; cp gpurun_out/train_kernel_stats.csv gpurun_out/r04_train_kernel_stats.csv; wc -l gpurun_out/r04_train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python $GRAFT_REPO_ROOT/tools/prof_encode.py > /tmp/prof_enc.log 2>&1
f=$(find /tmp/prof_enc -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04_encode_kernel_stats.csv
tail -2 /tmp/prof_enc.log | cut -c1-200; head -8 $GRAFT_REPO_ROOT/gpurun_out/r04_encode_kernel_stats.csv | cut -c1-150
