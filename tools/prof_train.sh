#!/bin/bash
# rocprofv3 kernel stats of the training step; summaries only are copied back (raw traces stay in /tmp on the box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 "$@" > /tmp/prof_train.log 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/train_kernel_stats.csv
tail -3 /tmp/prof_train.log | cut -c1-300
head -40 $R/gpurun_out/train_kernel_stats.csv | cut -c1-170
