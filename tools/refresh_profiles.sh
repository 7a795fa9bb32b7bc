#!/bin/bash
# One gpurun call that regenerates every artefact kept under profiles/ for the current tree:
#   full bench line (with cpu_baseline), rocprofv3 kernel stats + PMC traffic of bench.py, kernel stats of the training step.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/bench_c3.json.log 2>&1
tail -1 gpurun_out/bench_c3.json.log | cut -c1-600
bash tools/prof_bench.sh > gpurun_out/prof_bench.log 2>&1
tail -5 gpurun_out/prof_bench.log
python tools/bench_train.py > gpurun_out/train_bench.json.log 2>&1
tail -1 gpurun_out/train_bench.json.log
bash tools/prof_train.sh > gpurun_out/prof_train.log 2>&1
