#!/bin/bash
# Round-4 closing measurements, one GPU call:  gpurun -- bash tools/r04_final.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04/full_gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/full_gpu_suite.log
tail -3 gpurun_out/r04/full_gpu_suite.log
ROUND=r04 bash tools/measure_k6.sh > gpurun_out/r04/measure_k6.log 2>&1; tail -5 gpurun_out/r04/measure_k6.log
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/r04_k6_traffic.json gpurun_out/r04_bench_pmc.txt gpurun_out/r04_bench_kernel_stats.csv gpurun_out/r04_bench_under_rocprof.json.log profiles/ 2>/dev/null
timeout 1500 python bench.py > gpurun_out/r04_bench_c3.json.log 2> gpurun_out/r04/bench_stderr.log; echo "bench rc=$?"
tail -c 600 gpurun_out/r04_bench_c3.json.log
bash tools/trace_train.sh > gpurun_out/r04_train_timeline.txt 2>&1; head -1 gpurun_out/r04_train_timeline.txt
