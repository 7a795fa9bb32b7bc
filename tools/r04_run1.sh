#!/bin/bash
# round 4, first GPU pass: new tests, as-trained shape, encode timing with preallocated index
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graph or pad_tail or preallocated" > gpurun_out/r04/t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/t1.log
timeout 600 python bench.py --workload tvr_val --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r04/tvr_val.log 2>&1
timeout 600 python tools/bench_tvr_val.py > gpurun_out/r04/tvr_val_lat.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r04/c3.log 2>&1
tail -c 600 gpurun_out/r04/t1.log; tail -c 1500 gpurun_out/r04/tvr_val.log; tail -c 1200 gpurun_out/r04/tvr_val_lat.log; tail -c 2500 gpurun_out/r04/c3.log
