#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_split16.py -x -q -m gpu -s > gpurun_out/r04/t2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/t2.log
tail -c 5000 gpurun_out/r04/t2.log
