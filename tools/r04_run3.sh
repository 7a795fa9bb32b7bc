#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_split16.py -x -q -m gpu -s -k "rescore or exact_mode" > gpurun_out/r04/t3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/t3.log
tail -c 2500 gpurun_out/r04/t3.log
for f in bf16 f16; do
timeout 1200 python tools/bench_exact.py --mode f16s --filter $f --compare 0 --steps 5 --out gpurun_out/r04/exact_f16s_$f.json > gpurun_out/r04/exact_f16s_$f.log 2>&1
python - <<PY
import json
r=json.load(open("gpurun_out/r04/exact_f16s_$f.json"))
print("$f", r["ms_per_pass"], r["candidates"], r["stage_ms"], r["certificate"]["fail_rate"], r["certificate"]["eps_mean"])
PY
done
