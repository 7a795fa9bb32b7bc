#!/bin/bash
# Round-5 profile artefacts, one GPU call:  gpurun -- bash tools/refresh_profiles_r05.sh   (copy gpurun_out/r05_* to profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
ROUND=r05 bash tools/measure_k6.sh > $OUT/r05_measure_k6.log 2>&1; tail -4 $OUT/r05_measure_k6.log
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$OUT/r05_bench_stderr.log | tail -1 > $OUT/r05_bench_c3.json.log
rm -rf /tmp/pe; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o enc -- python $R/tools/prof_encode.py --videos 8192 > $OUT/r05_encode.log 2>&1
cp "$(find /tmp/pe -name '*kernel_stats.csv' | head -1)" $OUT/r05_encode_kernel_stats.csv
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 > $OUT/r05_train_prof.log 2>&1
cp "$(find /tmp/pt -name '*kernel_stats.csv' | head -1)" $OUT/r05_train_kernel_stats.csv
TAG=r05 PMC=1 bash $R/tools/prof_tvr_val.sh > $OUT/r05_tvr_val_prof.log 2>&1
cd $R; bash tools/trace_tvr_batch.sh > $OUT/r05_tvr_val_batch50_timeline.txt 2>&1
python tools/bench_e2e.py --bsz 50 > $OUT/r05_e2e_tvr_val.json.log 2>/dev/null
python tools/bench_ingest.py > $OUT/r05_ingest.json.log 2>/dev/null
python tools/bench_shard_emul.py > $OUT/r05_shard_emul.txt 2>&1
tail -2 $OUT/r05_encode.log; tail -c 400 $OUT/r05_bench_c3.json.log; echo; tail -3 $OUT/r05_shard_emul.txt | cut -c1-400
