#!/bin/bash
# Round-6 profile artefacts, one GPU call:  gpurun -- bash tools/refresh_profiles_r06.sh   (copy gpurun_out/r06_* to profiles/)
# (the K6 traffic / PMC / kernel-stats set of the bench command is its own call: ROUND=r06 bash tools/measure_k6.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$OUT/r06_bench_stderr.log | tail -1 > $OUT/r06_bench_c3.json.log
cd $R; bash tools/prof_encode.sh > $OUT/r06_encode_kernels.txt 2>&1; cp $OUT/encode_kernel_stats.csv $OUT/r06_encode_kernel_stats.csv
cd /tmp
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 > $OUT/r06_train_prof.log 2>&1
cp "$(find /tmp/pt -name '*kernel_stats.csv' | head -1)" $OUT/r06_train_kernel_stats.csv
TAG=r06 PMC=1 bash $R/tools/prof_tvr_val.sh > $OUT/r06_tvr_val_prof.log 2>&1
cd $R; bash tools/trace_tvr_batch.sh 2>&1 | grep -v "^W2026" > $OUT/r06_tvr_val_batch50_timeline.txt
bash tools/prof_query.sh f16s > $OUT/r06_query_f16s_kernels.txt 2>&1
python tools/bench_e2e.py --bsz 50 > $OUT/r06_e2e_tvr_val.json.log 2>/dev/null
python tools/bench_shard_emul.py > $OUT/r06_shard_emul.txt 2>&1
tail -2 $OUT/r06_encode_kernels.txt | cut -c1-200; tail -c 1600 $OUT/r06_bench_c3.json.log; echo; tail -3 $OUT/r06_shard_emul.txt | cut -c1-400
