#!/bin/bash
# Round-2 profile artefacts beyond tools/measure_k6.sh: encoder-only and training-step kernel stats, bench lines of the
# ragged workload.  One GPU call; copy gpurun_out/r02_* to profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o enc -- python $R/tools/prof_encode.py --videos 8192 > $OUT/r02_encode.log 2>&1
cp "$(find /tmp/pe -name '*kernel_stats.csv' | head -1)" $OUT/r02_encode_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 > $OUT/r02_train_prof.log 2>&1
cp "$(find /tmp/pt -name '*kernel_stats.csv' | head -1)" $OUT/r02_train_kernel_stats.csv
python $R/tools/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/r02_train_bench.json.log
python $R/bench.py --workload c3r --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_bench_c3r.json.log
XML_Q2C_NO_BUCKETS=1 python $R/bench.py --workload c3r --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_bench_c3r_unbucketed.json.log
python $R/bench.py --workload c2 2>/dev/null | tail -1 > $OUT/r02_bench_c2.json.log
python $R/bench.py --gpus 1 --force-sharded --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r02_bench_c3_forced_sharded_1rank.json.log
tail -2 $OUT/r02_encode.log; cat $OUT/r02_train_bench.json.log | cut -c1-400
