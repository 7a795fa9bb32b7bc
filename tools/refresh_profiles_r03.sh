#!/bin/bash
# Round-3 profile artefacts beyond tools/measure_k6.sh (one GPU call; copy gpurun_out/r03_* to profiles/):
#   the driver-style bench line (with the extra legs), encoder-only and training-step kernel stats (eager and captured),
#   exact-rank mode at the TVR shape against the plain f32 path, bf16-vs-f32 list agreement, the 8-way shard emulation.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
python $R/bench.py 2>/dev/null | tail -1 > $OUT/r03_bench_c3.json.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o enc -- python $R/tools/prof_encode.py --videos 8192 > $OUT/r03_encode.log 2>&1
cp "$(find /tmp/pe -name '*kernel_stats.csv' | head -1)" $OUT/r03_encode_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 > $OUT/r03_train_prof.log 2>&1
cp "$(find /tmp/pt -name '*kernel_stats.csv' | head -1)" $OUT/r03_train_kernel_stats.csv
python $R/tools/bench_train.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r03_train_bench.json.log
python $R/tools/bench_train.py --steps 20 --warmup 5 --graph 2>/dev/null | tail -1 >> $OUT/r03_train_bench.json.log
python $R/tools/bench_exact.py --init perturbed --compare 1000 --out $OUT/r03_exact_rank_perturbed.json > /dev/null 2>&1
python $R/tools/bench_exact.py --init reset --compare 1000 --out $OUT/r03_exact_rank_reset.json > /dev/null 2>&1
python $R/tools/rank_agreement.py --out $OUT/r03_bf16_vs_fp32_rank_agreement.json > /dev/null 2>&1
python $R/tools/bench_shard_emul.py > $OUT/r03_shard_emul.txt 2>&1
python $R/bench.py --gpus 1 --force-sharded --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r03_bench_c3_forced_sharded_1rank.json.log
tail -2 $OUT/r03_encode.log; cat $OUT/r03_train_bench.json.log | cut -c1-300; tail -5 $OUT/r03_shard_emul.txt
