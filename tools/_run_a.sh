mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r04/t_train.log 2>&1; echo "rc=$?" >> gpurun_out/r04/t_train.log
for i in 1 2; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 > gpurun_out/r04/train_fused_$i.log 2>&1
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 --separate-dropout > gpurun_out/r04/train_sep_$i.log 2>&1
done
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 > gpurun_out/r04/train_fused_eager.log 2>&1
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 --separate-dropout > gpurun_out/r04/train_sep_eager.log 2>&1
tail -3 gpurun_out/r04/t_train.log; tail -1 gpurun_out/r04/train_*.log
