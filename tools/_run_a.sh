mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r04/t_train.log 2>&1; echo "rc=$?" >> gpurun_out/r04/t_train.log
for i in 1 2; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 > gpurun_out/r04/train_tail_$i.log 2>&1
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 --separate-loss-tail > gpurun_out/r04/train_notail_$i.log 2>&1
done
tail -3 gpurun_out/r04/t_train.log; for f in gpurun_out/r04/train_*tail_?.log; do echo $f; tail -n 1 $f | cut -c1-120; done
