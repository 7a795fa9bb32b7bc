timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "graphed or golden" 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 2>&1 | tail -n 1 | cut -c1-70
done
bash tools/trace_train.sh > gpurun_out/r04_train_timeline.txt 2>&1; head -1 gpurun_out/r04_train_timeline.txt
