python -m pytest tests/test_gpu_train.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 2>&1 | tail -n 1 | cut -c1-70
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 --no-resid-cell 2>&1 | tail -n 1 | cut -c1-70
done
