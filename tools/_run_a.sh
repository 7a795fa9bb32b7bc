for i in 1 2; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 --one-stream 2>&1 | tail -n 1 | cut -c1-70
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 2>&1 | tail -n 1 | cut -c1-70
done
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 --one-stream 2>&1 | tail -n 1 | cut -c1-70
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 2>&1 | tail -n 1 | cut -c1-70
