timeout 300 python -X faulthandler tools/bench_train.py --graph --steps 30 --warmup 5 > /tmp/o.log 2>&1
grep -v "^Extension" /tmp/o.log | head -60 | cut -c1-250
