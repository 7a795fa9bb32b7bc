mkdir -p gpurun_out/r04
for i in 1 2 3; do
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 > gpurun_out/r04/eager_shadow_$i.log 2>&1
timeout 300 python tools/bench_train.py --steps 30 --warmup 5 --no-shadows > gpurun_out/r04/eager_noshadow_$i.log 2>&1
done
timeout 300 python tools/prof_cpu_train.py > gpurun_out/r04/prof_cpu_train.log 2>&1
for f in gpurun_out/r04/eager_*.log; do echo $f; tail -n 1 $f | cut -c1-80; done
