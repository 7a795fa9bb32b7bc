mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r04/t_train.log 2>&1; echo "rc=$?" >> gpurun_out/r04/t_train.log
tail -3 gpurun_out/r04/t_train.log
for i in 1 2; do
timeout 300 python tools/bench_train.py --graph --steps 30 --warmup 5 > gpurun_out/r04/train_adam_$i.log 2>&1
tail -n 1 gpurun_out/r04/train_adam_$i.log | cut -c1-100
done
echo base; python tools/bench_ln_bwd.py 2>&1 | grep -v amdgpu
echo noatom; XML_EXP_NOATOM=1 python tools/bench_ln_bwd.py 2>&1 | grep "layernorm_bwd"
for r in 2 8; do echo rpw $r;  XML_EXP_RPW=$r python tools/bench_ln_bwd.py 2>&1 | grep "layernorm_bwd"; done
echo rpw 8 noatom; XML_EXP_RPW=8 XML_EXP_NOATOM=1 python tools/bench_ln_bwd.py 2>&1 | grep "layernorm_bwd"
