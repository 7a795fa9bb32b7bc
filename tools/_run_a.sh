mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_train.py -q -m gpu -k "drops_its_batch" 2>&1 | tail -3
echo "--- same test with record_stream on the inputs disabled (must FAIL to prove the test bites)"
python - <<'PY' 2>&1 | grep "passed\|failed\|AssertionError" | head -5
import sys, os, pytest, torch
sys.path.insert(0, "tests"); sys.path.insert(0, os.getcwd())
import tvretrieval_amd.train as TR
src = open(TR.__file__).read()
orig = torch.Tensor.record_stream
import inspect
def rs(self, s):
    # skip only the calls made on caller inputs (from encode_context_train's loop over (sub_feat, sub_mask, video_mask))
    f = inspect.currentframe().f_back
    if "for t in (sub_feat, sub_mask, video_mask)" in (inspect.getsource(f.f_code) if f.f_code.co_name == "encode_context_train" else "") and f.f_lineno < 180:
        return
    return orig(self, s)
torch.Tensor.record_stream = rs
sys.exit(pytest.main(["tests/test_gpu_train.py", "-q", "-m", "gpu", "-k", "drops_its_batch"]))
PY
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04/t_train.log 2>&1; echo "rc=$?" >> gpurun_out/r04/t_train.log
tail -3 gpurun_out/r04/t_train.log
