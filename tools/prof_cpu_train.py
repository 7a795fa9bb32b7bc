"""Host-side profile of the eager training step (main thread: forward + optimizer; the backward functions run in autograd's
device thread and show up only as run_backward).   gpurun -- python tools/prof_cpu_train.py"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch  # noqa: E402,F401
import bench_train  # noqa: E402
import tvretrieval_amd.train as TR  # noqa: E402

orig = TR.xml_forward_train
pr = cProfile.Profile()
calls = [0]


def wrapped(*a, **k):
    calls[0] += 1
    if calls[0] <= 5:
        return orig(*a, **k)
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()


TR.xml_forward_train = wrapped
bench_train.xml_forward_train = wrapped if hasattr(bench_train, "xml_forward_train") else None
r = bench_train.run(steps=40, warmup=5)
print(r["ms_per_step"], r["breakdown_ms"], "profiled forward calls:", calls[0] - 5)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
