import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
import bench_train
bench_train.run(steps=3, warmup=3)
pr = cProfile.Profile()
pr.enable()
r = bench_train.run(steps=20, warmup=0)
pr.disable()
print(r["ms_per_step"], r["breakdown_ms"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
