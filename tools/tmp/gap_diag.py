import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
from tvretrieval_amd import inference as inf, ops
from tvretrieval_amd.model_xml import XML
_, _, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
cfg = bench.model_config(hidden, dv, ds, dq, ctx_mode, l)
dev = torch.device("cuda")
torch.manual_seed(0)
model = XML(cfg, compute_dtype=torch.bfloat16).to(dev).eval()
nq, nv = 10000, 21793
qf, qm = bench.synth_queries(nq, dq, dev)
with torch.no_grad():
    raw = list(bench.context_batches(0, nv, l, dv, ds, True, True, dev))
    index = inf.build_corpus_index(model, iter(raw), n_total=nv, l_ref=l, n_videos=nv)
    del raw
    stamps = []
    orig = ops.pack_plan
    def pp(mask):
        t0 = time.perf_counter(); r = orig(mask); t1 = time.perf_counter()
        stamps.append(("pack_plan blocked", (t1 - t0) * 1e3))
        return r
    ops.pack_plan = pp
    for _ in range(3):
        inf.vcmr_search(model, index, qf, qm)
    torch.cuda.synchronize()
    stamps.clear()
    t_prev = time.perf_counter()
    for i in range(4):
        t0 = time.perf_counter()
        inf.vcmr_search(model, index, qf, qm)
        t1 = time.perf_counter()
        stamps.append(("step %d cpu time" % i, (t1 - t0) * 1e3))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    stamps.append(("4 steps wall", (t2 - t_prev) * 1e3))
for s in stamps:
    print("%-24s %8.3f ms" % s)
