import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvretrieval_amd import inference as inf
from tvretrieval_amd.model_xml import XML
nq, nv, l, hidden, dv, ds, dq, ctx_mode, dtname = bench.WORKLOADS[os.environ.get("Q_WORKLOAD", "c3")]
dev = torch.device("cuda", 0)
cfg = bench.model_config(hidden, dv, ds, dq, ctx_mode, l)
from tvretrieval_amd import ops
# "f16s": the exact-rank mode's split-f16 model (f32 activations, every projection three f16 products)
model = XML(cfg, compute_dtype=ops.F16S if "f16s" in sys.argv else torch.bfloat16).to(dev).eval()
qf, qm = bench.synth_queries(nq, dq, dev)
with torch.no_grad():
    for _ in range(6):
        inf.stage_query_vectors(model, qf, qm)
    if "ctx" in sys.argv:
        for b in bench.context_batches(0, 1024, l, dv, ds, True, True, dev):
            model.encode_context(*b)
torch.cuda.synchronize()
