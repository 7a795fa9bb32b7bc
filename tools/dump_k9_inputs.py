"""GPU box: run the bench pipeline once and dump the K9 inputs of the first 64 queries (analysis aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvretrieval_amd import inference as inf
from tvretrieval_amd.model_xml import XML
nq, nv, l, hidden, dv, ds, dq, ctx_mode, dtname = bench.WORKLOADS["c3"]
nv = 4096
dev = torch.device("cuda", 0)
cfg = bench.model_config(hidden, dv, ds, dq, ctx_mode, l)
torch.manual_seed(0)
model = XML(cfg, compute_dtype=torch.bfloat16).to(dev).eval()
with torch.no_grad():
    index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, True, dev), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(256, dq, dev)
    qvec = inf.stage_query_vectors(model, qf, qm)
    q2c = inf.stage_q2c(index, qvec)
    tw, ti = inf.hip_ops.topk_rows(q2c, 100, alpha=20.0)
    st, ed = inf.stage_span_probs(model, index, qvec, ti)
torch.save(dict(st=st[:64].cpu(), ed=ed[:64].cpu(), w=tw[:64].cpu(), q2c=q2c[:8].cpu()), "gpurun_out/k9_inputs.pt")
print("w row0", tw[0, :5].tolist(), tw[0, -3:].tolist(), "st max/min", float(st[0, 0].max()), float(st[0, 0].min()))
