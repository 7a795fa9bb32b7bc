#!/usr/bin/env python
"""Does a host-synchronised pause in front of the headline pass change its device time?  (vcmr_search on resident queries,
HIP events around the pass: back to back vs after a synchronise + idle of 0 / 5 / 20 ms.)  GPU box only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tvretrieval_amd import inference as inf  # noqa: E402
from tvretrieval_amd import ops  # noqa: E402
from tvretrieval_amd.model_xml import XML  # noqa: E402


def main():
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, True, dev, None), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)
    with torch.no_grad():
        for _ in range(3):
            inf.vcmr_search(model, index, qf, qm)
        torch.cuda.synchronize()
        for idle_ms in (None, 0, 5, 20, 100):
            ts = []
            for _ in range(6):
                if idle_ms is not None:
                    torch.cuda.synchronize()
                    time.sleep(idle_ms * 1e-3)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                qvec = inf.stage_query_vectors(model, qf, qm)
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                q2c = inf.stage_q2c(index, qvec, ops)
                e2 = torch.cuda.Event(enable_timing=True); e2.record()
                tw, ti = ops.topk_rows(q2c, 100, alpha=20.0)
                fs, fi = inf.stage_moments(model, index, qvec, tw, ti)
                e.record()
                ts.append((s, e1, e2, e))
            torch.cuda.synchronize()
            print("idle", idle_ms, "ms: pass", [round(a.elapsed_time(d), 2) for a, b, c, d in ts], "query", [round(a.elapsed_time(b), 2) for a, b, c, d in ts],
                  "k6", [round(b.elapsed_time(c), 2) for a, b, c, d in ts], flush=True)


if __name__ == "__main__":
    main()
