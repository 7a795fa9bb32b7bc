"""The reference's as-trained shape (bench.WORKLOADS["tvr_val"]: TVR val, 10 895 queries x 2 179 videos, H = 256,
max_ctx_l = 100, real clip counts) served the way the reference serves it -- batches of eval_query_bsz = 50 queries
(xml/config.py) -- through inference.GraphedVcmrSearch: per-batch latency (host-synchronised, what a caller waits for) and
the rate of the 218 batches replayed back to back.  The one-pass throughput and K6's roofline fraction at K = 256 come
from bench.run(workload="tvr_val"); bench.py's extras leg reports both.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=50, exact=False):
    import bench
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["tvr_val"]
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    dt = torch.float32 if exact else torch.bfloat16
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=dt).to(dev).eval()
    lens = bench.real_clip_counts(nv, l)
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, True, dev, lens), n_total=nv,
                                       l_ref=l, **(dict(exact_filter=True) if exact else {}))
    qf, qm = bench.synth_queries(nq, dq, dev)
    n_b = (nq + batch - 1) // batch
    pad = n_b * batch - nq
    if pad:          # the last batch is filled up with copies of the first queries (a graph has one shape)
        qf, qm = torch.cat([qf, qf[:pad]]), torch.cat([qm, qm[:pad]])
    with torch.no_grad():
        g = inf.GraphedVcmrSearch(model, index, batch, qf.shape[1], dq)
        for b in range(3):
            g(qf[b * batch:(b + 1) * batch], qm[b * batch:(b + 1) * batch])
        torch.cuda.synchronize()
        lat = []
        for b in range(n_b):         # latency: submit one batch, wait for its lists
            t0 = time.perf_counter()
            out = g(qf[b * batch:(b + 1) * batch], qm[b * batch:(b + 1) * batch])
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        for b in range(n_b):         # rate: batches replayed back to back, one wait at the end
            out = g(qf[b * batch:(b + 1) * batch], qm[b * batch:(b + 1) * batch])
        torch.cuda.synchronize()
        stream_s = time.perf_counter() - t0
        # eager chain of the same batch, for the launch-overhead comparison
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(20):
            inf.vcmr_search(model, index, qf[b * batch:(b + 1) * batch].contiguous(), qm[b * batch:(b + 1) * batch].contiguous())
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / 20 * 1e3
    lat = np.asarray(lat)
    return {"batch": batch, "batches": n_b, "mode": "exact-rank" if exact else "bf16",
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)),
                           "p99": float(np.percentile(lat, 99)), "mean": float(lat.mean())},
            "eager_batch_ms": eager_ms,
            "back_to_back_queries_per_s": nq / stream_s, "back_to_back_ms_per_batch": stream_s / n_b * 1e3,
            "index_lpad": index.lpad, "bucketed": getattr(index.feat1n[index.modalities[0]], "plan", None) is not None,
            "what": "10 895 queries in %d batches of %d through one HIP graph per batch shape (GraphedVcmrSearch), "
                    "2 179 videos, H=256, max_ctx_l=100" % (n_b, batch)}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 50)))
