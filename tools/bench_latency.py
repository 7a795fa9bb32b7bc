"""Latency of one small query batch (the reference's eval_query_bsz = 50) over the full C3 corpus:
eager kernel chain vs one HIP-graph replay (inference.GraphedVcmrSearch).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 21793
    _, _, l, hidden, dv, ds, dq, ctx_mode, dtname = bench.WORKLOADS["c3"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, True, dev), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        eager = timeit(lambda: inf.vcmr_search(model, index, qf, qm))
        g = inf.GraphedVcmrSearch(model, index, nq, qf.shape[1], dq)
        graph = timeit(lambda: g(qf, qm))
    print(json.dumps(dict(queries_per_batch=nq, videos=nv, eager_ms=round(eager, 3), hip_graph_ms=round(graph, 3),
                          eager_qps=round(nq / eager * 1e3, 1), hip_graph_qps=round(nq / graph * 1e3, 1))))


if __name__ == "__main__":
    main()
