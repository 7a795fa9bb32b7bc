#!/usr/bin/env python
"""Where the chunked host-to-host pass loses time against the resident single pass: stage times of vcmr_search at several
query-chunk sizes on the headline index (c3), and vcmr_search_host for several chunkings.  GPU box only."""
import json
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
    if len(sys.argv) > 1:
        nv = int(sys.argv[1])
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, True, dev, None), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            r = fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n, r
    out = {}
    with torch.no_grad():
        for n in (10000, 5120, 4096, 2048, 1024):
            q, m = qf[:n].contiguous(), qm[:n].contiguous()
            t_all, _ = timed(lambda: inf.vcmr_search(model, index, q, m))
            t_q, qvec = timed(lambda: inf.stage_query_vectors(model, q, m))
            t_k6, q2c = timed(lambda: inf.stage_q2c(index, qvec, ops))
            t_k8, (tw, ti) = timed(lambda: ops.topk_rows(q2c, 100, alpha=20.0))
            t_k7, (st, ed, sm) = timed(lambda: inf.stage_span_probs(model, index, qvec, ti, ops, pair_w=tw, band=(2, 16)))
            t_k9, _ = timed(lambda: ops.moment_topk(st, ed, tw, index.l_ref, 2, 16, 200, summ=sm))
            out[n] = dict(all=t_all, per_10k=t_all * 10000 / n, query=t_q, k6=t_k6, k8=t_k8, k7=t_k7, k9=t_k9)
            print(n, json.dumps({k: round(v, 3) for k, v in out[n].items()}), flush=True)
        # host-to-host with different chunkings
        valid = qm > 0
        lens = valid.sum(1).cpu()
        row_start = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)]).to(torch.int64).pin_memory()
        rows16 = qf[valid].to(torch.float16).cpu().pin_memory()
        host = {"f16 ragged": dict(query_feat=rows16, row_start=row_start, lq=30),
                "f32 padded": dict(query_feat=qf.cpu().pin_memory(), query_mask=qm.cpu().pin_memory())}
        for name, hk in host.items():
            for chunk, growth in ((1024, 3), (2048, 3), (1024, 100), (512, 4), (100000, 1)):
                tm = {}
                inf.vcmr_search_host(model, index, chunk=chunk, chunk_growth=growth, **hk)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    inf.vcmr_search_host(model, index, chunk=chunk, chunk_growth=growth, timings=tm, **hk)
                dt = (time.perf_counter() - t0) / 5 * 1e3
                print("h2h", name, "first chunk", chunk, "growth", growth, "ms", round(dt, 2),
                      {k: round(v * 1e3, 2) if isinstance(v, float) else v for k, v in tm.items()}, flush=True)

if __name__ == "__main__":
    main()
