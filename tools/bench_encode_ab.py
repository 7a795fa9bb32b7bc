#!/usr/bin/env python
"""Corpus-encode throughput (videos/s, HIP events around build_corpus_index on resident raw features) of the headline model
for A/B runs between package trees: XML_PKG_ROOT=<tree> python tools/bench_encode_ab.py [n_videos].  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("XML_PKG_ROOT", ROOT))
sys.path.insert(1, ROOT)
import bench  # noqa: E402
from tvretrieval_amd import inference as inf  # noqa: E402
from tvretrieval_amd.model_xml import XML  # noqa: E402


def main():
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    raw = list(bench.context_batches(0, nv, l, dv, ds, True, True, dev, None))
    with torch.no_grad():
        for _ in range(2):
            inf.build_corpus_index(model, raw, n_total=nv, l_ref=l)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            inf.build_corpus_index(model, raw, n_total=nv, l_ref=l)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
    ts.sort()
    print("tree %s: %d videos, median %.2f ms -> %.0f videos/s (best %.0f)" % (os.environ.get("XML_PKG_ROOT", "."), nv, ts[2], nv / ts[2] * 1e3, nv / ts[0] * 1e3))


if __name__ == "__main__":
    main()
