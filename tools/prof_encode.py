"""Encoder-only workload for rocprofv3: corpus encode of N videos (C3 shape, bf16) + query encode of 10 000 queries.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_enc -o enc -- python tools/prof_encode.py [--videos 4096]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-queries", action="store_true")
    a = ap.parse_args()
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd import _lib
    lib = _lib.load()
    if hasattr(lib, "xml_debug_set_q2c_ablation") and os.environ.get("XML_ABL"):      # debug library: kernel A/B switches
        import ctypes
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(int(os.environ["XML_ABL"])))
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    raw = list(bench.context_batches(0, a.videos, l, dv, ds, True, True, dev))
    qf, qm = bench.synth_queries(nq, dq, dev)
    with torch.no_grad():
        inf.build_corpus_index(model, iter(raw[:1]), l_ref=l)
        if not a.no_queries:
            inf.stage_query_vectors(model, qf, qm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            inf.build_corpus_index(model, iter(raw), l_ref=l)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(0 if a.no_queries else a.reps):
            inf.stage_query_vectors(model, qf, qm)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("corpus encode: %.1f videos/s   query encode: %.3f ms per %d queries"
          % (a.videos * a.reps / (t1 - t0), (t2 - t1) / a.reps * 1e3, nq))


if __name__ == "__main__":
    main()
