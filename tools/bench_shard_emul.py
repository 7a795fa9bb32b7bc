"""Per-rank compute time of the corpus-sharded VCMR pass at world size W, measured on ONE GPU.

The multi-GPU bench is the driver's to run; this tool answers "where would a rank's time go at W = 8?" so that the
fixed (non-scaling) stages can be attacked on a 1-GPU box.  It runs rank 0's shard (1/W of the corpus, 1/W of the
query encoding) through the same stage functions as tvretrieval_amd.dist.sharded_vcmr_search; the all-gathers are
replaced by local stand-ins of the same shape (the other ranks' top-k lists are this rank's list with jittered
scores and shifted ids, which gives the expected 1/W ownership of the global top-k).  Collective time is NOT
included (8 MB + 16 MB per rank at Nq = 10 K).

    python tools/bench_shard_emul.py [--world 8] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--workload", default="c3")
    a = ap.parse_args()
    from tvretrieval_amd import dist as xd
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops
    from tvretrieval_amd.model_xml import XML

    dev = torch.device("cuda", 0)
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, dtname = bench.WORKLOADS[a.workload]
    dtype = torch.bfloat16 if dtname == "bf16" else torch.float32
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=dtype).to(dev).eval()
    W = a.world
    lo, hi = xd.shard_range(nv, 0, W, align=bench.SHARD_ALIGN)
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(lo, hi, l, dv, ds, model.use_video, model.use_sub, dev),
                                       video_offset=lo, n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)
    per = (nq + W - 1) // W
    k, n_out = 100, 200
    acc = {}

    def timed(name, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(); e.record(); torch.cuda.synchronize()
        acc.setdefault(name, []).append(s.elapsed_time(e))
        return r

    g = torch.Generator(device=dev).manual_seed(5)
    with torch.no_grad():
        for rep in range(a.reps + 1):
            local = timed("query_encode_1/W", lambda: inf.stage_query_vectors(model, qf[:per].contiguous(), qm[:per].contiguous()))
            qvec = {m: v.repeat(W, 1)[:nq].contiguous() for m, v in local.items()}          # stand-in for the all-gather
            q2c = timed("q2c_k6", lambda: inf.stage_q2c(index, qvec))
            loc_s, loc_i = timed("topk_local_k8", lambda: ops.topk_rows(q2c, k, alpha=0.0))
            parts_s, parts_i = [loc_s], [loc_i + index.video_offset]
            for r in range(1, W):                                                           # stand-in for the all-gather
                parts_s.append(loc_s + 0.002 * torch.randn(loc_s.shape, device=dev, generator=g))
                parts_i.append(loc_i + r * index.n_videos)
            all_s, all_i = torch.cat(parts_s, 1).contiguous(), torch.cat(parts_i, 1).contiguous()
            top_w, top_gid = timed("topk_global_k8", lambda: ops.topk_rows(all_s, k, alpha=20.0, idx_in=all_i))
            own = (top_gid >= lo) & (top_gid < lo + index.n_videos)

            def prep():
                pl = torch.where(own, top_gid - lo, torch.full_like(top_gid, -1)).contiguous()
                wl = torch.where(own, top_w, torch.zeros_like(top_w)).contiguous()
                return pl, wl
            pair_local, w_local = timed("ownership_masks", prep)
            st, ed = timed("convse_k7_owned", lambda: inf.stage_span_probs(model, index, qvec, pair_local, zero_skipped=False))
            fs, fi = timed("moment_k9_owned", lambda: ops.moment_topk(st, ed, w_local, index.l_ref, 2, 16, n_out))
            all_fs = fs.repeat(1, W).contiguous()
            all_fi = fi.repeat(1, W).contiguous()
            timed("merge_topn", lambda: ops.topk_rows(all_fs, n_out, alpha=0.0, idx_in=all_fi))
    res = {k2: round(sorted(v[1:])[len(v[1:]) // 2], 3) for k2, v in acc.items()}
    res["sum_ms"] = round(sum(res.values()), 3)
    res["owned_fraction"] = round(float(own.float().mean()), 4)
    print(json.dumps(dict(world=W, videos_on_rank=index.n_videos, stage_ms=res)))


if __name__ == "__main__":
    main()
