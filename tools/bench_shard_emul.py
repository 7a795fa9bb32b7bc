"""Per-rank compute time of the corpus-sharded VCMR pass at world size W, measured on ONE GPU.

The multi-GPU bench is the driver's to run; this tool answers "where would a rank's time go at W = 8?" so that the
fixed (non-scaling) stages can be attacked on a 1-GPU box.  It runs rank 0's shard (1/W of the corpus, 1/W of the
query encoding) through the same stage functions as tvretrieval_amd.dist.sharded_vcmr_search; the all-gathers are
replaced by local stand-ins of the same shape (the other ranks' top-k lists are this rank's list with jittered
scores and shifted ids, which gives the expected 1/W ownership of the global top-k).  Collective time is NOT
included (a rank receives 31 + 16 + 16 MB per pass at Nq = 10 K).  `chain_async_ms` is the same chain launched
without a host synchronisation between the stages, as the real pass runs.

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
    ap.add_argument("--sharded-rerank", action="store_true", help="phase 2 on the video's owner (feat2 sharded too)")
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
    index.feat2_all, index.mask_all = index.feat2, index.mask
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

    def chain(timed):
        local = timed("query_encode_1/W", lambda: inf.stage_query_vectors(model, qf[:per].contiguous(), qm[:per].contiguous()))
        qvec = {m: v.repeat(W, 1)[:nq].contiguous() for m, v in local.items()}          # stand-in for the all-gather
        q2c = timed("q2c_k6", lambda: inf.stage_q2c(index, qvec))
        loc_s, loc_i = timed("topk_local_k8", lambda: ops.topk_rows(q2c, k, alpha=0.0))

        def pack1():
            return xd._pack_rows(loc_s, loc_i + index.video_offset, W * per, float("-inf"), 2 ** 31 - 1)
        buf = timed("pack_topk", pack1)
        # stand-in for the all-to-all: my slice's rows of the other ranks = mine, jittered, ids shifted
        mine_s, mine_i = loc_s[:per], loc_i[:per] + index.video_offset
        parts_s = [mine_s] + [mine_s + 0.002 * torch.randn(mine_s.shape, device=dev, generator=g) for _ in range(1, W)]
        parts_i = [mine_i + r * index.n_videos for r in range(W)]
        recv = torch.stack([xd._pack_rows(a, b, per, 0.0, -1) for a, b in zip(parts_s, parts_i)], 0)

        def unpack1():
            cand = recv.view(W, per, 2, k).permute(2, 1, 0, 3).contiguous()
            return cand[0].reshape(per, W * k).view(torch.float32), cand[1].reshape(per, W * k)
        cand_s, cand_i = timed("unpack_topk", unpack1)
        own_w, own_gid = timed("merge_topk_k8", lambda: ops.topk_rows(cand_s, k, alpha=20.0, idx_in=cand_i))
        if not a.sharded_rerank:     # owner rerank: my query slice x its global top-k against the (replicated) feat2;
            qv = {m: v[:per] for m, v in qvec.items()}      # the local shard stands in for the corpus-wide copy
            gid = (own_gid % index.n_videos).contiguous()
            st, ed = timed("convse_k7_slice", lambda: inf.stage_span_probs(model, index, qv, gid, replicated=True))
            timed("moment_k9_slice", lambda: ops.moment_topk(st, ed, own_w, index.l_ref, 2, 16, n_out))
            return None
        top_w = own_w.repeat(W, 1)[:nq].contiguous()                                    # stand-in for the all-gather
        top_gid = own_gid.repeat(W, 1)[:nq].contiguous()
        own = (top_gid >= lo) & (top_gid < lo + index.n_videos)

        def prep():
            pl = torch.where(own, top_gid - lo, torch.full_like(top_gid, -1)).contiguous()
            wl = torch.where(own, top_w, torch.zeros_like(top_w)).contiguous()
            return pl, wl
        pair_local, w_local = timed("ownership_masks", prep)
        st, ed = timed("convse_k7_owned", lambda: inf.stage_span_probs(model, index, qvec, pair_local, zero_skipped=False))
        fs, fi = timed("moment_k9_owned", lambda: ops.moment_topk(st, ed, w_local, index.l_ref, 2, 16, n_out))
        buf2 = timed("pack_moments", lambda: xd._pack_rows(fs, fi, W * per, 0.0, -1))
        recv2 = buf2[:per].unsqueeze(0).repeat(W, 1, 1, 1).contiguous()                  # stand-in for the all-to-all

        def unpack2():
            cand = recv2.view(W, per, 2, n_out).permute(2, 1, 0, 3).contiguous()
            return cand[0].reshape(per, W * n_out).view(torch.float32), cand[1].reshape(per, W * n_out)
        c_s, c_i = timed("unpack_moments", unpack2)
        timed("merge_moments_k8", lambda: ops.topk_rows(c_s, n_out, alpha=0.0, idx_in=c_i))
        return own

    chain_ms = []
    with torch.no_grad():
        for rep in range(a.reps + 1):
            own = chain(timed)
        for rep in range(a.reps + 1):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            chain(lambda name, fn: fn())
            e.record()
            torch.cuda.synchronize()
            chain_ms.append(s.elapsed_time(e))
    res = {k2: round(sorted(v[1:])[len(v[1:]) // 2], 3) for k2, v in acc.items()}
    res["sum_ms"] = round(sum(res.values()), 3)
    if own is not None:
        res["owned_fraction"] = round(float(own.float().mean()), 4)
    res["rerank"] = "video owner" if a.sharded_rerank else "query owner"
    res["chain_async_ms"] = round(sorted(chain_ms[1:])[len(chain_ms[1:]) // 2], 3)
    print(json.dumps(dict(world=W, videos_on_rank=index.n_videos, stage_ms=res)))


if __name__ == "__main__":
    main()
