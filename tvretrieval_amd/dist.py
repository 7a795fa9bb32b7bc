"""Corpus-sharded VCMR over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed path at all (its nn.DataParallel wrapper is dead code).
Videos are independent units, so the corpus is partitioned by contiguous video ranges and every per-(query, video)
quantity is computed locally.  One exchange is NOT enough for exact results: the reference keeps only moments of
the GLOBAL top-k videos of each query (xml/inference.py:347-348,365-367).  Exact two-phase scheme:

  phase 1  local K6 + local top-k  ->  all-gather (score f32, global video id i32)  ->  identical global top-k
           on every rank (same kernel, same tie rule: score desc, video id asc);
  phase 2  each rank runs ConvSE + banded moment top-n only for the global top-k videos it owns (slots of videos
           owned elsewhere are skipped), with flat indices expressed in the GLOBAL slot order
           ->  all-gather (score f32, flat i32)  ->  top-n merge (score desc, flat asc).

Candidate sets of different ranks are disjoint and their union is the single-GPU candidate set, so the merged
lists equal the single-GPU lists.  Query encoding is sharded too (each rank encodes Nq/P queries, all-gather of the
modular vectors).  Collectives carry Nq*k*8 B and Nq*n*8 B per rank (8 MB + 16 MB at Nq=10 K): issue them per
>= 1 K queries so they are bandwidth- not latency-bound on the point-to-point xGMI links.
"""
import torch
import torch.distributed as dist

from . import inference as inf
from . import ops as hip_ops


# world-size-1 groups skip the collectives; tests flip this to push the real RCCL calls (dtypes, contiguity, API use)
# through a single-rank group, which is all a 1-GPU box can exercise
SKIP_TRIVIAL_COLLECTIVES = True


def shard_range(n_total, rank, world, align=1):
    """Contiguous, balanced [lo, hi) of `n_total` items for `rank`; boundaries multiples of `align`."""
    blocks = (n_total + align - 1) // align
    base, rem = divmod(blocks, world)
    lo_b = rank * base + min(rank, rem)
    hi_b = lo_b + base + (1 if rank < rem else 0)
    return min(lo_b * align, n_total), min(hi_b * align, n_total)


def _all_gather_cat(t, group, world):
    """all-gather equal-shaped (Nq, k) tensors and lay them out as (Nq, world*k)."""
    if world == 1 and (SKIP_TRIVIAL_COLLECTIVES or not dist.is_initialized()):
        return t
    t = t.contiguous()
    out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)   # rank-major concat
    dist.all_gather_into_tensor(out, t, group=group)
    return out.view(world, t.shape[0], t.shape[1]).permute(1, 0, 2).reshape(t.shape[0], world * t.shape[1]).contiguous()


def _all_gather_rows(t, group, world):
    """all-gather equal-shaped (n, H) row blocks into (world*n, H)."""
    if world == 1 and (SKIP_TRIVIAL_COLLECTIVES or not dist.is_initialized()):
        return t
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def encode_queries_sharded(model, query_feat, query_mask, group=None):
    """Each rank encodes a contiguous 1/P slice of the (replicated) raw queries; all-gather the modular vectors."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and (SKIP_TRIVIAL_COLLECTIVES or not dist.is_initialized()):
        return inf.stage_query_vectors(model, query_feat, query_mask)
    nq = query_feat.shape[0]
    per = (nq + world - 1) // world
    lo, hi = min(rank * per, nq), min((rank + 1) * per, nq)
    out = {}
    if hi > lo:
        local = inf.stage_query_vectors(model, query_feat[lo:hi].contiguous(), query_mask[lo:hi].contiguous())
    else:   # more ranks than queries: encode one dummy row to keep shapes
        local = inf.stage_query_vectors(model, query_feat[:1].contiguous(), query_mask[:1].contiguous())
    for m, v in local.items():
        buf = v.new_zeros((per, v.shape[1]))
        if hi > lo:
            buf[:hi - lo] = v
        out[m] = _all_gather_rows(buf, group, world)[:nq].contiguous()
    return out


def sharded_vcmr_search(model, index, query_feat, query_mask, max_vcmr_video=100, max_before_nms=200,
                        q2c_alpha=20.0, min_pred_l=2, max_pred_l=16, group=None, ops=hip_ops, qvec=None):
    """Exact corpus-sharded counterpart of inference.vcmr_search.  `index` is this rank's CorpusIndex
    (index.video_offset = global id of its first video, index.n_total = corpus size).  Every rank returns the same
    global result: top_scores/top_indices (Nq, k) with GLOBAL video ids, flat_scores/flat_indices (Nq, n)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if qvec is None:
        qvec = encode_queries_sharded(model, query_feat, query_mask, group)
    k = min(max_vcmr_video, index.n_total)
    # ---- phase 1: global top-k videos ------------------------------------------------------------------
    q2c = inf.stage_q2c(index, qvec, ops)
    k_loc = min(k, index.n_videos)
    loc_s, loc_i = ops.topk_rows(q2c, k_loc, alpha=0.0)
    loc_i = loc_i + index.video_offset
    if k_loc < k:       # tiny shard: pad with -inf so that every rank contributes k slots
        pad_s = loc_s.new_full((loc_s.shape[0], k - k_loc), float("-inf"))
        pad_i = loc_i.new_full((loc_i.shape[0], k - k_loc), 2 ** 31 - 1)
        loc_s, loc_i = torch.cat([loc_s, pad_s], 1), torch.cat([loc_i, pad_i], 1)
    all_s = _all_gather_cat(loc_s, group, world)
    all_i = _all_gather_cat(loc_i, group, world)
    top_w, top_gid = ops.topk_rows(all_s, k, alpha=q2c_alpha, idx_in=all_i)
    # ---- phase 2: moments of the global top-k videos this rank owns -------------------------------------
    lo, hi = index.video_offset, index.video_offset + index.n_videos
    own = (top_gid >= lo) & (top_gid < hi)
    pair_local = torch.where(own, top_gid - lo, torch.full_like(top_gid, -1)).contiguous()
    st, ed = inf.stage_span_probs(model, index, qvec, pair_local, ops, zero_skipped=False)   # K9 skips w == 0 pairs
    w_local = torch.where(own, top_w, torch.zeros_like(top_w)).contiguous()   # w == 0 marks slots owned elsewhere:
    loc_fs, loc_fi = ops.moment_topk(st, ed, w_local, index.l_ref, min_pred_l, max_pred_l, max_before_nms)  # skipped
    all_fs = _all_gather_cat(loc_fs, group, world)
    all_fi = _all_gather_cat(loc_fi, group, world)
    if world == 1 and (SKIP_TRIVIAL_COLLECTIVES or not dist.is_initialized()):
        fs, fi = loc_fs, loc_fi
    else:
        fs, fi = ops.topk_rows(all_fs, max_before_nms, alpha=0.0, idx_in=all_fi)
        fi = torch.where(fs > 0, fi, torch.full_like(fi, -1))
    return dict(top_scores=top_w, top_indices=top_gid, flat_scores=fs, flat_indices=fi, q2c_local=q2c)
