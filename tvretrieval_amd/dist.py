"""Corpus-sharded VCMR over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed path at all (its nn.DataParallel wrapper is dead code).
Videos are independent units, so the corpus is partitioned by contiguous video ranges and every per-(query, video)
quantity is computed locally.  One exchange is NOT enough for exact results: the reference keeps only moments of
the GLOBAL top-k videos of each query (xml/inference.py:347-348,365-367).  Exact two-phase scheme,
with the merges partitioned by QUERY OWNER (rank r owns the contiguous query slice r: it encodes those queries and
merges their candidate lists), so that no rank receives or merges a list it does not need:

  phase 0  each rank encodes its query slice  ->  all-gather of the modular query vectors (both modalities, one call);
  phase 1  local K6 + local top-k for ALL queries  ->  all-to-all (score f32, global video id i32: the rows of slice r
           go to rank r)  ->  the owner merges P lists into the global top-k (same kernel, same tie rule: score desc,
           video id asc)  ->  all-gather of the merged (weight, video id) rows: every rank knows the global top-k;
  phase 2  each rank runs ConvSE + banded moment top-n only for the global top-k videos it owns (slots of videos
           owned elsewhere are skipped), with flat indices expressed in the GLOBAL slot order
           ->  all-to-all (score f32, flat i32) by query owner  ->  top-n merge (score desc, flat asc) on the owner.

Candidate sets of different ranks are disjoint and their union is the single-GPU candidate set, so the merged
lists equal the single-GPU lists.  The final lists of a query live on its owner rank (where the per-query temporal NMS
runs); `gather_results=True` adds one all-gather so that every rank holds all of them.

Owner rerank (replicate_rerank_features): the similarity operand feat1n stays sharded, but the ConvSE-side features
feat2 (4.3 GB per modality at TVR scale) fit on every 288 GB GPU many times over.  With a corpus-wide copy resident,
phase 2 needs no exchange at all: the owner of a query runs K7 + K9 for its slice over the global top-k it has just
merged.  Two collectives per pass (query vectors, local top-k) instead of four, K7 / K9 work exactly 1/P of the
single-GPU work, and the result is computed by the same kernels on the same inputs as the single-GPU pass.
Per rank and pass a rank RECEIVES Nq*(2*H*2 + 2*k*8 + n*8) bytes (31 + 16 + 16 MB at Nq = 10 K, H = 768, k = 100,
n = 200) in 4 collectives; the earlier all-gather-everything scheme received 31 + 64 + 128 MB in 6 and merged every
query's lists on every rank.
"""
import torch
import torch.distributed as dist

from . import inference as inf
from . import ops as hip_ops


# world-size-1 groups skip the collectives; tests flip this to push the real RCCL calls (dtypes, contiguity, API use)
# through a single-rank group, which is all a 1-GPU box can exercise
SKIP_TRIVIAL_COLLECTIVES = True

# optional stage-boundary callback f(name) (bench.py records a HIP event per call to split a pass into stages,
# collectives included); None in normal operation
STAGE_MARK = None


def _mark(name):
    if STAGE_MARK is not None:
        STAGE_MARK(name)


def shard_range(n_total, rank, world, align=1):
    """Contiguous, balanced [lo, hi) of `n_total` items for `rank`; boundaries multiples of `align`."""
    blocks = (n_total + align - 1) // align
    base, rem = divmod(blocks, world)
    lo_b = rank * base + min(rank, rem)
    hi_b = lo_b + base + (1 if rank < rem else 0)
    return min(lo_b * align, n_total), min(hi_b * align, n_total)


def query_slice(nq, rank, world):
    """[lo, hi) of the queries owned by `rank` and the (uniform) slice length `per` used for padding."""
    per = (nq + world - 1) // world
    lo = min(rank * per, nq)
    return lo, min(lo + per, nq), per


def _trivial(world):
    return world == 1 and (SKIP_TRIVIAL_COLLECTIVES or not dist.is_initialized())


def _pack_rows(score, idx, rows, pad_score, pad_idx):
    """(n, c) f32 scores + (n, c) i32 ids -> one (rows, 2, c) i32 buffer (score bits in [:, 0]); rows >= n are padding."""
    n, c = score.shape
    buf = torch.empty((rows, 2, c), dtype=torch.int32, device=score.device)
    buf[:n, 0] = score.contiguous().view(torch.int32)
    buf[:n, 1] = idx
    if rows > n:
        buf[n:, 0] = torch.tensor(pad_score, dtype=torch.float32).view(torch.int32).item()
        buf[n:, 1] = pad_idx
    return buf


def _host_staged(t, group):
    """gloo has no device collectives for all-to-all (and only some for all-gather): ranks that keep their tensors on a GPU
    but talk over gloo -- the two-ranks-on-one-GPU test configuration -- stage through host memory."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all_single(out, inp, group):
    if _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)


def _all_gather_into_tensor(out, inp, group):
    if _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _exchange_by_owner(buf, group, world, per):
    """buf (world*per, 2, c): rows of slice r go to rank r.  Returns the candidates of MY slice from every rank as
    (score (per, world*c) f32, ids (per, world*c) i32), source-rank-major within a row."""
    c = buf.shape[2]
    out = torch.empty_like(buf)
    _all_to_all_single(out, buf, group)
    cand = out.view(world, per, 2, c).permute(2, 1, 0, 3).contiguous()          # (2, per, world, c)
    return cand[0].reshape(per, world * c).view(torch.float32), cand[1].reshape(per, world * c)


def check_shards(index, group=None):
    """Collective validation of the shard layout, once per index: every rank learns every rank's shard size, and an empty
    shard raises on ALL ranks.  (Raised only where the shard is empty, the error would leave the other ranks waiting in
    the next collective: a hang instead of a failure.)"""
    if getattr(index, "_shards_checked", False):
        return
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if _trivial(world):
        sizes = [index.n_videos]
    else:
        mine = torch.tensor([index.n_videos], dtype=torch.int64, device=index.device)
        allv = torch.empty((world,), dtype=torch.int64, device=index.device)
        _all_gather_into_tensor(allv, mine, group)
        sizes = [int(x) for x in allv.cpu().tolist()]
    index._shards_checked = True
    empty = [r for r, n in enumerate(sizes) if n == 0]
    if empty:
        raise ValueError("rank(s) %s hold an empty corpus shard (n_total=%d over %d ranks): use fewer ranks"
                         % (empty, index.n_total, world))


def replicate_rerank_features(index, group=None):
    """One-off, after the shard is encoded: all-gather the ConvSE-side context features (feat2) and clip masks of
    every shard into corpus-wide copies index.feat2_all[m] (n_total, lpad, H) / index.mask_all[m] (n_total, lpad).
    The similarity operand feat1n -- the 85 % of the pass -- stays sharded.  With the copies resident, the owner of a
    query reranks its global top-k videos itself (sharded_vcmr_search, owner_rerank): the second exchange and the
    broadcast of the global top-k disappear.  Costs n_total*lpad*H*2 B per modality and GPU (4.3 GB at TVR scale,
    of 288 GB)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    check_shards(index, group)
    if _trivial(world):
        index.feat2_all, index.mask_all, index.vlen_all = index.feat2, index.mask, index.vlen
        return index
    dev = index.device
    meta = torch.tensor([index.video_offset, index.n_videos], dtype=torch.int64, device=dev)
    metas = torch.empty((world * 2,), dtype=torch.int64, device=dev)
    _all_gather_into_tensor(metas, meta, group)
    metas = metas.view(world, 2).cpu().tolist()
    n_max = max(n for _, n in metas)
    feat2_all, mask_all = {}, {}

    def gather(src):
        pad = src.new_zeros((n_max,) + tuple(src.shape[1:]))
        pad[:src.shape[0]] = src
        out = src.new_empty((world * n_max,) + tuple(src.shape[1:]))
        _all_gather_into_tensor(out, pad, group)
        full = src.new_empty((index.n_total,) + tuple(src.shape[1:]))
        for r, (off, n) in enumerate(metas):
            full[off:off + n] = out[r * n_max:r * n_max + n]
        return full
    for m in index.modalities:
        f2 = index.feat2[m]
        if hasattr(f2, "inv"):       # split-f16 rows (exact-rank mode on an ops.F16S model): the halves and the row scales
            feat2_all[m] = type(f2)(gather(f2.data), gather(f2.inv))
        else:
            feat2_all[m] = gather(f2)
        mask_all[m] = gather(index.mask[m])
    index.feat2_all, index.mask_all = feat2_all, mask_all
    if index.vlen is not None:        # valid lengths of the whole corpus; ragged anywhere = ragged for the owner's K7 / K9
        index.vlen_all = gather(index.vlen)
        index.ragged = bool((index.vlen_all < index.l_ref).any().item())
    return index


class TorchExchange(object):
    """The two exchanges of a sharded pass through torch.distributed: the CPU tests' gloo groups, and the fallback when
    the C-ABI path is not available.  Same results as RcclExchange by construction (same merge kernel, same tie rule)."""
    name = "torch.distributed"

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def allgather_rows(self, buf):
        out = torch.empty((self.world * buf.shape[0],) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
        _all_gather_into_tensor(out, buf.contiguous(), self.group)
        return out

    def allgather_topk(self, loc_s, loc_i, k, alpha, ops):
        """Global top-k of ALL queries on every rank (the plain all-gather scheme)."""
        nq, c = loc_s.shape
        gs = self.allgather_rows(loc_s.contiguous()).view(self.world, nq, c).permute(1, 0, 2).reshape(nq, self.world * c)
        gi = self.allgather_rows(loc_i.contiguous()).view(self.world, nq, c).permute(1, 0, 2).reshape(nq, self.world * c)
        return ops.topk_rows(gs.contiguous(), k, alpha=alpha, idx_in=gi.contiguous())

    def topk_by_owner(self, loc_s, loc_i, k, alpha, ops):
        nq = loc_s.shape[0]
        q_lo, q_hi, per = query_slice(nq, self.rank, self.world)
        cand_s, cand_i = _exchange_by_owner(_pack_rows(loc_s, loc_i, self.world * per, float("-inf"), 2 ** 31 - 1),
                                            self.group, self.world, per)
        own_w, own_gid = ops.topk_rows(cand_s, k, alpha=alpha, idx_in=cand_i)
        return own_w[:q_hi - q_lo], own_gid[:q_hi - q_lo]


class RcclExchange(object):
    """The same two exchanges inside libxmlhip.so (xml_rccl_allgather, xml_rccl_topk_by_owner: grouped send/recv +
    un-permute + top-k, three launches) on an ncclComm_t of our own."""
    name = "libxmlhip RCCL (C ABI)"

    def __init__(self, group=None):
        from .rccl import RcclComm
        self.comm = RcclComm(group)
        self.world, self.rank = self.comm.world, self.comm.rank

    def allgather_rows(self, buf):
        return self.comm.allgather(buf.contiguous())

    def allgather_topk(self, loc_s, loc_i, k, alpha, ops):
        return self.comm.allgather_topk(loc_s.contiguous(), loc_i.contiguous(), k, alpha)

    def topk_by_owner(self, loc_s, loc_i, k, alpha, ops):
        return self.comm.topk_by_owner(loc_s.contiguous(), loc_i.contiguous(), k, alpha)


_EXCHANGES = {}
FORCE_TORCH_EXCHANGE = False        # tests / A-B runs: keep GPU ranks on torch.distributed collectives


def default_exchange(device, group=None):
    """RcclExchange for GPU ranks (created once per group: ncclCommInitRank is collective), TorchExchange otherwise."""
    key = (id(group), str(device))
    ex = _EXCHANGES.get(key)
    if ex is None:
        use_rccl = torch.device(device).type == "cuda" and dist.is_initialized() and not FORCE_TORCH_EXCHANGE \
            and dist.get_backend(group) == "nccl"
        ex = RcclExchange(group) if use_rccl else TorchExchange(group)
        _EXCHANGES[key] = ex
    return ex


def encode_queries_sharded(model, query_feat, query_mask, group=None, exchange=None):
    """Each rank encodes its contiguous 1/P slice of the (replicated) raw queries; ONE all-gather carries the modular
    vectors of all modalities."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if _trivial(world):
        return inf.stage_query_vectors(model, query_feat, query_mask)
    ex = exchange or default_exchange(query_feat.device, group)
    nq = query_feat.shape[0]
    lo, hi, per = query_slice(nq, rank, world)
    if hi > lo:
        local = inf.stage_query_vectors(model, query_feat[lo:hi].contiguous(), query_mask[lo:hi].contiguous())
    else:   # more ranks than queries: encode one dummy row to keep shapes
        local = inf.stage_query_vectors(model, query_feat[:1].contiguous(), query_mask[:1].contiguous())
    names = sorted(local)
    v0 = local[names[0]]
    hdim = v0.shape[1]
    buf = v0.new_zeros((per, len(names), hdim))
    if hi > lo:
        for j, m in enumerate(names):
            buf[:hi - lo, j] = local[m]
    out = ex.allgather_rows(buf)
    return {m: out[:nq, j].contiguous() for j, m in enumerate(names)}


def _local_scores_topk(index, qvec, k, ops):
    """K6 over this shard + its local top-k: (q2c or None, raw scores (Nq, k), GLOBAL video ids (Nq, k)).
    Exact-rank shards (index.exact: f32 model, bf16 filter image): the local top-k comes out of the filter / re-score /
    certificate chain (inference.stage_exact_topk) -- exact per shard, and the global f32 top-k is a subset of the union of
    the shards' exact top-k lists, so the owner's merge (same scores, same tie rule) yields the f32 path's global list."""
    if index.exact is not None:
        k_loc = min(k, index.n_videos)
        loc_s, loc_i, info = inf.stage_exact_topk(index, qvec, k_loc, 0.0, ops)
        _mark("exact_local_topk")
        return None, _pad_local(loc_s, loc_i + index.video_offset, k, k_loc)
    q2c = inf.stage_q2c(index, qvec, ops)
    _mark("q2c_k6")
    loc = _local_topk(index, q2c, k, ops)
    _mark("topk_local_k8")
    return q2c, loc


def _pad_local(loc_s, loc_i, k, k_loc):
    if k_loc < k:       # tiny shard: pad with -inf so that every rank contributes k slots
        pad_s = loc_s.new_full((loc_s.shape[0], k - k_loc), float("-inf"))
        pad_i = loc_i.new_full((loc_i.shape[0], k - k_loc), 2 ** 31 - 1)
        loc_s, loc_i = torch.cat([loc_s, pad_s], 1), torch.cat([loc_i, pad_i], 1)
    return loc_s.contiguous(), loc_i.contiguous()


def _local_topk(index, q2c, k, ops):
    """Local top-k of every query over this shard, with GLOBAL video ids, padded to k slots per rank."""
    k_loc = min(k, index.n_videos)
    loc_s, loc_i = ops.topk_rows(q2c, k_loc, alpha=0.0)
    loc_i = loc_i + index.video_offset
    if k_loc < k:       # tiny shard: pad with -inf so that every rank contributes k slots
        pad_s = loc_s.new_full((loc_s.shape[0], k - k_loc), float("-inf"))
        pad_i = loc_i.new_full((loc_i.shape[0], k - k_loc), 2 ** 31 - 1)
        loc_s, loc_i = torch.cat([loc_s, pad_s], 1), torch.cat([loc_i, pad_i], 1)
    return loc_s.contiguous(), loc_i.contiguous()


def _owner_pass(model, index, qvec, ex, k, n_out, q2c_alpha, min_pred_l, max_pred_l, ops, n_chunks):
    """Owner-rerank pass over query chunks, software-pipelined on GPUs: the exchange of chunk c (comm stream: grouped
    send/recv + merge) runs under K6 of chunk c + 1; K7 / K9 of chunk c are issued behind K6 of chunk c + 1.
    Returns per-chunk results and the global query ids this rank owns (chunk-wise slices)."""
    names = sorted(qvec)
    nq = qvec[names[0]].shape[0]
    world, rank = ex.world, ex.rank
    n_chunks = max(1, min(int(n_chunks), nq))
    # chunk boundaries on multiples of 2048 queries when there are enough of them: K6 walks query groups of 8 x 256 rows
    # per XCD, a chunk that ends inside a group would leave part of the chip idle for that group
    align = 2048 if nq >= 2 * 2048 * n_chunks else 1
    bounds = [min(nq, ((nq * c) // n_chunks + align // 2) // align * align) for c in range(n_chunks)] + [nq]
    dev = qvec[names[0]].device
    cuda = dev.type == "cuda" and n_chunks > 1
    main = torch.cuda.current_stream(dev) if cuda else None
    side = torch.cuda.Stream(dev) if cuda else None
    parts, owned, q2c_parts = [], [], []

    def finish(p):
        c_lo, qv_c, own, ev = p
        if ev is not None:
            main.wait_event(ev)
        own_w, own_gid = own
        n_c = qv_c[names[0]].shape[0]
        o_lo, o_hi, _ = query_slice(n_c, rank, world)
        if o_hi > o_lo:
            qv = {m: v[o_lo:o_hi] for m, v in qv_c.items()}
            top_w, top_gid = own_w.contiguous(), own_gid.contiguous()
            vl = inf.ragged_lengths(index, ops, replicated=True)
            rk = dict(pair_vid=top_gid, vid_len=vl) if vl is not None else {}
            st, ed = inf.stage_span_probs(model, index, qv, top_gid, ops, replicated=True, vid_len=vl)
            _mark("convse_k7")
            fs, fi = ops.moment_topk(st, ed, top_w, index.l_ref, min_pred_l, max_pred_l, n_out, **rk)
            _mark("moment_k9")
        else:   # more ranks than queries in this chunk
            top_w, top_gid = own_w[:0], own_gid[:0]
            fs, fi = own_w.new_zeros((0, n_out)), own_gid.new_full((0, n_out), -1)
        parts.append((top_w, top_gid, fs, fi))
        owned.append(torch.arange(c_lo + o_lo, c_lo + o_hi, device=dev))

    pending = None
    for c in range(n_chunks):
        c_lo, c_hi = bounds[c], bounds[c + 1]
        if c_hi <= c_lo:
            continue
        qv_c = {m: qvec[m][c_lo:c_hi].contiguous() for m in names} if n_chunks > 1 else qvec
        q2c, (loc_s, loc_i) = _local_scores_topk(index, qv_c, k, ops)
        if q2c is not None:
            q2c_parts.append(q2c)
        if pending is not None:
            finish(pending)             # K7 / K9 of the previous chunk, behind this chunk's K6 in the stream
        if cuda:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                own = ex.topk_by_owner(loc_s, loc_i, k, q2c_alpha, ops)
                ev = torch.cuda.Event()
                ev.record(side)
            for t in (loc_s, loc_i):
                t.record_stream(side)
            for t in own:
                t.record_stream(main)
            pending = (c_lo, qv_c, own, ev)
        else:
            own = ex.topk_by_owner(loc_s, loc_i, k, q2c_alpha, ops)
            _mark("exchange+merge_topk")
            pending = (c_lo, qv_c, own, None)
    finish(pending)
    cat = lambda j: torch.cat([p[j] for p in parts]) if len(parts) > 1 else parts[0][j]       # noqa: E731
    q2c_all = None if not q2c_parts else (torch.cat(q2c_parts) if len(q2c_parts) > 1 else q2c_parts[0])
    return cat(0), cat(1), cat(2), cat(3), (torch.cat(owned) if len(owned) > 1 else owned[0]), q2c_all


def video_owner_local_moments(model, index, qvec, top_w, top_gid, n_out, min_pred_l, max_pred_l, ops=hip_ops):
    """Phase 2 of the video-owner scheme on one rank: K7 + K9 for the global top-k videos THIS shard holds (slots of videos
    owned elsewhere are skipped: pair -1, weight 0), flat indices in the GLOBAL slot order -> (scores, flat) (Nq, n_out),
    empty slots score 0 / flat -1.  The per-rank lists are disjoint; their top-n merge is the single-GPU list."""
    lo, hi = index.video_offset, index.video_offset + index.n_videos
    own = (top_gid >= lo) & (top_gid < hi)
    pair_local = torch.where(own, top_gid - lo, torch.full_like(top_gid, -1)).contiguous()
    vl = inf.ragged_lengths(index, ops)
    rk = dict(pair_vid=pair_local, vid_len=vl) if vl is not None else {}
    st, ed = inf.stage_span_probs(model, index, qvec, pair_local, ops, zero_skipped=False, vid_len=vl)   # K9 skips w == 0 pairs
    _mark("convse_k7")
    w_local = torch.where(own, top_w, torch.zeros_like(top_w)).contiguous()   # w == 0 marks slots owned elsewhere:
    loc_fs, loc_fi = ops.moment_topk(st, ed, w_local, index.l_ref, min_pred_l, max_pred_l, n_out, **rk)     # skipped
    _mark("moment_k9")
    return loc_fs, loc_fi


def moment_merge_payload(loc_fi):
    """Empty slots carry score 0 / flat -1: give them the largest payload so that real moments win ties at score 0 in the
    merge (xml_topk_rows orders by score desc, payload asc)."""
    return torch.where(loc_fi >= 0, loc_fi, torch.full_like(loc_fi, 2 ** 31 - 1)).contiguous()


def sharded_vcmr_search(model, index, query_feat, query_mask, max_vcmr_video=100, max_before_nms=200,
                        q2c_alpha=20.0, min_pred_l=2, max_pred_l=16, group=None, ops=hip_ops, qvec=None,
                        gather_results=True, owner_rerank=None, n_chunks=1, exchange=None):
    """Exact corpus-sharded counterpart of inference.vcmr_search.  `index` is this rank's CorpusIndex
    (index.video_offset = global id of its first video, index.n_total = corpus size).
    Returns top_scores/top_indices (., k) with GLOBAL video ids and flat_scores/flat_indices (., n): all Nq rows on
    every rank with gather_results=True, else the rows this rank owns -- `query_index` (global query ids, ascending) and,
    when they are one contiguous slice (n_chunks == 1), `query_range` (sharded rerank: top_* are known everywhere and
    always complete).
    owner_rerank (default: whenever replicate_rerank_features(index) was called): phase 2 runs on the query's owner
    against the corpus-wide feat2 copy -- two collectives per pass instead of four.
    n_chunks > 1 (owner rerank): the pass is pipelined over query chunks (see _owner_pass); ownership is then per chunk.
    exchange: TorchExchange / RcclExchange (default: RCCL through the C ABI for GPU ranks, torch.distributed otherwise)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    trivial = _trivial(world)
    ex = exchange or (None if trivial else default_exchange(query_feat.device if query_feat is not None
                                                            else next(iter(qvec.values())).device, group))
    _mark("start")
    if qvec is None:
        qvec = encode_queries_sharded(model, query_feat, query_mask, group, ex)
    _mark("query_encode+allgather")
    nq = next(iter(qvec.values())).shape[0]
    k = min(max_vcmr_video, index.n_total)
    q_lo, q_hi, per = query_slice(nq, rank, world)
    if owner_rerank is None:
        owner_rerank = index.feat2_all is not None
    assert not owner_rerank or index.feat2_all is not None, "call replicate_rerank_features(index) first"
    if not owner_rerank and not trivial and max_before_nms > 256:
        # the video-owner rerank merges the per-rank moment lists with xml_topk_rows (k <= 256); the query-owner rerank
        # (replicate_rerank_features) and the single-GPU pass take n_out up to 1024 (xml_moment_topk)
        raise ValueError("sharded rerank merges moment lists of at most 256 entries; max_before_nms=%d needs the owner "
                         "rerank (call replicate_rerank_features(index) first)" % max_before_nms)
    check_shards(index, group)      # collective on the first pass over an index: an empty shard raises on EVERY rank
    if not trivial and owner_rerank:
        top_w, top_gid, fs, fi, owned, q2c = _owner_pass(model, index, qvec, ex, k, max_before_nms, q2c_alpha,
                                                          min_pred_l, max_pred_l, ops, n_chunks)
        res = dict(q2c_local=q2c, query_index=owned)
        if gather_results:
            # every rank ends up with all rows: all-gather the owned rows (padded to the largest owner), then scatter
            # them to their global query positions
            n_max = torch.tensor([owned.numel()], device=owned.device)
            dist.all_reduce(n_max, op=dist.ReduceOp.MAX, group=group)
            n_max = int(n_max.item())

            def gather(t, fill):
                pad = t.new_full((n_max,) + tuple(t.shape[1:]), fill)
                pad[:t.shape[0]] = t
                return ex.allgather_rows(pad)
            gq = gather(owned, -1).long()
            ok = gq >= 0

            def place(t, fill):
                g = gather(t, fill)
                out = g.new_full((nq,) + tuple(t.shape[1:]), fill)
                out[gq[ok]] = g[ok]
                return out
            top_w, top_gid, fs, fi = place(top_w, 0.0), place(top_gid, -1), place(fs, 0.0), place(fi, -1)
            res["query_index"] = torch.arange(nq, device=owned.device)
            res["query_range"] = (0, nq)
        elif n_chunks <= 1:
            res["query_range"] = (q_lo, q_hi)
        res.update(top_scores=top_w, top_indices=top_gid, flat_scores=fs, flat_indices=fi)
        return res
    # ---- phase 1: global top-k videos ------------------------------------------------------------------
    if trivial and index.exact is not None:
        q2c = None
        top_w, top_gid, _ = inf.stage_exact_topk(index, qvec, min(k, index.n_videos), q2c_alpha, ops)
        top_gid = top_gid + index.video_offset
        _mark("exact_topk")
    elif trivial:
        q2c = inf.stage_q2c(index, qvec, ops)
        _mark("q2c_k6")
        top_w, top_gid = ops.topk_rows(q2c, k, alpha=q2c_alpha)
        top_gid = top_gid + index.video_offset
        _mark("topk_k8")
    else:
        q2c, (loc_s, loc_i) = _local_scores_topk(index, qvec, k, ops)
        top_w, top_gid = ex.allgather_topk(loc_s, loc_i, k, q2c_alpha, ops)       # every rank: global top-k of all queries
        _mark("allgather+merge_topk")
    # ---- phase 2: moments of the global top-k videos this rank owns -------------------------------------
    loc_fs, loc_fi = video_owner_local_moments(model, index, qvec, top_w, top_gid, max_before_nms, min_pred_l, max_pred_l, ops)
    if trivial:
        fs, fi = loc_fs, loc_fi
    else:
        loc_fi = moment_merge_payload(loc_fi)
        if gather_results:
            fs, fi = ex.allgather_topk(loc_fs.contiguous(), loc_fi, max_before_nms, 0.0, ops)
        else:
            fs, fi = ex.topk_by_owner(loc_fs.contiguous(), loc_fi, max_before_nms, 0.0, ops)
        fi = torch.where(fs > 0, fi, torch.full_like(fi, -1))
        _mark("exchange+merge_moments")
    if trivial or gather_results:
        q_lo, q_hi = 0, nq
    return dict(top_scores=top_w, top_indices=top_gid, flat_scores=fs, flat_indices=fi, q2c_local=q2c,
                query_range=(q_lo, q_hi), query_index=torch.arange(q_lo, q_hi, device=top_w.device))
