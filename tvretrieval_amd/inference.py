"""Corpus-level retrieval drivers: host-side mirror of the hot loops of the reference's
baselines/crossmodal_moment_localization/inference.py ("xml/inference.py").

  compute_context_info     <- xml/inference.py:32-97     (HOT LOOP A: encode the corpus once)
  compute_query2ctx_info   <- xml/inference.py:252-445   (HOT LOOP B: per query batch, VCMR / SVMR / VR)
  vcmr_search              the device part of HOT LOOP B (:302-386) on resident tensors, used by bench.py

What changed by design (outputs are the same lists):
  * the corpus lives in HBM as a `CorpusIndex`: feat1 stored L2-normalised (the reference re-normalises it for
    every query batch, xml/model_xml.py:447), feat2 and masks padded to a multiple of 16 clips;
  * the (Nq, Nv, L) st/ed tensors are never built: ConvSE runs only on the top-k (and GT) videos per query;
  * the (Nq, k, L, L) product + full sort is replaced by a banded top-n kernel.
Everything on the device goes through tvretrieval_amd.ops (HIP); numpy only formats the result lists.
"""
import numpy as np
import torch

from . import ops as hip_ops


def _round_up(x, m):
    return (x + m - 1) // m * m


class CorpusIndex(object):
    """Encoded corpus resident in HBM.

    modalities : list of "video" / "sub" in model order
    feat1n[m]  : (Nv, lpad, H) compute dtype, L2-normalised rows (zero rows beyond each batch's own length); on the
                 HIP backend at lpad == 128 an ops.TiledRows (K6's slice-major tile layout) -- feat1n_rows(m) gives
                 the row-major tensor back
    feat2[m]   : (Nv, lpad, H) compute dtype
    mask[m]    : (Nv, lpad) float32
    l_ref      : the reference's context length = global max clip count (xml/inference.py:71-87)
    video_offset : global index of local video 0 (corpus shards, tvretrieval_amd.dist)
    """

    def __init__(self, modalities, feat1n, feat2, mask, l_ref, video_offset=0, n_total=None):
        self.modalities = list(modalities)
        self.feat1n, self.feat2, self.mask = feat1n, feat2, mask
        self.l_ref = int(l_ref)
        self.lpad = int(feat2[self.modalities[0]].shape[1])
        self.n_videos = int(feat2[self.modalities[0]].shape[0])
        self.video_offset = int(video_offset)
        self.n_total = int(n_total if n_total is not None else self.n_videos)
        self.feat2_all = self.mask_all = None      # corpus-wide copies of feat2 / mask (dist.replicate_rerank_features)
        self.exact = None                          # ExactFilter: feat1n is the bf16 FILTER image of an f32 index (exact-rank mode)
        # ragged corpora: valid clips per video (1 + index of the last unmasked clip over the modalities) -- K7 neither
        # fetches the clip rows nor writes the (exactly zero) probabilities beyond it, K9 does not read them
        self.vlen = self.vlen_all = None
        self.ragged = False

    def set_valid_lengths(self):
        """Per-video valid length from the masks (one small device pass + one host read at build time)."""
        pos = None
        for m in self.modalities:
            mk = self.mask[m]
            if not isinstance(mk, torch.Tensor):
                return self
            last = ((mk != 0).to(torch.int32) * torch.arange(1, mk.shape[1] + 1, device=mk.device, dtype=torch.int32)).amax(1)
            pos = last if pos is None else torch.maximum(pos, last)
        # The maximum over modalities is exact, also when one modality of a video has no valid clip: the reference averages
        # the masked LOGITS of the two streams, (video + sub) / 2 with -1e10 at masked positions, and takes the softmax
        # afterwards (xml/model_xml.py:436-453, xml/inference.py:365-370) -- a position masked in one stream sits at -5e9, in
        # both at -1e10, and exp() of either against a valid position is exactly 0
        # (tests/test_gpu_model.py::test_video_with_one_empty_modality_matches_the_reference_lists).
        pos = torch.where(pos == 0, torch.full_like(pos, self.l_ref), pos)     # no valid clip at all: the masked softmax is
        self.vlen = pos.clamp_max(self.l_ref).to(torch.int32).contiguous()     # uniform, not zero -> nothing is skipped
        self.ragged = bool((self.vlen < self.l_ref).any().item())
        return self

    def feat1n_rows(self, m):
        t = self.feat1n[m]
        return t.to_rows() if hasattr(t, "to_rows") else t

    @property
    def device(self):
        return self.feat2[self.modalities[0]].device

    def hbm_bytes(self):
        tot = 0
        for d in (self.feat1n, self.feat2, self.mask, self.feat2_all or {}, self.mask_all or {},
                  self.exact.feat1n_f32 if self.exact is not None else {}):
            for t in d.values():
                tot += t.numel() * t.element_size()
                if hasattr(t, "inv"):
                    tot += t.inv.numel() * 4
        return tot


class ExactFilter(object):
    """Exact-rank mode of a CorpusIndex (include/xmlhip.h "Exact-rank mode"): the model runs in f32, `index.feat1n` holds the
    similarity operand ROUNDED ONCE to bf16 (K6 is a filter), and this object holds what turns the filter's candidates into
    the f32 path's lists:
      feat1n_f32[m]  (Nv, lpad, H) f32 row-major, L2-normalised -- re-scoring operand, and the f32 K6 operand of the fallback
      e_c[m]         largest bf16 rounding-error norm || c - c_b ||_2 over the corpus rows of modality m
      n_candidates   M: candidates per query proposed by the bf16 pass (K8 emits at most 256)"""

    def __init__(self, feat1n_f32, e_c, n_candidates=256, mode="f32"):
        self.feat1n_f32, self.e_c, self.n_candidates, self.mode = feat1n_f32, e_c, int(n_candidates), mode
        # mode "f32"  : bf16 filter image, f32 rows re-scored with the exact-f32 MFMA (round 3)
        # mode "f16s" : f16 filter image (the hi planes), feat1n_f32 holds ops.SplitRows (hi + lo halves at the fixed unit
        #               scale) re-scored as split-f16 products, index.feat2 are SplitRows too: every f32-grade stage on the
        #               16-bit MFMA pipe (include/xmlhip.h "Exact-rank mode on the 16-bit pipe")
        self.tier2_rows = None        # capacity of the on-device second tier (None: sized from the batch); tests shrink it
        self.tier2_cap = 1024         # candidates per second-tier query


# Filter operand of the split-f16 exact mode.  The K6 kernel is power-limited and every mantissa bit of its operands costs
# ~0.85 ms of a 65 ms launch (tools/bench_k6_dtype.py: bf16 64.8, f16 69.3 ms on one box); an f16 filter needs 128
# candidates per query, a bf16 filter 256 -- and the 128-row re-score chunks make the second 128 candidates cheap.
EXACT_F16S_FILTER = "bf16"          # "bf16" (256 candidates) or "f16" (128 candidates)


def _exact_filter_operands(f1_raw, mask, plan, ops, mode="f32", filter_dtype=None):
    """raw f32 feat1 (Nv, lpad, H) -> (filter image for K6, re-score operand, largest rounding-error norm).
    mode "f32": bf16 tiles + f32 normalised rows; mode "f16s": bf16 or f16 tiles + SplitRows."""
    if f1_raw.dtype != torch.float32:
        raise ValueError("exact-rank mode needs f32 activations (XML(cfg, compute_dtype=torch.float32 or ops.F16S)); got %s"
                         % f1_raw.dtype)
    fn = ops.l2norm_rows(f1_raw)
    if mode == "f16s" and (filter_dtype or EXACT_F16S_FILTER) == "f16":
        sr, hi, err = ops.split_f16_rows(fn, ops.F16_UNIT_LOG2, want_hi=True, want_err=True)
        e_c = float(err.max()) if err.numel() else 0.0
        return ops.pack_q2c_corpus(hi, mask, plan, normalize=False), sr, e_c
    if mode == "f16s":
        sr = ops.split_f16_rows(fn, ops.F16_UNIT_LOG2)
        fb, err = ops.round_bf16_rows_err(fn)
        e_c = float(err.max()) if err.numel() else 0.0
        return ops.pack_q2c_corpus(fb, mask, plan, normalize=False), sr, e_c
    fb, err = ops.round_bf16_rows_err(fn)
    e_c = float(err.max()) if err.numel() else 0.0
    return ops.pack_q2c_corpus(fb, mask, plan, normalize=False), fn, e_c


def exact_mode_of(model, ops=hip_ops):
    """which exact-rank pipeline an index built from this model gets"""
    return "f16s" if getattr(model, "compute_dtype", None) is getattr(ops, "F16S", object()) else "f32"


def index_lpad(l_ref, model, ops=hip_ops):
    """Padded clip count of the index tensors: l_ref rounded up to 16 -- or 128 when that lets K6 take its persistent tiled
    kernel (which is built for 128-column video groups): the reference's as-trained shape, max_ctx_l = 100
    (xml/config.py:86-88), pads 100 -> 128 (the length-bucketed layout then packs the videos of <= 64 / <= 32 clips 4 / 8 to
    a tile, so the padding rows of SHORT videos cost no MFMA work) instead of 112 on the slow per-modality kernels."""
    lp = _round_up(int(l_ref), 16)
    dt = getattr(model, "compute_dtype", torch.float32)
    if dt is getattr(ops, "F16S", object()):
        dt = torch.float16            # (its K6 operand is the f16 hi plane)
    if 64 < lp < 128 and hasattr(ops, "q2c_tiled_ok") and model is not None and \
            ops.q2c_tiled_ok(128, model.config.hidden_size, dt):
        return 128
    return lp


def pad_batch(seqs, device=None, dtype=torch.float32):
    """pad_sequences_1d (utils/tensor_utils.py:5-53) for a list of (L_i, D) arrays -> (N, Lmax, D), (N, Lmax).
    One concatenate + one masked assignment instead of a Python loop of per-sequence tensor copies (50 ms -> 2 ms for a
    batch of 50 queries: the loop, not the device, was what eval_epoch waited for)."""
    n = len(seqs)
    arrs = [s.detach().cpu().numpy() if isinstance(s, torch.Tensor) else np.asarray(s) for s in seqs]
    lens = np.fromiter((len(a) for a in arrs), dtype=np.int64, count=n)
    lmax = int(lens.max())
    valid = np.arange(lmax)[None, :] < lens[:, None]
    np_dtype = torch.empty(0, dtype=dtype).numpy().dtype
    out = np.zeros((n, lmax) + tuple(arrs[0].shape[1:]), dtype=np_dtype)
    out[valid] = np.concatenate(arrs, axis=0)
    out, mask = torch.from_numpy(out), torch.from_numpy(valid.astype(np.float32))
    if device is not None:
        out, mask = out.to(device, non_blocking=True), mask.to(device, non_blocking=True)
    return out, mask


class IndexStorage(object):
    """Device memory of a corpus index, allocated (and zero-filled: touched) BEFORE the encode -- a resident engine maps its
    index once; hipMalloc + first touch of tens of GB inside the encode loop is what a fresh process pays, not the encoder
    (bench.py times the two separately).  f1 / f2 / mk: per-modality (n_videos, lpad, H) compute dtype x2 and (n_videos,
    lpad) f32; tiles: per-modality flat buffer of the K6 tile image (full-length corpora: the size is known up front)."""

    def __init__(self, model, n_videos, l_ref, ops=hip_ops, device=None, tiles=True):
        # tiles=False: an exact-rank index builds its filter image itself (from the normalised f32 rows)
        mods = [n for n, u in (("video", model.use_video), ("sub", model.use_sub)) if u]
        dev = device if device is not None else next(model.parameters()).device
        dt, h = getattr(model, "act_dtype", model.compute_dtype), model.config.hidden_size
        self.n_videos, self.l_ref, self.lpad = int(n_videos), int(l_ref), index_lpad(l_ref, model, ops)
        self.f1 = {m: torch.zeros((self.n_videos, self.lpad, h), dtype=dt, device=dev) for m in mods}
        self.f2 = {m: torch.zeros((self.n_videos, self.lpad, h), dtype=dt, device=dev) for m in mods}
        self.mk = {m: torch.zeros((self.n_videos, self.lpad), dtype=torch.float32, device=dev) for m in mods}
        self.tiles = {}
        if tiles and hasattr(ops, "q2c_tiled_numel") and self.lpad == 128 and dt in (torch.float32, torch.bfloat16):
            n = ops.q2c_tiled_numel(self.n_videos * self.lpad, h, dt)
            if n:
                self.tiles = {m: torch.zeros(n, dtype=dt, device=dev) for m in mods}

    def nbytes(self):
        return sum(t.numel() * t.element_size() for d in (self.f1, self.f2, self.mk, self.tiles) for t in d.values())


def build_corpus_index(model, context_batches, ops=hip_ops, keep_raw=False, video_offset=0, n_total=None,
                        l_ref=None, n_videos=None, exact_filter=False, storage=None):
    """Encode context batches and assemble the resident index.

    context_batches: iterable of (video_feat, video_mask, sub_feat, sub_mask) device tensors (unused modality:
    None).  Each batch is encoded at its own padded length and zero-filled beyond it when concatenated, exactly
    like cat_tensor (xml/inference.py:71-87), so rows >= a batch's max length are 0 and rows between a video's
    length and its batch max hold the encoder's outputs at padded positions (both observable through the 5-tap
    ConvSE).
    exact_filter=True (f32 model): exact-rank mode -- feat1n becomes the bf16 filter image, index.exact the f32 operands
    (ExactFilter); vcmr_search then returns the f32 path's lists at close to the bf16 path's speed."""
    mods = [n for n, u in (("video", model.use_video), ("sub", model.use_sub)) if u]
    if storage is not None:
        assert l_ref is None or int(l_ref) == storage.l_ref
        assert n_videos is None or int(n_videos) == storage.n_videos
        l_ref, n_videos = storage.l_ref, storage.n_videos
    if n_videos is not None and l_ref is not None:
        return _build_corpus_index_prealloc(model, context_batches, ops, keep_raw, video_offset, n_total, int(l_ref),
                                            int(n_videos), mods, exact_filter, storage)
    parts = {m: dict(f1=[], f2=[], mk=[]) for m in mods}
    for video_feat, video_mask, sub_feat, sub_mask in context_batches:
        v1, v2, s1, s2 = model.encode_context(video_feat, video_mask, sub_feat, sub_mask)
        if "video" in parts:
            parts["video"]["f1"].append(v1), parts["video"]["f2"].append(v2), parts["video"]["mk"].append(video_mask.float())
        if "sub" in parts:
            parts["sub"]["f1"].append(s1), parts["sub"]["f2"].append(s2), parts["sub"]["mk"].append(sub_mask.float())
    batch_max = max(t.shape[1] for t in parts[mods[0]]["f2"])
    l_ref = batch_max if l_ref is None else int(l_ref)
    assert l_ref >= batch_max
    lpad = index_lpad(l_ref, model, ops)

    def cat(tensors):
        n = sum(t.shape[0] for t in tensors)
        out = tensors[0].new_zeros((n, lpad) + tuple(tensors[0].shape[2:]))
        r = 0
        for t in tensors:
            out[r:r + t.shape[0], :t.shape[1]] = t
            r += t.shape[0]
        return out

    feat1n, feat2, mask, raw, ex_f32, ex_ec = {}, {}, {}, {}, {}, {}
    for m in mods:
        mask[m] = cat(parts[m]["mk"])
    # ragged corpora: one length-bucketed layout shared by the modalities (2 / 4 / 8 videos per K6 tile)
    plan = ops.q2c_pack_plan([mask[m] for m in mods]) if (hasattr(ops, "q2c_pack_plan") and lpad == 128) else None
    for m in mods:
        f1 = cat(parts[m]["f1"])
        feat2[m] = cat(parts[m]["f2"])
        if exact_filter:
            feat1n[m], ex_f32[m], ex_ec[m] = _exact_filter_operands(f1, mask[m], plan, ops, exact_mode_of(model, ops))
            if exact_mode_of(model, ops) == "f16s":
                feat2[m] = ops.split_f16_rows(feat2[m])
        elif hasattr(ops, "pack_q2c_corpus"):    # HIP backend: normalised + slice-major tiles for the persistent K6 kernel
            feat1n[m] = ops.pack_q2c_corpus(f1, mask[m], plan, normalize=True)
        else:
            feat1n[m] = ops.l2norm_rows(f1)
        if keep_raw:
            raw[m] = f1
    idx = CorpusIndex(mods, feat1n, feat2, mask, l_ref, video_offset, n_total)
    if getattr(ops, "RAGGED_ROWS", False):
        idx.set_valid_lengths()
    idx.raw_feat1 = raw
    if exact_filter:
        idx.exact = _make_exact_filter(ex_f32, ex_ec, exact_mode_of(model, ops))
    return idx


def _make_exact_filter(ex_f32, ex_ec, mode):
    # f16 filter: rounding errors 8x smaller than bf16's -> the certificate holds with half the candidates
    return ExactFilter(ex_f32, ex_ec, n_candidates=128 if (mode == "f16s" and EXACT_F16S_FILTER == "f16") else 256, mode=mode)


def _build_corpus_index_prealloc(model, context_batches, ops, keep_raw, video_offset, n_total, l_ref, n_videos, mods,
                                 exact_filter=False, storage=None):
    """build_corpus_index when the number of videos and the corpus-wide length are known up front (a resident engine knows
    its corpus): the three index tensors per modality are allocated once and every encoded batch is written into its
    rows -- no growing list of per-batch outputs, no concatenation pass, and the per-batch activations are recycled by the
    allocator instead of each batch mapping fresh memory.  Same contents as the list + cat path (zero rows beyond a
    batch's own padded length)."""
    lpad = index_lpad(l_ref, model, ops)
    f1, f2, mk = {}, {}, {}
    if storage is not None:       # (zero-filled by IndexStorage; used for ONE build)
        f1, f2, mk = dict(storage.f1), dict(storage.f2), dict(storage.mk)
    r = 0
    for video_feat, video_mask, sub_feat, sub_mask in context_batches:
        # batches of full padded length are encoded STRAIGHT into their rows of the index tensors (the last layer of each
        # branch writes there): no per-batch copies.  The first batch, shorter batches and non-HIP backends go through copies.
        outs = [None, None, None, None]
        direct = bool(f1) and getattr(ops, "ENCODE_INTO_INDEX", False)
        if direct:
            for i, (m, feat) in enumerate((("video", video_feat), ("sub", sub_feat))):
                b = feat.shape[0] if feat is not None else 0
                if m in mods and feat is not None and feat.shape[1] == lpad and r + b <= n_videos:
                    outs[2 * i], outs[2 * i + 1] = f1[m][r:r + b], f2[m][r:r + b]
                elif m in mods:
                    direct = False
        if direct:
            v1, v2, s1, s2 = model.encode_context(video_feat, video_mask, sub_feat, sub_mask, outs=tuple(outs))
        else:
            v1, v2, s1, s2 = model.encode_context(video_feat, video_mask, sub_feat, sub_mask)
        for m, a1, a2, am in (("video", v1, v2, video_mask), ("sub", s1, s2, sub_mask)):
            if m not in mods:
                continue
            if m not in f1:
                f1[m] = a1.new_zeros((n_videos, lpad, a1.shape[2]))
                f2[m] = a2.new_zeros((n_videos, lpad, a2.shape[2]))
                mk[m] = torch.zeros((n_videos, lpad), dtype=torch.float32, device=a1.device)
            b, lb = a1.shape[0], a1.shape[1]
            assert r + b <= n_videos and lb <= l_ref
            if not direct:
                f1[m][r:r + b, :lb] = a1
                f2[m][r:r + b, :lb] = a2
            mk[m][r:r + b, :lb] = am.float()
        r += v1.shape[0] if v1 is not None else s1.shape[0]
    assert r == n_videos, "n_videos=%d but the batches held %d" % (n_videos, r)
    plan = ops.q2c_pack_plan([mk[m] for m in mods]) if (hasattr(ops, "q2c_pack_plan") and lpad == 128) else None
    feat1n, raw, ex_f32, ex_ec = {}, {}, {}, {}
    for m in mods:
        if exact_filter:
            feat1n[m], ex_f32[m], ex_ec[m] = _exact_filter_operands(f1[m], mk[m], plan, ops, exact_mode_of(model, ops))
            if exact_mode_of(model, ops) == "f16s":
                f2[m] = ops.split_f16_rows(f2[m])
        else:
            tile_buf = storage.tiles.get(m) if (storage is not None and plan is None) else None
            kw = dict(out=tile_buf) if tile_buf is not None else {}
            feat1n[m] = ops.pack_q2c_corpus(f1[m], mk[m], plan, normalize=True, **kw) if hasattr(ops, "pack_q2c_corpus") \
                else ops.l2norm_rows(f1[m])
        if keep_raw:
            raw[m] = f1[m]
    idx = CorpusIndex(mods, feat1n, f2, mk, l_ref, video_offset, n_total)
    if getattr(ops, "RAGGED_ROWS", False):
        idx.set_valid_lengths()
    idx.raw_feat1 = raw
    if exact_filter:
        idx.exact = _make_exact_filter(ex_f32, ex_ec, exact_mode_of(model, ops))
    return idx


def compute_context_info(model, eval_dataset, opt, ops=hip_ops):
    """Mirror of compute_context_info (xml/inference.py:32-97): same dict keys, plus "index" (CorpusIndex).
    `eval_dataset` follows the reference's dataset contract (set_data_mode("context"), items with "meta" and
    "model_inputs" {video_feat, sub_feat}); batches of opt.eval_context_bsz in dataset order."""
    eval_dataset.set_data_mode("context")
    device = opt.device
    metas = []

    def batches():
        n = len(eval_dataset)
        for b in range(0, n, opt.eval_context_bsz):
            items = [eval_dataset[i] for i in range(b, min(n, b + opt.eval_context_bsz))]
            metas.extend(e["meta"] for e in items)
            vf = vm = sf = sm = None
            if model.use_video:
                vf, vm = pad_batch([e["model_inputs"]["video_feat"] for e in items], device)
            if model.use_sub:
                sf, sm = pad_batch([e["model_inputs"]["sub_feat"] for e in items], device)
            yield vf, vm, sf, sm

    index = build_corpus_index(model, batches(), ops=ops, keep_raw=True)
    lr = index.l_ref
    g = lambda d, m: d[m][:, :lr] if m in d else None
    return dict(video_metas=metas,
                video_feat1=g(index.raw_feat1, "video"), video_feat2=g(index.feat2, "video"),
                video_mask=g(index.mask, "video"),
                sub_feat1=g(index.raw_feat1, "sub"), sub_feat2=g(index.feat2, "sub"), sub_mask=g(index.mask, "sub"),
                index=index)


# ---------------------------------------------------------------------------------------------------------
# device stages of HOT LOOP B
# ---------------------------------------------------------------------------------------------------------
def stage_query_vectors(model, query_feat, query_mask, n_valid_tokens=None):
    """encode_query -> per-modality modular query vectors, in index.modalities order.
    n_valid_tokens: see XML.encode_query (host-known token count: no read-back in the packed encoder)."""
    vq, sq = model.encode_query(query_feat, query_mask, **(dict(n_valid_tokens=n_valid_tokens) if n_valid_tokens is not None else {}))
    out = {}
    if model.use_video:
        out["video"] = vq
    if model.use_sub:
        out["sub"] = sq
    return out


import os as _os
# Measured in round 4 (profiles/r04_notes.md) and NOT the default: handing K9 its selection threshold from K7 takes K9 from
# 0.65 to 0.45-0.53 ms at the TVR shape, but the extra epilogue work costs K7 as much or more (+0.2 ms with one maximum per
# group of 16 rows, +0.45 ms with the 8 largest rows of a pair), and at the as-trained shape the weaker bound makes K9 slower.
RAGGED_ROWS = _os.environ.get("XML_RAGGED_ROWS", "1") == "1"     # K7 / K9 skip the zero tails of short videos (A/B: 0)
K7_SUMMARIES = _os.environ.get("XML_K7_SUMMARIES", "0") == "1"   # vcmr_search: K7 emits per-pair candidate summaries for K9 (False: K9 makes its own first pass; A/B)
K6_TIMER = None   # bench.py: callable returning (start, end) torch.cuda.Event pair recorded around each K6 launch


def _k6(index, qn, ops, normalize_q=False):
    """One fused K6 launch over the local corpus (both modalities), bracketed by the bench's HIP events."""
    mods = index.modalities
    if normalize_q and not getattr(ops, "Q2C_NORMALIZE_Q", False):
        qn, normalize_q = [ops.l2norm_rows(q) for q in qn], False
    kw = dict(normalize_q=True) if normalize_q else {}
    ev = K6_TIMER() if K6_TIMER is not None else None
    if ev:
        ev[0].record()
    q2c = ops.q2c_scores_fused(qn, [index.feat1n[m] for m in mods], [index.mask[m] for m in mods], **kw)
    if ev:
        ev[1].record()
    return q2c


def stage_q2c(index, qvec, ops=hip_ops):
    """K6 over the local corpus: (Nq, Nv_local) f32 = mean over modalities of max-over-clips cosine (one launch)."""
    if index.exact is not None:
        raise ValueError("this index is an exact-rank FILTER image (bf16 operands of an f32 model): use stage_exact_topk")
    return _k6(index, [qvec[m].contiguous() for m in index.modalities], ops, normalize_q=True)


EXACT_TIER2_CAP = 4096        # second-tier candidates per failing query beyond which the f32 K6 row is computed instead


def exact_slack(hidden):
    """f32 accumulation bound of the two dot products a certificate compares: the filter's (exact bf16 products, f32
    accumulate) and the f32 path's own (fma chain): each |fl(sum) - sum| <= n u |x| |y|, u = 2^-24."""
    return 2.0 * (hidden + 1) * 2.0 ** -24


def stage_exact_topk(index, qvec, k, alpha, ops=hip_ops, defer_check=False):
    """Exact-rank replacement of K6 + K8: dispatches on index.exact.mode (round-3 f32 re-score / split-f16 pipeline)."""
    if index.exact.mode == "f16s":
        return stage_exact_topk_f16s(index, qvec, k, alpha, ops, defer_check)
    return stage_exact_topk_f32(index, qvec, k, alpha, ops)


def stage_exact_topk_f16s(index, qvec, k, alpha, ops=hip_ops, defer_check=False):
    """The exact-rank chain on the 16-bit MFMA pipe, with no host read-back on the common path (capturable):
      1. q normalised (f32) and split: hi plane (f16) -> K6 FILTER over the corpus's hi planes; rounding-error norms e_q;
      2. K8: the M best filter scores per query (M = 128: the f16 filter's error bound is 8x tighter than bf16's);
      3. xml_q2c_rescore on split-f16 rows (hi.hi + lo.hi + hi.lo): f32-grade scores of the candidates; K8 -> top-k;
      4. certificate  b_M + eps_q < T_k  per query (eps from the rounding-error norms, as in the f32 mode);
      5. ON-DEVICE second tier, fixed capacity: the failing queries are brought to the front by a stable sort of the fail
         flags, the first R slots get every video whose filter score reaches T_k - eps (<= C per query), re-scored and
         re-selected, and written back through a select on the fail flag -- slots of passing queries change nothing;
      6. only if more than R queries fail or a query has more than C such videos (scores closer together than any 16-bit
         filter resolves) the overflow flag is raised: eager callers (defer_check=False) read it -- one 4-byte read-back
         after everything is enqueued -- and re-score those queries against the whole corpus; graphed callers get the flag.
    Returns (top_w = exp(alpha s) (Nq, k) f32, top_i (Nq, k) int32, info dict)."""
    ex = index.exact
    mods = index.modalities
    if k > min(ex.n_candidates, index.n_videos):
        raise ValueError("exact-rank mode: top-%d videos asked of %d candidates per query (ExactFilter.n_candidates; K8 "
                         "proposes at most 256) -- lower max_vcmr_video or raise n_candidates" % (k, ex.n_candidates))
    masks = [index.mask[m] for m in mods]
    q_sr, q_hi, eq = [], [], []
    for m in mods:
        q = qvec[m].contiguous()
        if q.dtype != torch.float32:
            raise ValueError("exact-rank mode needs f32 query vectors (an f32 / ops.F16S model)")
        qn = ops.l2norm_rows(q)
        if index.feat1n[m].dtype == torch.float16:              # f16 filter: the hi plane of the split
            sr, hi, e = ops.split_f16_rows(qn, ops.F16_UNIT_LOG2, want_hi=True, want_err=True)
        else:                                                   # bf16 filter
            sr = ops.split_f16_rows(qn, ops.F16_UNIT_LOG2)
            hi, e = ops.round_bf16_rows_err(qn)
        q_sr.append(sr), q_hi.append(hi), eq.append(e)
    nq, hidden = q_hi[0].shape
    filt = _k6(index, q_hi, ops)
    m_c = min(ex.n_candidates, index.n_videos)
    cand_s, cand_i = ops.topk_rows(filt, m_c, alpha=0.0)
    rows_c = [ex.feat1n_f32[m] for m in mods]                    # SplitRows (Nv, lpad, H)
    cand_r = ops.q2c_rescore(q_sr, rows_c, masks, cand_i)
    top_w, top_i = ops.topk_rows(cand_r, k, alpha=0.0, idx_in=cand_i)
    outside = index.n_videos > m_c
    # three product sums per re-scored value, one per filter value: f32 accumulation bound of each + the second-order terms
    # of the two error norms (e_q e_c) the first-order formula of xml_exact_certificate leaves out
    slack = 2.0 * exact_slack(hidden) + 4.0 * max(ex.e_c[m] for m in mods) ** 2 + 2.0 ** -20
    fail, eps, thr_all, n_fail = ops.exact_certificate(cand_s, top_w, eq, [ex.e_c[m] for m in mods], slack, alpha, outside)
    info = dict(fail=fail, eps=eps, q2c_filter=filt, cand_indices=cand_i, cand_filter=cand_s, cand_scores=cand_r,
                n_candidates=m_c, n_fail_dev=n_fail, n_full_rows=0, overflow_dev=None)
    if outside:
        r_cap = min(nq, ex.tier2_rows if ex.tier2_rows is not None else max(32, nq // 64))
        c_cap = min(int(ex.tier2_cap), index.n_videos)
        if c_cap < k:
            raise ValueError("exact-rank mode: tier2_cap %d < k %d" % (c_cap, k))
        order = torch.argsort(fail, descending=True, stable=True)[:r_cap]          # failing queries first; a permutation
        is_fail = fail.index_select(0, order) != 0
        inf_ = torch.full((), float("inf"), device=fail.device)
        thr = torch.where(is_fail, thr_all.index_select(0, order), inf_).contiguous()      # passing slots select nothing
        cand2, cnt2 = ops.select_ge_rows(filt.index_select(0, order).contiguous(), thr, c_cap)
        q_sub = [sr[order] for sr in q_sr]
        q_sub = [ops.SplitRows(sr.data.contiguous(), sr.inv.contiguous()) for sr in q_sub]
        full = ops.q2c_rescore(q_sub, rows_c, masks, cand2)
        # slots of PASSING queries selected nothing: their rows are c_cap times -inf, and the top-k of an all-tied row with
        # payloads takes K8's one-block-minimum-per-element path (0.46 ms for 156 such rows -- more than the second tier's
        # re-score).  Their results are discarded below; a ramp of distinct values keeps them on the fast path.
        ramp = -torch.arange(c_cap, dtype=torch.float32, device=full.device)
        full = torch.where(is_fail[:, None], full, ramp[None, :])
        fw, fi = ops.topk_rows(full, k, alpha=alpha, idx_in=cand2)
        top_w.index_copy_(0, order, torch.where(is_fail[:, None], fw, top_w.index_select(0, order)))
        top_i.index_copy_(0, order, torch.where(is_fail[:, None], fi, top_i.index_select(0, order)))
        overflow = ((cnt2 > c_cap) & is_fail).any() | (n_fail[0] > r_cap)
        info["overflow_dev"] = overflow
        if not defer_check:
            nf = int(n_fail.item())
            info["n_fail"] = nf
            if bool(overflow.item()):        # third tier: the overflowing queries against the whole corpus
                bad = torch.nonzero(fail, as_tuple=False).reshape(-1)
                if nf <= r_cap:              # only the second-tier slots whose candidate list overflowed
                    bad = order[:nf][(cnt2[:nf] > c_cap)]
                every = torch.arange(index.n_videos, dtype=torch.int32, device=fail.device)
                for b0 in range(0, bad.numel(), 16):
                    rows = bad[b0:b0 + 16]
                    qs = [ops.SplitRows(sr.data.index_select(0, rows).contiguous(), sr.inv.index_select(0, rows).contiguous())
                          for sr in q_sr]
                    allv = ops.q2c_rescore(qs, rows_c, masks, every.repeat(rows.numel(), 1).contiguous())
                    fw, fi = ops.topk_rows(allv, k, alpha=alpha)
                    top_w.index_copy_(0, rows, fw)
                    top_i.index_copy_(0, rows, fi)
                info["n_full_rows"] = int(bad.numel())
    elif not defer_check:
        info["n_fail"] = 0
    return top_w, top_i, info


def stage_exact_topk_f32(index, qvec, k, alpha, ops=hip_ops):
    """Exact-rank replacement of K6 + K8 (index.exact is set, f32 query vectors): the f32 path's top-k videos per query.
      1. q_b = rne_bf16(normalize(q)) with its rounding-error norm e_q; bf16 K6 over the filter image (the timed K6);
      2. K8 proposes the M best filter scores per query (raw values + ids);
      3. xml_q2c_rescore: those (q, v) pairs against the f32 operands -> f32 scores of the candidates;
      4. K8 again on the (Nq, M) re-scored values with the ids as payload: top-k by (score desc, id asc);
      5. per-query certificate  b_M + eps_q < T_k ; the few queries that fail get a second tier -- every video whose filter
         score reaches T_k - eps_q (nothing below that line can enter the top-k) is re-scored too -- or, when the scores are
         too close together for that (> EXACT_TIER2_CAP such videos), a full f32 K6 row (the f32 path itself).
    Returns (top_w = exp(alpha s) (Nq, k) f32, top_i (Nq, k) int32, info dict)."""
    ex = index.exact
    mods = index.modalities
    if k > min(ex.n_candidates, index.n_videos):
        raise ValueError("exact-rank mode: top-%d videos asked of %d candidates per query (ExactFilter.n_candidates; K8 "
                         "proposes at most 256) -- lower max_vcmr_video or raise n_candidates" % (k, ex.n_candidates))
    masks = [index.mask[m] for m in mods]
    qn = [ops.l2norm_rows(qvec[m].contiguous()) for m in mods]
    if qn[0].dtype != torch.float32:
        raise ValueError("exact-rank mode needs f32 query vectors (an f32 model)")
    qb, eq = [], []
    for q in qn:
        b, e = ops.round_bf16_rows_err(q)
        qb.append(b), eq.append(e)
    filt = _k6(index, qb, ops)
    m_c = min(ex.n_candidates, index.n_videos)
    cand_s, cand_i = ops.topk_rows(filt, m_c, alpha=0.0)
    f32rows = [ex.feat1n_f32[m] for m in mods]
    cand_r = ops.q2c_rescore(qn, f32rows, masks, cand_i)
    top_w, top_i = ops.topk_rows(cand_r, k, alpha=0.0, idx_in=cand_i)
    fail, eps, thr_all, n_fail = ops.exact_certificate(cand_s, top_w, eq, [ex.e_c[m] for m in mods],
                                                       exact_slack(qn[0].shape[1]), alpha, index.n_videos > m_c)
    nf = int(n_fail.item())        # (host sync: the launch shapes below depend on it)
    n_full = 0
    if nf:
        rows = torch.nonzero(fail, as_tuple=False).reshape(-1)
        qsub = [q.index_select(0, rows).contiguous() for q in qn]
        # Second tier: T_k (the k-th re-scored value) is a LOWER bound of the final k-th score, so only videos whose filter
        # score reaches T_k - eps can still enter: usually a few dozen more than M.  Re-score exactly those.
        thr = thr_all.index_select(0, rows).contiguous()
        frows = filt.index_select(0, rows).contiguous()
        cap = int(ops.select_ge_rows(frows, thr).max())
        if cap <= EXACT_TIER2_CAP:
            cand2, _ = ops.select_ge_rows(frows, thr, cap)
            full = ops.q2c_rescore(qsub, f32rows, masks, cand2)                       # (-inf where a row has fewer candidates)
            fw, fi = ops.topk_rows(full, k, alpha=alpha, idx_in=cand2)
        else:   # scores so close together that the filter cannot separate them: the f32 K6 row itself
            n_full = nf
            full = ops.q2c_scores_fused(qsub, f32rows, masks)
            fw, fi = ops.topk_rows(full, k, alpha=alpha)
        top_w.index_copy_(0, rows, fw)
        top_i.index_copy_(0, rows, fi)
    info = dict(n_fail=nf, n_full_rows=n_full, fail=fail, eps=eps, q2c_filter=filt, cand_indices=cand_i, cand_filter=cand_s,
                cand_scores=cand_r, n_candidates=m_c)
    return top_w, top_i, info


def ragged_lengths(index, ops=hip_ops, replicated=False):
    """The valid-length array K7 / K9 take for a ragged corpus (None: full rows -- every video full length, a backend
    without the entry, or RAGGED_ROWS switched off)."""
    if not (RAGGED_ROWS and getattr(ops, "RAGGED_ROWS", False) and index.ragged):
        return None
    return index.vlen_all if replicated else index.vlen


def query_linears(model, index, qvec):
    """{video,sub}_query_linear of the modular query vectors (xml/model_xml.py:459-460,507): K7's query operand."""
    return [getattr(model, m + "_query_linear")(qvec[m].contiguous()) for m in index.modalities]


SMALL_BATCH_FORK = 256      # query batches up to this size: the query linears run beside K6 + K8 on a second stream
_FORK_STREAMS = {}


def _fork_stream(device):
    key = str(torch.device(device))
    if key not in _FORK_STREAMS:
        _FORK_STREAMS[key] = torch.cuda.Stream(device=device)
    return _FORK_STREAMS[key]


def fork_query_linears(model, index, qvec, ops=hip_ops):
    """Small batches (the reference's eval_query_bsz = 50: ~20 short kernels in a chain, K6 on a handful of the 256 CUs):
    K7's query operand needs only the query vectors, so its two projections are issued on a second stream next to K6 + K8
    instead of between K8 and K7.  Returns (q_lin, event to wait for) or None; capturable (the fork and the join become
    graph edges)."""
    q0 = qvec[index.modalities[0]]
    if ops is not hip_ops or not q0.is_cuda or q0.shape[0] > SMALL_BATCH_FORK:
        return None
    main = torch.cuda.current_stream(q0.device)
    side = _fork_stream(q0.device)
    fork = torch.cuda.Event()
    fork.record(main)
    side.wait_event(fork)
    with torch.cuda.stream(side):
        q_lin = query_linears(model, index, qvec)
        done = torch.cuda.Event()
        done.record(side)
    for t in q_lin:
        t.record_stream(main)           # allocated on the side stream, consumed (and freed) on the main one
    for m in index.modalities:
        qvec[m].record_stream(side)
    return q_lin, done


def stage_span_probs(model, index, qvec, pair_vid, ops=hip_ops, zero_skipped=True, replicated=False, pair_w=None,
                     band=None, vid_len=None, q_lin=None):
    """K7 on the listed (query, local video) pairs -> softmaxed st / ed (Nq, K, lpad).
    replicated=True: pair_vid holds GLOBAL video ids into the corpus-wide copies index.feat2_all / index.mask_all
    (tvretrieval_amd.dist.replicate_rerank_features).
    vid_len (ragged_lengths(index)): the entries beyond a video's valid length are left unwritten -- hand the same array to
    ops.moment_topk(..., pair_vid=pair_vid, vid_len=vid_len)."""
    mods = index.modalities
    if q_lin is None:
        q_lin = query_linears(model, index, qvec)
    merged = bool(model.config.merge_two_stream and len(mods) == 2)
    feat2, mask = (index.feat2_all, index.mask_all) if replicated else (index.feat2, index.mask)
    if getattr(feat2[mods[0]], "dtype", None) is getattr(ops, "F16S", object()):
        q_lin = [ops.split_f16_rows(q.float().contiguous()) for q in q_lin]      # per-row scales: q' is not normalised
    # band = (min_l, max_l) [+ pair_w]: K7 also returns the per-pair candidate summaries K9 starts from (st, ed, summ)
    kw = dict(pair_w=pair_w, band=band) if (band is not None and hasattr(ops, "MOMENT_SUMM")) else {}
    if vid_len is not None:
        kw["vid_len"] = vid_len
    return ops.convse_rerank(q_lin, [feat2[m] for m in mods], [mask[m] for m in mods], pair_vid,
                             model._conv_weights(), index.l_ref, merged, model.config.conv_kernel_size, softmax=True,
                             zero_skipped=zero_skipped, **kw)


def pad_moment_tail(flat_scores, flat_indices, k_videos, l_ref, min_pred_l=None, max_pred_l=None):
    """Opt-in reference-shaped tail of a moment list.  The reference sorts the WHOLE (k, L, L) product tensor and always
    returns max_before_nms rows (xml/inference.py:381-386; SVMR: utils/tensor_utils.py:133-141): when fewer candidates
    have a positive score -- short videos, the length mask -- its tail is rows of score exactly 0 in an order torch.sort
    leaves unspecified.  The kernels mark those rows flat = -1; this fills them with the lowest flat positions of the
    (k, L, L) tensor that are NOT in the list (a list with empty rows holds every positive-score candidate, so every other
    position scores exactly 0 in the reference too) -- the order a stable descending sort would produce.
    Index plumbing on the device, in place; returns (flat_scores, flat_indices)."""
    n_out = flat_indices.shape[1]
    dev = flat_indices.device
    total = int(k_videos) * l_ref * l_ref
    cnt = (flat_indices >= 0).sum(1)
    rows = torch.nonzero(cnt < n_out, as_tuple=False).reshape(-1)
    if rows.numel() == 0:
        return flat_scores, flat_indices
    if total < n_out:          # (the reference itself fails when max_before_nms exceeds k * L * L)
        raise ValueError("pad_tail: max_before_nms = %d exceeds the %d positions of the (k, L, L) tensor" % (n_out, total))
    # among the first 2 n_out positions at least n_out are not in a list of < n_out entries
    cand = torch.arange(min(total, 2 * n_out), device=dev, dtype=flat_indices.dtype)
    pos = torch.arange(n_out, device=dev)
    for c in range(0, rows.numel(), 256):
        r = rows[c:c + 256]
        have = flat_indices[r]                                                        # (b, n_out), -1 = empty
        # membership by scatter, O(b (|cand| + n_out)) memory (a (b, |cand|, n_out) comparison is 512 MB per chunk at the
        # evaluation default max_before_nms = 1000): list entries inside [0, |cand|) mark their position as taken
        taken = torch.zeros((r.numel(), cand.numel() + 1), dtype=torch.bool, device=dev)
        slot = torch.where((have >= 0) & (have < cand.numel()), have, torch.full_like(have, cand.numel())).long()
        taken.scatter_(1, slot, True)
        free = ~taken[:, :cand.numel()]                                               # (b, |cand|): not in the list
        order = torch.argsort((~free).to(torch.int8), dim=1, stable=True)             # free positions first, ascending
        fill = cand[order[:, :n_out]]                                                 # (b, n_out)
        k = (pos[None, :] - cnt[r][:, None])                                          # index into fill for the empty rows
        empty = k >= 0
        flat_indices[r] = torch.where(empty, torch.gather(fill, 1, k.clamp_min(0)), have)
        flat_scores[r] = flat_scores[r].masked_fill(empty, 0.0)
    return flat_scores, flat_indices


def stage_video_topk(model, index, qvec, max_vcmr_video=100, q2c_alpha=20.0, ops=hip_ops, external_top=None,
                     defer_exact_check=False):
    """K6 + K8 (or the exact-rank chain, or a caller's video lists): (q2c or None, top_w (Nq, K) f32 = exp(alpha s) desc,
    top_i (Nq, K) int32, exact-rank info or None).  Per query independent of the rest of the batch."""
    exact = None
    if external_top is None and index.exact is not None:
        q2c = None          # (the f32 (Nq, Nv) matrix is never formed; exact["q2c_filter"] is the bf16 pass's)
        # defer_exact_check (split-f16 exact mode): no host read-back inside the pass; out["exact"]["overflow_dev"] (device
        # bool, None when every video is a candidate) says whether the on-device second tier's capacity was exceeded
        top_w, top_i, exact = stage_exact_topk(index, qvec, min(max_vcmr_video, index.n_videos), q2c_alpha, ops,
                                               **(dict(defer_check=True) if defer_exact_check else {}))
    elif external_top is None:
        q2c = stage_q2c(index, qvec, ops)
        k = min(max_vcmr_video, index.n_videos)
        top_w, top_i = ops.topk_rows(q2c, k, alpha=q2c_alpha)
    else:   # external video-retrieval results replace K6/K8 (xml/inference.py:349-355): (meta idx int32, exp(alpha*s))
        q2c = None
        top_i, top_w = external_top
    return q2c, top_w, top_i, exact


def stage_moments(model, index, qvec, top_w, top_i, min_pred_l=2, max_pred_l=16, max_before_nms=200, ops=hip_ops,
                  pad_tail=False, q_lin=None):
    """K7 + K9 on the selected (query, video) pairs -> (flat_scores (Nq, n) f32 desc, flat_indices (Nq, n) int32).
    q_lin: the query linears' outputs when the caller already has them (fork_query_linears)."""
    if hasattr(ops, "MOMENT_SUMM") and K7_SUMMARIES:
        # K7 hands K9 the 8 largest row maxima of every pair (taken while the rows were in its registers): K9 reads the
        # 1 GB of span probabilities once instead of twice
        st, ed, summ = stage_span_probs(model, index, qvec, top_i, ops, pair_w=top_w.contiguous(), band=(min_pred_l, max_pred_l))
        fs, fi = ops.moment_topk(st, ed, top_w, index.l_ref, min_pred_l, max_pred_l, max_before_nms, summ=summ)
    else:
        vl = ragged_lengths(index, ops)
        rk = dict(pair_vid=top_i, vid_len=vl) if vl is not None else {}
        st, ed = stage_span_probs(model, index, qvec, top_i, ops, vid_len=vl, q_lin=q_lin)
        fs, fi = ops.moment_topk(st, ed, top_w, index.l_ref, min_pred_l, max_pred_l, max_before_nms, **rk)
    if pad_tail:
        pad_moment_tail(fs, fi, top_i.shape[1], index.l_ref, min_pred_l, max_pred_l)
    return fs, fi


def vcmr_search(model, index, query_feat, query_mask, max_vcmr_video=100, max_before_nms=200, q2c_alpha=20.0,
                min_pred_l=2, max_pred_l=16, svmr_video=None, ops=hip_ops, external_top=None, pad_tail=False,
                defer_exact_check=False, n_valid_tokens=None):
    """Device part of compute_query2ctx_info for one query batch (xml/inference.py:308-386), single GPU.

    Returns device tensors:
      top_scores (Nq,K) f32 = exp(alpha*q2c) desc, top_indices (Nq,K) int32 video (meta) indices,
      flat_scores (Nq,n) f32 desc, flat_indices (Nq,n) int32 into (K, l_ref, l_ref)  [-1 = no candidate]
      and, if svmr_video (Nq,) int32 is given, svmr_scores / svmr_flat (Nq,n) over (l_ref, l_ref).
    external_top = (indices (Nq,K) int32, weights (Nq,K) f32): use these videos / weights instead of K6 + K8.
    pad_tail=True: always max_before_nms rows, the reference's shape -- missing candidates become zero-score rows at
    length-masked positions (pad_moment_tail) instead of flat = -1.
    n_valid_tokens (host int): query_mask.sum() when the masks were built on the host (prefix masks) -- the packed query
    encoder of large batches then needs no read-back and the whole pass is enqueued without a host synchronisation."""
    qvec = stage_query_vectors(model, query_feat, query_mask, n_valid_tokens)
    forked = fork_query_linears(model, index, qvec, ops) if not (hasattr(ops, "MOMENT_SUMM") and K7_SUMMARIES) else None
    q2c, top_w, top_i, exact = stage_video_topk(model, index, qvec, max_vcmr_video, q2c_alpha, ops, external_top,
                                                defer_exact_check)
    q_lin = None
    if forked is not None:
        q_lin, lin_done = forked
        torch.cuda.current_stream(q_lin[0].device).wait_event(lin_done)
    fs, fi = stage_moments(model, index, qvec, top_w, top_i, min_pred_l, max_pred_l, max_before_nms, ops, pad_tail, q_lin=q_lin)
    out = dict(q2c=q2c, top_scores=top_w, top_indices=top_i, flat_scores=fs, flat_indices=fi)
    if exact is not None:
        out["exact"] = exact
    if svmr_video is not None:
        pv = svmr_video.to(torch.int32).reshape(-1, 1).contiguous()
        st1, ed1 = stage_span_probs(model, index, qvec, pv, ops, q_lin=q_lin)
        ss, sf = ops.moment_topk(st1, ed1, None, index.l_ref, min_pred_l, max_pred_l, max_before_nms)
        if pad_tail:
            pad_moment_tail(ss, sf, 1, index.l_ref, min_pred_l, max_pred_l)
        out.update(svmr_scores=ss, svmr_flat=sf, svmr_st=st1[:, 0], svmr_ed=ed1[:, 0])
    return out


_HOST_STREAMS = {}


def _host_streams(device):
    key = str(torch.device(device))
    if key not in _HOST_STREAMS:
        _HOST_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _HOST_STREAMS[key]


class HostSearchBuffers(object):
    """Pinned host memory + device staging of vcmr_search_host, allocated once and reused across calls (pinning a gigabyte of
    queries is a page-locking system call, not part of a search)."""

    def __init__(self, n_queries, width, chunk, lq, d_in, ragged_rows, device, dtype):
        self.rec = torch.empty((n_queries, width, 4), dtype=torch.int32, pin_memory=True)      # xml_moment records
        self.cnt = torch.empty((n_queries,), dtype=torch.int32, pin_memory=True)
        self.rec_dev = torch.empty((n_queries, width, 4), dtype=torch.int32, device=device)
        self.cnt_dev = torch.zeros((n_queries,), dtype=torch.int32, device=device)
        # two staging sets: chunk c + 1 is copied in while chunk c is searched
        if ragged_rows:
            self.stage = [dict(rows=torch.empty((ragged_rows, d_in), dtype=dtype, device=device),
                               start=torch.empty((chunk + 1,), dtype=torch.int64, device=device)) for _ in range(2)]
        else:
            self.stage = [dict(qf=torch.empty((chunk, lq, d_in), dtype=dtype, device=device),
                               qm=torch.empty((chunk, lq), dtype=torch.float32, device=device)) for _ in range(2)]
        # ONE pair of copy streams per device, shared by every buffer set: the runtime multiplexes streams onto a few hardware
        # queues (four by default) and a stream that lands on the compute stream's queue executes in order WITH it -- a fifth
        # stream made copies and K6 take turns instead of overlapping (profiles/r06_h2h_notes.md)
        self.copy_stream, self.back_stream = _host_streams(device)   # host -> device, device -> host (PCIe is full duplex)
        self.last_done = None                                       # event: the pass that last used these buffers has finished
        self.freed = [None, None]                                   # events: staging set i has been read by its chunk


class PendingHostSearch(object):
    """A vcmr_search_host pass that has been enqueued (wait=False).  result() blocks until its records are in host memory."""

    def __init__(self, buffers, done, fill_timings):
        self.buffers, self.done, self._fill = buffers, done, fill_timings

    def result(self):
        from .results import MOMENT_DTYPE
        self.done.synchronize()
        if self._fill is not None:
            self._fill()
            self._fill = None
        return self.buffers.rec.numpy().view(MOMENT_DTYPE)[..., 0], self.buffers.cnt.numpy()


def host_chunks(nq, first=1024, growth=3, largest=16384):
    """Query ranges of vcmr_search_host: a small first chunk (its copy is the only one nothing overlaps), then chunks growing
    by `growth` -- a chunk's copy (PCIe: ~1.7 ms per 1 024 padded f32 queries) hides behind the previous chunk's K6 (~6.5 ms
    per 1 024 queries on the TVR corpus) as long as it is not more than ~4 x as long.  Boundaries are multiples of 1 024
    queries = four of K6's 256-row query tiles."""
    first = max(256, (int(first) // 256) * 256)
    out, b, n = [], 0, first
    while b < nq:
        e = min(nq, b + n)
        if nq - e < first // 2:          # no sliver at the end
            e = nq
        out.append((b, e))
        b, n = e, min(n * growth, largest)          # (bounded staging memory for very large query sets)
    return out


def vcmr_search_host(model, index, query_feat, query_mask=None, row_start=None, meta2vid=None, chunk=1024, lq=None,
                     clip_length=1.5, buffers=None, timings=None, ops=hip_ops, max_vcmr_video=100, max_before_nms=200,
                     q2c_alpha=20.0, min_pred_l=2, max_pred_l=16, pad_tail=False, chunk_growth=3, wait=True):
    """VCMR from host memory to host memory: the path of the reference's query loop (xml/inference.py:302-314 moves every
    batch host -> device, :383-386 moves the lists device -> host; start_end_dataset.py:362-370) for a whole query set.

    Queries arrive in one of two host layouts (pinned for the copies to be asynchronous):
      * query_feat (Nq, Lq, D) f32 + query_mask (Nq, Lq) f32 -- what the reference's collate delivers (padded);
      * query_feat (rows, D) f16 / f32 + row_start (Nq + 1,) int64 -- the feature store's layout (ingest.FeatureStore): the
        queries' token rows back to back, un-normalised; the collate (truncate to lq, l2-normalise, pad, mask) runs on the
        device (xml_ingest_rows).
    The set is cut into growing chunks (host_chunks).  Chunk c + 1 is copied host -> device on a side stream while chunk c
    runs the per-query half of the search -- query encoder, K6 against the whole corpus, K8.  The pair half (K7 reads every
    selected video's feat2 tile once per launch whatever the number of queries: 8.6 GB on the TVR corpus -- one launch, not
    one per chunk) and K9 then run once over all queries, xml_moments_decode turns the lists into [video_idx, st, ed,
    score] records and ONE device -> host copy returns them.  Per query the arithmetic is that of vcmr_search on the whole
    set: the records are bitwise the single-launch records.
    meta2vid (Nv,) int32 device: meta index -> video2idx value (None: the meta index itself).
    Returns (records (Nq, n) numpy view of results.MOMENT_DTYPE on the pinned buffer, count (Nq,) int32 numpy).
    wait=False: returns a PendingHostSearch as soon as the pass is enqueued; its result() gives the same pair.  Two passes can
    be in flight (two buffer sets are kept on the index): the first copy of pass i + 1 then runs under the pair half of pass
    i and the device never idles between query sets -- an idle gap of a few milliseconds costs the next K6 launch 2-3 % by
    itself (clock / cache state; tools/bench_idle_effect.py).  The records of a pass stay valid until the second next pass.
    timings (dict, optional): h2d_s / device_s / decode_s / d2h_s from HIP events (copy and compute overlap: the wall clock
    is the caller's to take)."""
    from .results import MOMENT_DTYPE
    import time as _time
    t_enter = _time.perf_counter()
    dev = next(model.parameters()).device
    ragged = row_start is not None
    nq = (row_start.numel() - 1) if ragged else query_feat.shape[0]
    d_in = query_feat.shape[-1]
    lq = int(lq if lq is not None else (model.config.max_desc_l if ragged else query_feat.shape[1]))
    width = int(max_before_nms)
    bounds = host_chunks(nq, chunk, chunk_growth)
    rs_host = row_start.numpy() if ragged else None
    max_q = max(e - b for b, e in bounds)
    max_rows = max(int(rs_host[e] - rs_host[b]) for b, e in bounds) if ragged else 0
    if buffers is None:
        # kept on the index between calls: pinning host pages and mapping device memory are system calls, not search time
        key = (nq, width, max_q, lq, d_in, max_rows, str(dev), query_feat.dtype)
        cache = index.__dict__.setdefault("_host_search_buffers", {})
        if key not in cache:
            cache.clear()
            cache[key] = [[HostSearchBuffers(nq, width, max_q, lq, d_in, max_rows, dev, query_feat.dtype) for _ in range(2)], 0]
        ring = cache[key]
        buffers = ring[0][ring[1]]
        ring[1] ^= 1
    if buffers.last_done is not None:
        buffers.last_done.synchronize()                  # (the pass before the previous one: normally long finished)
    main = torch.cuda.current_stream(dev)
    side = buffers.copy_stream
    if meta2vid is None:
        meta2vid = torch.arange(index.n_videos, dtype=torch.int32, device=dev)
    copied, freed = [None, None], buffers.freed
    ev_h2d = []

    def evt():
        return torch.cuda.Event(enable_timing=timings is not None)

    def send(c):
        b, e = bounds[c]
        st = buffers.stage[c & 1]
        with torch.cuda.stream(side):
            if freed[c & 1] is not None:
                side.wait_event(freed[c & 1])          # the chunk that read this staging set is done with it
            t0, t1 = evt(), evt()
            t0.record(side)
            if ragged:
                r0, r1 = int(rs_host[b]), int(rs_host[e])
                st["rows"][:r1 - r0].copy_(query_feat[r0:r1], non_blocking=True)
                st["start"][:e - b + 1].copy_(row_start[b:e + 1], non_blocking=True)
            else:
                st["qf"][:e - b].copy_(query_feat[b:e], non_blocking=True)
                st["qm"][:e - b].copy_(query_mask[b:e], non_blocking=True)
            t1.record(side)
            ev_h2d.append((t0, t1))
            copied[c & 1] = t1

    with torch.no_grad():
        send(0)
        t_sent = _time.perf_counter()
        qvecs, tws, tis = [], [], []
        t_first = evt()
        for c, (b, e) in enumerate(bounds):
            if c + 1 < len(bounds):
                send(c + 1)
            st = buffers.stage[c & 1]
            main.wait_event(copied[c & 1])
            if c == 0:
                t_first.record(main)
            # the host holds the queries' lengths: the packed encoder is told its token count instead of reading it back
            # (a read-back per chunk is a host synchronisation per chunk)
            if ragged:
                r0 = int(rs_host[b])
                qf, qm = ops.ingest_rows(st["rows"], st["start"][:e - b + 1] - r0, e - b, lq, lq, normalize=True)
                ln = np.minimum(np.diff(rs_host[b:e + 1]), lq)
                n_tok = int(ln.sum()) if (e > b and ln.min() >= 1) else None
            else:
                qf, qm = st["qf"][:e - b], st["qm"][:e - b]
                mh = query_mask[b:e].numpy()
                prefix = bool((mh[:, 0] == 1).all() and ((mh == 0) | (mh == 1)).all() and (mh[:, 1:] <= mh[:, :-1]).all())
                n_tok = int(mh.sum()) if prefix else None
            qvec = stage_query_vectors(model, qf, qm, n_tok)
            done = evt()
            done.record(main)                            # the staging set is free once the query encoder has read it
            freed[c & 1] = done
            _, tw, ti, _ = stage_video_topk(model, index, qvec, max_vcmr_video, q2c_alpha, ops)
            qvecs.append(qvec), tws.append(tw), tis.append(ti)
        one = len(bounds) == 1
        qvec = {m: (qvecs[0][m] if one else torch.cat([q[m] for q in qvecs])) for m in qvecs[0]}
        top_w = tws[0] if one else torch.cat(tws)
        top_i = tis[0] if one else torch.cat(tis)
        fs, fi = stage_moments(model, index, qvec, top_w, top_i, min_pred_l, max_pred_l, max_before_nms, ops, pad_tail)
        t1, t2, t2b, t3 = evt(), evt(), evt(), torch.cuda.Event(enable_timing=timings is not None)
        t1.record(main)
        ops.moments_decode(fs, flat=fi, top_idx=top_i, meta2vid=meta2vid, l_ref=index.l_ref, clip_length=clip_length,
                           seconds=True, out=buffers.rec_dev, out_count=buffers.cnt_dev)
        t2.record(main)
        back = buffers.back_stream                       # the records leave on their own stream: `main` is free for the next pass
        back.wait_event(t2)
        with torch.cuda.stream(back):
            t2b.record(back)
            buffers.rec.copy_(buffers.rec_dev, non_blocking=True)
            buffers.cnt.copy_(buffers.cnt_dev, non_blocking=True)
            t3.record(back)
        buffers.last_done = t3
    t_enq = _time.perf_counter()

    def fill():
        timings.update(host_before_first_copy_s=t_sent - t_enter, host_enqueue_s=t_enq - t_sent,
                       h2d_s=sum(a.elapsed_time(b) for a, b in ev_h2d) * 1e-3,
                       h2d_exposed_s=ev_h2d[0][0].elapsed_time(ev_h2d[0][1]) * 1e-3,
                       device_s=t_first.elapsed_time(t1) * 1e-3, decode_s=t1.elapsed_time(t2) * 1e-3,
                       d2h_s=t2b.elapsed_time(t3) * 1e-3, chunks=len(bounds), chunk_queries=[e - b for b, e in bounds])
    pending = PendingHostSearch(buffers, t3, fill if timings is not None else None)
    return pending.result() if wait else pending


class GraphedVcmrSearch(object):
    """vcmr_search for ONE fixed query-batch shape, captured once into a HIP graph and replayed.

    The reference serves queries in batches of `eval_query_bsz` = 50 (xml/config.py); at that size the pass is a
    chain of ~25 short kernels and the time goes to launch latency, not to the kernels.  One graph launch replaces
    the chain.  Inputs are copied into static buffers, the returned tensors are the graph's static outputs
    (overwritten by the next call: copy them if they must outlive it).  Weights are packed and workspaces sized by
    two eager warm-up passes before the capture; the corpus index and the model weights must not be re-allocated
    afterwards (re-create the object after load_state_dict / an optimizer step)."""

    def __init__(self, model, index, nq, lq, d_in, **search_kwargs):
        if index.exact is not None and index.exact.mode != "f16s":
            raise ValueError("the f32 exact-rank mode reads its certificate on the host (the fallback's launch shape "
                             "depends on it): not capturable -- build the index from an ops.F16S model")
        search_kwargs = dict(search_kwargs)
        self.exact = index.exact is not None
        if self.exact:           # split-f16 exact mode: second tier on the device, overflow flag checked after the replay
            search_kwargs["defer_exact_check"] = True
        dev = next(model.parameters()).device
        self.query_feat = torch.zeros((nq, lq, d_in), dtype=torch.float32, device=dev)
        self.query_mask = torch.zeros((nq, lq), dtype=torch.float32, device=dev)
        self.query_mask[:, 0] = 1.0
        self._args = (model, index)
        self._kw = search_kwargs
        # the packed-token query encoder (large batches) reads its plan back on the host and launches shapes that depend on
        # the batch's valid-token count: the graph keeps the padded path -- in the warm-ups too, so that the workspaces they
        # size are the captured path's
        from . import model_xml
        pack_was, model_xml.PACK_QUERY_TOKENS = model_xml.PACK_QUERY_TOKENS, False
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):
                    vcmr_search(model, index, self.query_feat, self.query_mask, **self._kw)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.out = vcmr_search(model, index, self.query_feat, self.query_mask, **self._kw)
        finally:
            model_xml.PACK_QUERY_TOKENS = pack_was

    def __call__(self, query_feat, query_mask):
        if tuple(query_feat.shape) != tuple(self.query_feat.shape) or tuple(query_mask.shape) != tuple(self.query_mask.shape):
            raise ValueError("GraphedVcmrSearch was captured for queries %s / masks %s, got %s / %s"
                             % (tuple(self.query_feat.shape), tuple(self.query_mask.shape), tuple(query_feat.shape),
                                tuple(query_mask.shape)))
        self.query_feat.copy_(query_feat)
        self.query_mask.copy_(query_mask)
        self.graph.replay()
        if self.exact and self.out["exact"]["overflow_dev"] is not None and bool(self.out["exact"]["overflow_dev"].item()):
            # more failing queries / closer scores than the captured second tier holds: this batch through the eager pass
            model, index = self._args
            kw = {k: v for k, v in self._kw.items() if k != "defer_exact_check"}
            with torch.no_grad():
                return vcmr_search(model, index, self.query_feat, self.query_mask, **kw)
        return self.out


def decode_flat(flat, l_ref):
    """(r, st_idx, ed_idx) of the reference's flat index over (K, L, L) (np.unravel_index, :423-425)."""
    flat = np.asarray(flat).astype(np.int64)
    r = flat // (l_ref * l_ref)
    rem = flat - r * l_ref * l_ref
    return r, rem // l_ref, rem % l_ref


class _GraphedBatches(object):
    """compute_query2ctx_info's per-batch search through ONE captured graph (opt.graph_search): batches are padded on the host
    into pinned buffers of the captured shape (eval_query_bsz, max_desc_l, D), sent in one asynchronous copy each, and replayed.
    A batch of 50 queries is ~25 short kernels: issued one by one the host, not the device, sets the pace (TVR val, 218
    batches: 0.19 s of search of which 0.09 s device time)."""

    def __init__(self, model, index, bsz, lq, d_in, with_gt, **search_kwargs):
        dev = next(model.parameters()).device
        self.gt = torch.zeros(bsz, dtype=torch.int32, device=dev) if with_gt else None
        self.g = GraphedVcmrSearch(model, index, bsz, lq, d_in, svmr_video=self.gt, **search_kwargs)
        self.bsz, self.lq, self.d = bsz, lq, d_in
        self.ring = [dict(qf=torch.zeros((bsz, lq, d_in), dtype=torch.float32, pin_memory=True),
                          qm=torch.zeros((bsz, lq), dtype=torch.float32, pin_memory=True),
                          gt=torch.zeros(bsz, dtype=torch.int32, pin_memory=True), ev=None) for _ in range(3)]
        for r in self.ring:
            r["np"] = (r["qf"].numpy(), r["qm"].numpy(), r["gt"].numpy())
        self.i = 0

    def fits(self, seqs):
        return len(seqs) <= self.bsz and max(len(a) for a in seqs) <= self.lq and seqs[0].shape[1] == self.d

    def __call__(self, seqs, gt_rows):
        r = self.ring[self.i]
        self.i = (self.i + 1) % len(self.ring)
        if r["ev"] is not None:
            r["ev"].synchronize()                  # the copies that last read this pinned set are done
        qf, qm, gt = r["np"]
        n = len(seqs)
        lens = np.fromiter((len(a) for a in seqs), dtype=np.int64, count=n)
        valid = np.arange(self.lq)[None, :] < lens[:, None]
        qf[:] = 0.0
        qf[:n][valid] = np.concatenate(seqs, axis=0)
        qm[:] = 0.0
        qm[:n] = valid
        qm[n:, 0] = 1.0                            # filler rows of the last batch: one valid (zero) token, results ignored
        g = self.g
        g.query_feat.copy_(r["qf"], non_blocking=True)
        g.query_mask.copy_(r["qm"], non_blocking=True)
        if self.gt is not None:
            gt[:] = 0
            gt[:n] = gt_rows
            self.gt.copy_(r["gt"], non_blocking=True)
        r["ev"] = torch.cuda.Event()
        r["ev"].record()
        g.graph.replay()
        return g.out


class _ResultSink(object):
    """Device-side result buffer of one task for a whole query set: K10 writes each batch's 16-byte records into its rows,
    the host fetches the buffer ONCE at the end (no per-batch synchronisation, no per-query Python)."""

    def __init__(self, n_queries, width, device):
        self.rec = torch.empty((n_queries, width, 4), dtype=torch.int32, device=device)
        self.cnt = torch.zeros((n_queries,), dtype=torch.int32, device=device)
        self.n = 0

    def rows(self, b, nb):
        self.n = max(self.n, b + nb)
        return dict(out=self.rec[b:b + nb], out_count=self.cnt[b:b + nb])

    def fetch(self, desc_ids, descs, scale=None, int_spans=False):
        from .results import MOMENT_DTYPE, MomentResults
        n = self.n
        rec = self.rec[:n].cpu().numpy().view(MOMENT_DTYPE)[..., 0]
        return MomentResults.from_records(desc_ids[:n], descs[:n], rec, self.cnt[:n].cpu().numpy(), scale=scale,
                                          int_spans=int_spans)


def compute_query2ctx_info(model, eval_dataset, opt, ctx_info, max_before_nms=1000, max_n_videos=100,
                           tasks=("SVMR",), ops=hip_ops, as_arrays=False):
    """Mirror of compute_query2ctx_info (xml/inference.py:252-445).  Same result dict:
    {"VCMR"|"SVMR"|"VR": [dict(desc_id, desc, predictions=[[video_idx, st, ed, score], ...]), ...]}.
    opt.external_inference_vr_res_path (xml/inference.py:264-273,349-355): re-rank the videos of another model's VR
    submission instead of this model's own top-k.
    opt.pad_tail=True (not a reference option): every list has max_before_nms rows like the reference's -- zero-score rows
    where fewer candidates exist -- instead of the positive-score prefix.
    as_arrays=True: each task as a results.MomentResults (the (Nq, n) columns K10 produced) instead of nested lists; the
    reference's "numpy tail" (:391-445) is the device epilogue xml_moments_decode either way -- the lists, when asked for,
    are built from the columns in one C call (results.MomentResults.to_list)."""
    is_svmr, is_vr, is_vcmr = "SVMR" in tasks, "VR" in tasks, "VCMR" in tasks
    index = ctx_info["index"]
    video2idx = eval_dataset.video2idx
    video_metas = ctx_info["video_metas"]
    external_query2video = None
    if getattr(opt, "external_inference_vr_res_path", None) is not None:
        # load_external_vr_res2 (xml/inference.py:244-249,264-273): desc_id -> top video predictions of another model
        import json
        from .postproc import get_submission_top_n
        with open(opt.external_inference_vr_res_path, "r") as f:
            ext = get_submission_top_n(json.load(f), top_n=max_n_videos)["VR"]
        external_query2video = {e["desc_id"]: e["predictions"] for e in ext}
        video_idx2meta_idx = {video2idx[m["vid_name"]]: i for i, m in enumerate(video_metas)}
    meta_vid = np.array([video2idx[m["vid_name"]] for m in video_metas])
    meta2vid = torch.from_numpy(meta_vid.astype(np.int32)).to(opt.device)
    eval_dataset.set_data_mode("query")
    eval_dataset.load_gt_vid_name_for_query(is_svmr)
    name2meta = {e["vid_name"]: i for i, e in enumerate(video_metas)}
    clip = opt.clip_length
    l_ref = index.l_ref
    n = len(eval_dataset)
    sink_vcmr = _ResultSink(n, max_before_nms, opt.device) if is_vcmr else None
    sink_svmr = _ResultSink(n, max_before_nms, opt.device) if is_svmr else None
    sink_vr = None
    desc_ids, descs = [], []
    search_kw = dict(max_vcmr_video=max_n_videos, max_before_nms=max_before_nms, q2c_alpha=opt.q2c_alpha,
                     min_pred_l=opt.min_pred_l, max_pred_l=opt.max_pred_l, ops=ops, pad_tail=getattr(opt, "pad_tail", False))
    graphed = None          # opt.graph_search (not a reference option): the batches through one captured graph
    want_graph = (getattr(opt, "graph_search", False) and external_query2video is None and ops is hip_ops and n > 0
                  and torch.device(opt.device).type == "cuda" and not getattr(opt, "debug", False))
    def emit(out, b, nb, gt):
        """K10 on the device: (flat index, local rank) -> [video_idx, st, ed, score] records (xml/inference.py:402-439)"""
        nonlocal sink_vr
        if is_vr:
            n_vr = min(100, out["top_indices"].shape[1])
            if sink_vr is None:
                sink_vr = _ResultSink(n, n_vr, opt.device)
            ops.moments_decode(out["top_scores"], top_idx=out["top_indices"], meta2vid=meta2vid, n=n_vr,
                               **sink_vr.rows(b, nb))
        if is_vcmr:
            ops.moments_decode(out["flat_scores"], flat=out["flat_indices"], top_idx=out["top_indices"], meta2vid=meta2vid,
                               l_ref=l_ref, clip_length=clip, seconds=True, **sink_vcmr.rows(b, nb))
        if is_svmr:     # clip units on the device; the float64 scaling of get_svmr_res_from_st_ed_probs (:229-233) on the host
            ops.moments_decode(out["svmr_scores"], flat=out["svmr_flat"], row_vid=gt, meta2vid=meta2vid, l_ref=l_ref,
                               seconds=False, **sink_svmr.rows(b, nb))

    # Replayed batches of a split-f16 exact-rank index: the captured second tier can overflow (more failing queries / closer
    # scores than it holds).  GraphedVcmrSearch.__call__ reads the flag back after every replay; here that would put a host
    # synchronisation into every batch, so each batch's flag is kept on the device and the flagged batches are searched
    # again eagerly -- and their records overwritten -- before the sinks are fetched.
    exact_flags = []        # (first query, device flag)
    for b in range(0, n, opt.eval_query_bsz):
        items = [eval_dataset[i] for i in range(b, min(n, b + opt.eval_query_bsz))]
        metas = [e["meta"] for e in items]
        nb = len(metas)
        desc_ids.extend(m["desc_id"] for m in metas)
        descs.extend(m["desc"] for m in metas)
        seqs = [e["model_inputs"]["query_feat"] for e in items]
        if want_graph and graphed is None:
            want_graph = False
            arr0 = seqs[0].detach().cpu().numpy() if isinstance(seqs[0], torch.Tensor) else np.asarray(seqs[0])
            try:
                lq = min(int(getattr(opt, "max_desc_l", 30)), int(model.config.max_desc_l))
                graphed = _GraphedBatches(model, index, opt.eval_query_bsz, lq, int(arr0.shape[1]), is_svmr, **search_kw)
            except ValueError:      # (an index whose exact-rank mode is not capturable: the eager pass)
                graphed = None
        if graphed is not None:
            seqs = [a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a) for a in seqs]
        if graphed is not None and graphed.fits(seqs):
            gt_rows = np.fromiter((name2meta[m["vid_name"]] for m in metas), dtype=np.int32, count=nb) if is_svmr else None
            out = graphed(seqs, gt_rows)
            gt = graphed.gt[:nb] if is_svmr else None
            if out.get("exact") is not None and out["exact"].get("overflow_dev") is not None:
                exact_flags.append((b, out["exact"]["overflow_dev"].reshape(-1)[:1].clone()))
            out = {k: (v[:nb] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == graphed.bsz else v)
                   for k, v in out.items()}
            qf = None
        else:
            qf, qm = pad_batch(seqs, opt.device)
            gt = None
            if is_svmr:
                gt = torch.tensor([name2meta[m["vid_name"]] for m in metas], dtype=torch.int32, device=opt.device)
        external_top = None
        if external_query2video is not None:
            info = [external_query2video[m["desc_id"]] for m in metas]
            ext_i = torch.tensor([[video_idx2meta_idx[p[0]] for p in e] for e in info], dtype=torch.int32)
            ext_w = torch.exp(opt.q2c_alpha * torch.tensor([[p[3] for p in e] for e in info], dtype=torch.float32))
            external_top = (ext_i.to(opt.device).contiguous(), ext_w.to(opt.device).contiguous())
        if qf is not None:
            out = vcmr_search(model, index, qf, qm, svmr_video=gt, external_top=external_top, **search_kw)
        emit(out, b, nb, gt)
        if getattr(opt, "debug", False):
            break
    if exact_flags:
        flagged = torch.cat([f for _, f in exact_flags]).cpu().numpy()            # ONE read-back for the whole query set
        for (b, _), over in zip(exact_flags, flagged):
            if not over:
                continue
            items = [eval_dataset[i] for i in range(b, min(n, b + opt.eval_query_bsz))]
            qf, qm = pad_batch([e["model_inputs"]["query_feat"] for e in items], opt.device)
            gt = None
            if is_svmr:
                gt = torch.tensor([name2meta[e["meta"]["vid_name"]] for e in items], dtype=torch.int32, device=opt.device)
            emit(vcmr_search(model, index, qf, qm, svmr_video=gt, **search_kw), b, len(items), gt)
    res = {}
    if is_svmr:
        res["SVMR"] = sink_svmr.fetch(desc_ids, descs, scale=clip)
    if is_vcmr:
        res["VCMR"] = sink_vcmr.fetch(desc_ids, descs)
    if is_vr and sink_vr is not None:
        res["VR"] = sink_vr.fetch(desc_ids, descs, int_spans=True)
    res = {k: v for k, v in res.items() if len(v) != 0}
    if not as_arrays:
        from .results import to_lists
        res = to_lists(res)
    return res


def compute_query2ctx_info_svmr_only(model, eval_dataset, opt, ctx_info, max_before_nms=1000, max_n_videos=200,
                                     tasks=("SVMR",), ops=hip_ops, as_arrays=False):
    """Mirror of compute_query2ctx_info_svmr_only (xml/inference.py:107-167): every query is scored against its
    ground-truth video only (K7 with one pair per query + K9 with k = 1); no corpus-wide similarity."""
    index = ctx_info["index"]
    video2idx = eval_dataset.video2idx
    video_metas = ctx_info["video_metas"]
    name2meta = {e["vid_name"]: i for i, e in enumerate(video_metas)}
    meta2vid = torch.tensor([video2idx[m["vid_name"]] for m in video_metas], dtype=torch.int32).to(opt.device)
    eval_dataset.set_data_mode("query")
    eval_dataset.load_gt_vid_name_for_query(True)
    clip, l_ref = opt.clip_length, index.l_ref
    n = len(eval_dataset)
    sink = _ResultSink(n, max_before_nms, opt.device)
    desc_ids, descs = [], []
    for b in range(0, n, opt.eval_query_bsz):
        items = [eval_dataset[i] for i in range(b, min(n, b + opt.eval_query_bsz))]
        metas = [e["meta"] for e in items]
        desc_ids.extend(m["desc_id"] for m in metas)
        descs.extend(m["desc"] for m in metas)
        qf, qm = pad_batch([e["model_inputs"]["query_feat"] for e in items], opt.device)
        gt = torch.tensor([name2meta[m["vid_name"]] for m in metas], dtype=torch.int32, device=opt.device)
        qvec = stage_query_vectors(model, qf, qm)
        st1, ed1 = stage_span_probs(model, index, qvec, gt.reshape(-1, 1).contiguous(), ops)
        ss, sf = ops.moment_topk(st1, ed1, None, l_ref, opt.min_pred_l, opt.max_pred_l, max_before_nms)
        if getattr(opt, "pad_tail", False):
            pad_moment_tail(ss, sf, 1, l_ref, opt.min_pred_l, opt.max_pred_l)
        ops.moments_decode(ss, flat=sf, row_vid=gt, meta2vid=meta2vid, l_ref=l_ref, seconds=False, **sink.rows(b, len(metas)))
        if getattr(opt, "debug", False):
            break
    res = sink.fetch(desc_ids, descs, scale=clip)
    return dict(SVMR=res if as_arrays else res.to_list())


def get_eval_res(model, eval_dataset, opt, tasks, max_after_nms, ops=hip_ops, as_arrays=False):
    """Mirror of get_eval_res (xml/inference.py:448-464)."""
    context_info = compute_context_info(model, eval_dataset, opt, ops=ops)
    if "VCMR" in tasks or "VR" in tasks:
        eval_res = compute_query2ctx_info(model, eval_dataset, opt, context_info, max_before_nms=opt.max_before_nms,
                                          max_n_videos=opt.max_vcmr_video, tasks=tasks, ops=ops, as_arrays=as_arrays)
    else:
        eval_res = compute_query2ctx_info_svmr_only(model, eval_dataset, opt, context_info,
                                                    max_before_nms=opt.max_before_nms, max_n_videos=max_after_nms,
                                                    tasks=tasks, ops=ops, as_arrays=as_arrays)
    eval_res["video2idx"] = eval_dataset.video2idx
    return eval_res


def eval_epoch(model, eval_dataset, opt, tasks=("SVMR",), max_after_nms=100, ground_truth=None, ops=hip_ops,
               as_arrays=False, timings=None, ctx_info=None):
    """The in-memory part of eval_epoch (xml/inference.py:473-531): raw results -> top-n submission -> metrics, and
    the same again after temporal NMS when opt.nms_thd != -1.  (File writing stays with the caller.)
    Returns (submission, metrics, submission_after_nms, metrics_after_nms).

    Everything between the device and the metrics works on (Nq, n) arrays (results.MomentResults: K10's records, one D2H per
    task; batched NMS; the evaluator's array path); the reference's nested lists are built once at the end for the two
    returned submissions -- or not at all with as_arrays=True (the tasks stay MomentResults; `.to_list()` on demand).
    timings: dict that receives the wall-clock split {search, top_n, eval, nms, eval_nms, lists} in seconds.
    ctx_info: a compute_context_info result to reuse (skips the corpus encode).

    Reference quirk kept on purpose: get_submission_top_n truncates the RAW lists in place (clip_alignment_with_language/
    inference.py:503-515), so the NMS stage (xml/inference.py:507-515) only ever sees the first max_after_nms (100)
    candidates, not max_before_nms, and the after-NMS metrics are computed with eval_retrieval's default
    use_desc_type=True.  opt.nms_on_full_lists=True runs NMS on the untruncated lists instead (not the reference)."""
    import time
    from . import evaluate, postproc
    from .results import to_lists
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + t[-1] - t[-2]
    if ctx_info is None:
        raw = get_eval_res(model, eval_dataset, opt, tasks, max_after_nms, ops=ops, as_arrays=True)
    elif "VCMR" in tasks or "VR" in tasks:
        raw = compute_query2ctx_info(model, eval_dataset, opt, ctx_info, max_before_nms=opt.max_before_nms,
                                     max_n_videos=opt.max_vcmr_video, tasks=tasks, ops=ops, as_arrays=True)
        raw["video2idx"] = eval_dataset.video2idx
    else:
        raw = compute_query2ctx_info_svmr_only(model, eval_dataset, opt, ctx_info, max_before_nms=opt.max_before_nms,
                                               max_n_videos=max_after_nms, tasks=tasks, ops=ops, as_arrays=True)
        raw["video2idx"] = eval_dataset.video2idx
    lap("search")
    full = None
    if getattr(opt, "nms_on_full_lists", False):
        full = {k: (v if k == "video2idx" else v.copy()) for k, v in raw.items()}
    submission = postproc.get_submission_top_n(raw, top_n=max_after_nms)      # truncates `raw` in place, like the reference
    if full is not None:
        raw = full
    lap("top_n")
    use_desc_type = getattr(opt, "dset_name", "tvr") == "tvr"
    metrics = None
    if ground_truth is not None:
        metrics = evaluate.eval_retrieval(submission, ground_truth, iou_thds=(0.5, 0.7), verbose=False,
                                          match_number=not getattr(opt, "debug", False), use_desc_type=use_desc_type)
    lap("eval")
    sub_nms = metrics_nms = None
    if getattr(opt, "nms_thd", -1) != -1:
        sub_nms = dict(video2idx=raw["video2idx"])
        for k, fn in (("SVMR", postproc.post_processing_svmr_nms), ("VCMR", postproc.post_processing_vcmr_nms)):
            if k in raw:
                sub_nms[k] = fn(raw[k], nms_thd=opt.nms_thd, max_before_nms=opt.max_before_nms,
                                max_after_nms=max_after_nms)         # (arrays in, new arrays out: nothing to deep-copy)
        lap("nms")
        if ground_truth is not None:
            metrics_nms = evaluate.eval_retrieval(sub_nms, ground_truth, iou_thds=(0.5, 0.7), verbose=False,
                                                  match_number=not getattr(opt, "debug", False))
        lap("eval_nms")
    if not as_arrays:
        submission = to_lists(submission)
        sub_nms = to_lists(sub_nms) if sub_nms is not None else None
        lap("lists")
    return submission, metrics, sub_nms, metrics_nms
