"""Thin tensor-level wrappers over the C ABI (include/xmlhip.h).

PyTorch is plumbing here: it owns device memory (torch.empty), the current HIP stream and nothing else.
Every function launches hand-written gfx950 kernels from libxmlhip.so and fails loudly when the library or a
GPU tensor is missing -- there is no eager/CPU fallback.
"""
import ctypes

import os

import numpy as np

import torch

from . import _lib
from ._lib import XML_BF16, XML_F32, ConvseDesc, check

class _SplitF16(object):
    """dtype tag of the split-f16 forms (XML_F16S, include/xmlhip.h): f32-grade values carried as hi + lo halves and
    multiplied on the 16-bit MFMA pipe.  As a MODEL compute dtype (XML(cfg, compute_dtype=ops.F16S)) the activations stay
    torch.float32 and every projection weight is a SplitWeight."""

    def __repr__(self):
        return "tvretrieval_amd.ops.F16S"


F16S = _SplitF16()
_DT = {torch.float32: XML_F32, torch.bfloat16: XML_BF16, torch.float16: _lib.XML_F16, F16S: _lib.XML_F16S}
F16_UNIT_LOG2 = 14         # XML_F16_UNIT_LOG2: fixed scale of unit-norm rows in f16 / split-f16 form


def act_dtype(dtype):
    """Storage dtype of the activations of a model computing in `dtype`."""
    return torch.float32 if dtype is F16S else dtype


class SplitWeight(object):
    """A projection weight (n, k) packed by xml_pack_weights_f16s: (n, 3k) f16 [hi | hi | lo] at one power-of-two scale +
    a 16-byte trailer holding 1 / scale.  Quacks like a tensor for the wrappers below (shape / dtype / device / data_ptr)."""

    def __init__(self, data, n, k):
        self.data, self.shape, self.dtype, self.device, self.is_cuda = data, (int(n), int(k)), F16S, data.device, data.is_cuda

    def data_ptr(self):
        return self.data.data_ptr()

    def is_contiguous(self):
        return True


class SplitRows(object):
    """Rows of f32-grade values in split-f16 form (xml_split_f16_rows): `data` int32 (..., k) -- 4 bytes per element, per 32
    elements [32 x hi | 32 x lo] halves -- and `inv` f32 (...) = 1 / scale of every row."""

    def __init__(self, data, inv):
        self.data, self.inv, self.shape, self.dtype, self.device, self.is_cuda = data, inv, tuple(data.shape), F16S, \
            data.device, data.is_cuda

    def data_ptr(self):
        return self.data.data_ptr()

    def is_contiguous(self):
        return self.data.is_contiguous()

    def numel(self):
        return self.data.numel()

    def element_size(self):
        return 4

    def __getitem__(self, idx):
        return SplitRows(self.data[idx], self.inv[idx])

    def float(self):
        return unsplit_f16_rows(self)


def dt_of(t):
    try:
        return _DT[t.dtype if hasattr(t, "dtype") else t]
    except KeyError:
        raise _lib.XmlHipError("unsupported dtype %s" % (t.dtype if hasattr(t, "dtype") else t))


def _req(t, name, dtype=None):
    if not isinstance(t, (torch.Tensor, SplitWeight, SplitRows)):
        raise _lib.XmlHipError("%s: expected a tensor" % name)
    if not t.is_cuda:
        raise _lib.XmlHipError("%s: tensor must live on the GPU (got %s); the HIP path has no CPU fallback"
                               % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise _lib.XmlHipError("%s: expected dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.XmlHipError("%s: tensor must be contiguous" % name)
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = torch._C._cuda_getCurrentRawStream      # (device index) -> hipStream_t as int; ~0.3 us
_cur_device = torch._C._cuda_getDevice


def _stream():
    """The current HIP stream of the current device as a void*.  (torch.cuda.current_stream() builds a Stream object through
    three Python layers -- ~10 us per call, 90 calls per training step: 0.9 ms of a 6 ms step was spent asking for it.)"""
    return ctypes.c_void_p(_raw_stream(_cur_device()))


_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream).  The C ABI never allocates."""
    key = (device.index, _raw_stream(device.index if device.index is not None else _cur_device()))
    buf = _ws_cache.get(key)
    nbytes = max(int(nbytes), 256)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def convert(x, dtype):
    """xml_convert: dtype conversion on the device (round-to-nearest-even to bf16)."""
    _req(x, "x")
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_lib.load().xml_convert(_p(x), dt_of(x), _p(out), dt_of(dtype), x.numel(), _stream()), "xml_convert")
    return out


def pack_weights(w_f32, dtype, out=None):
    _req(w_f32, "w", torch.float32)
    if dtype is F16S:
        assert out is None, "pack_weights: out= is for the plain low-precision copies"
        return pack_weights_f16s(w_f32)
    if out is None:
        out = torch.empty(w_f32.shape, dtype=dtype, device=w_f32.device)
    else:
        _req(out, "out", dtype)
        assert out.numel() == w_f32.numel(), "pack_weights: out has the wrong size"
    check(_lib.load().xml_pack_weights(_p(w_f32), _p(out), dt_of(dtype), w_f32.numel(), _stream()), "xml_pack_weights")
    return out


def linear_ln_relu_pos(x, ln_in_g, ln_in_b, w, b, pos, ln_pos_g, ln_pos_b):
    """K1+K2.  x (N, L, D_in) f32 or compute dtype; w (H, ceil8(D_in)) compute dtype, zero columns beyond D_in
    -> (N, L, H)."""
    _req(x, "x"); _req(w, "w"); _req(pos, "pos", act_dtype(w.dtype))
    n, seq_len, d_in = x.shape
    hidden = w.shape[0]
    assert w.shape[1] == (d_in + 7) // 8 * 8 and pos.shape[1] == hidden and pos.shape[0] >= seq_len, "shape mismatch"
    for t, nm in ((ln_in_g, "ln_in_g"), (ln_in_b, "ln_in_b"), (b, "b"), (ln_pos_g, "ln_pos_g"), (ln_pos_b, "ln_pos_b")):
        _req(t, nm, torch.float32)
    lib = _lib.load()
    dt = dt_of(w)
    rows = n * seq_len
    y = torch.empty((n, seq_len, hidden), dtype=act_dtype(w.dtype), device=x.device)
    nb = lib.xml_linear_ln_relu_pos_workspace_bytes(rows, d_in, hidden, dt)
    ws = _workspace(nb, x.device)
    check(lib.xml_linear_ln_relu_pos(_p(x), dt_of(x), _p(ln_in_g), _p(ln_in_b), _p(w), _p(b), _p(pos), _p(ln_pos_g),
                                     _p(ln_pos_b), _p(y), rows, seq_len, d_in, hidden, dt, _p(ws), ws.numel(),
                                     _stream()), "xml_linear_ln_relu_pos")
    return y


def check_count(v, what):
    if v < 0:
        check(v, what)
    return v


ENCODE_INTO_INDEX = True      # attention_block(out=...) exists: inference.build_corpus_index encodes into the index tensors


def _req_w(w, name, x):
    """weight of a projection applied to activations x: x's dtype, or a SplitWeight for f32 activations (XML_F16S)."""
    _req(w, name)
    if not (w.dtype == x.dtype or (w.dtype is F16S and x.dtype == torch.float32)):
        raise _lib.XmlHipError("%s: weight dtype %s does not go with activations of dtype %s" % (name, w.dtype, x.dtype))
    return w


def attention_block(x, key_mask, wqkv, bqkv, wo, bo, ln_g, ln_b, n_heads, out=None):
    """K3+K4 (BertAttention).  x (N, L, H); key_mask (N, L) f32.  out: optional (N, L, H) contiguous destination (e.g. a
    slice of a preallocated index tensor), must not alias x."""
    _req(x, "x"); _req(key_mask, "key_mask", torch.float32); _req_w(wqkv, "wqkv", x); _req_w(wo, "wo", x)
    for t, nm in ((bqkv, "bqkv"), (bo, "bo"), (ln_g, "ln_g"), (ln_b, "ln_b")):
        _req(t, nm, torch.float32)
    n, seq_len, hidden = x.shape
    lib = _lib.load()
    dt = dt_of(wqkv)
    if out is None:
        y = torch.empty_like(x)
    else:
        _req(out, "out", x.dtype)
        assert out.shape == x.shape and out.data_ptr() != x.data_ptr()
        y = out
    ws = _workspace(lib.xml_attention_block_workspace_bytes(n, seq_len, hidden, dt), x.device)
    check(lib.xml_attention_block(_p(x), _p(key_mask), _p(wqkv), _p(bqkv), _p(wo), _p(bo), _p(ln_g), _p(ln_b), _p(y),
                                  n, seq_len, hidden, n_heads, dt, _p(ws), ws.numel(), _stream()),
          "xml_attention_block")
    return y


def attention_core(q, k, v, q_mask, k_mask, n_heads):
    """BertSelfAttention.forward behind its projections: q (N, Lq, H), k / v (N, Lk, H) projected states (compute dtype),
    k_mask (N, Lk) f32, q_mask (N, Lq) f32 or None -> context layer (N, Lq, H)."""
    _req(q, "q"); _req(k, "k", q.dtype); _req(v, "v", q.dtype); _req(k_mask, "k_mask", torch.float32)
    if q_mask is not None:
        _req(q_mask, "q_mask", torch.float32)
    n, lq, hidden = q.shape
    lk = k.shape[1]
    assert k.shape == v.shape and k.shape[0] == n and k.shape[2] == hidden
    out = torch.empty((n, lq, hidden), dtype=q.dtype, device=q.device)
    check(_lib.load().xml_attention_core(_p(q), hidden, _p(k), hidden, _p(v), hidden, _p(q_mask), _p(k_mask), _p(out), n, lq,
                                         lk, hidden, int(n_heads), dt_of(q), _stream()), "xml_attention_core")
    return out


def conv1d_rows(x, w):
    """nn.Conv1d(1, 1, k, padding=k // 2, bias=False) on the last dimension of x (..., L) f32; w (k,) f32."""
    _req(x, "x", torch.float32); _req(w, "w", torch.float32)
    y = torch.empty_like(x)
    l = x.shape[-1]
    check(_lib.load().xml_conv1d_rows(_p(x), _p(w), _p(y), x.numel() // l, l, w.numel(), _stream()), "xml_conv1d_rows")
    return y


def cross_attention(main_x, main_mask, side_x, side_mask, wq, bq, wkv, bkv, ln_g, ln_b, n_heads):
    """LN(MHA(main, side, side, main_mask (x) side_mask) + main), xml/model_xml.py:369-371."""
    _req(main_x, "main_x"); _req(side_x, "side_x", main_x.dtype)
    _req(main_mask, "main_mask", torch.float32); _req(side_mask, "side_mask", torch.float32)
    _req_w(wq, "wq", main_x); _req_w(wkv, "wkv", main_x)
    for t, nm in ((bq, "bq"), (bkv, "bkv"), (ln_g, "ln_g"), (ln_b, "ln_b")):
        _req(t, nm, torch.float32)
    n, lq, hidden = main_x.shape
    lk = side_x.shape[1]
    lib = _lib.load()
    dt = dt_of(wq)
    y = torch.empty_like(main_x)
    ws = _workspace(lib.xml_cross_attention_workspace_bytes(n, lq, lk, hidden, dt), main_x.device)
    check(lib.xml_cross_attention(_p(main_x), _p(main_mask), _p(side_x), _p(side_mask), _p(wq), _p(bq), _p(wkv),
                                  _p(bkv), _p(ln_g), _p(ln_b), _p(y), n, lq, lk, hidden, n_heads, dt, _p(ws),
                                  ws.numel(), _stream()), "xml_cross_attention")
    return y


def modular_pool(enc, mask, w_m):
    """K5.  enc (N, Lq, H); mask (N, Lq) f32; w_m (n_mod, H) f32 -> (n_mod, N, H)."""
    _req(enc, "enc"); _req(mask, "mask", torch.float32); _req(w_m, "w_m", torch.float32)
    n, lq, hidden = enc.shape
    n_mod = w_m.shape[0]
    out = torch.empty((n_mod, n, hidden), dtype=enc.dtype, device=enc.device)
    check(_lib.load().xml_modular_pool(_p(enc), _p(mask), _p(w_m), _p(out), n, lq, hidden, n_mod, dt_of(enc),
                                       _stream()), "xml_modular_pool")
    return out


def linear(x, w, b=None, relu=False, addend=None):
    """y = x W^T + b [+ addend].  x (..., K), w (N, K) same dtype; addend (..., N) of x's dtype, added in the GEMM epilogue."""
    _req(x, "x"); _req_w(w, "w", x)
    if b is not None:
        _req(b, "b", torch.float32)
    k = x.shape[-1]
    rows = x.numel() // k
    y = torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)
    if addend is not None:
        _req(addend, "addend", x.dtype)
        assert w.dtype is not F16S and addend.numel() == y.numel()
        check(_lib.load().xml_linear_add(_p(x), _p(w), _p(b), _p(addend), _p(y), rows, w.shape[0], k, int(relu), dt_of(x),
                                         _stream()), "xml_linear_add")
        return y
    if w.dtype is F16S:
        lib = _lib.load()
        ws = _workspace(lib.xml_linear_f16s_workspace_bytes(rows, k), x.device)
        check(lib.xml_linear_f16s(_p(x), _p(w), _p(b), _p(y), rows, w.shape[0], k, int(relu), _p(ws), ws.numel(), _stream()),
              "xml_linear_f16s")
        return y
    check(_lib.load().xml_linear(_p(x), _p(w), _p(b), _p(y), rows, w.shape[0], k, int(relu), dt_of(x), _stream()),
          "xml_linear")
    return y


def l2norm_rows(x):
    _req(x, "x")
    d = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.load().xml_l2norm_rows(_p(x), _p(y), x.numel() // d, d, dt_of(x), _stream()), "xml_l2norm_rows")
    return y


def l2norm_rows_eps(x, eps=1e-5):
    """dataset normalisation x / (||x|| + eps) on the device (f32)."""
    _req(x, "x", torch.float32)
    d = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.load().xml_l2norm_rows_eps(_p(x), _p(y), x.numel() // d, d, float(eps), _stream()), "xml_l2norm_rows_eps")
    return y


def ingest_rows(src, row_start, n, lmax, max_len, normalize=True, eps=1e-5, out_dtype=torch.float32):
    """Context collate on the device (xml_ingest_rows): src (rows, d) f32 / f16 = the batch's raw clip rows back to back,
    row_start (n + 1,) int64 -> (features (n, lmax, d) out_dtype [x / (||x|| + eps) per clip, zero padding], mask (n, lmax))."""
    _req(src, "src"); _req(row_start, "row_start", torch.int64)
    assert src.dtype in (torch.float32, torch.float16) and row_start.numel() == n + 1
    d = src.shape[-1]
    out = torch.empty((n, lmax, d), dtype=out_dtype, device=src.device)
    mask = torch.empty((n, lmax), dtype=torch.float32, device=src.device)
    check(_lib.load().xml_ingest_rows(_p(src), dt_of(src), _p(row_start), _p(out), dt_of(out_dtype), _p(mask), int(n), int(lmax),
                                      d, int(max_len), float(eps), int(bool(normalize)), _stream()), "xml_ingest_rows")
    return out, mask


def add_layernorm(a, b, g, beta, out_dtype=None):
    _req(a, "a"); _req(g, "g", torch.float32); _req(beta, "beta", torch.float32)
    out_dtype = out_dtype or (b.dtype if b is not None else a.dtype)
    d = a.shape[-1]
    y = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    check(_lib.load().xml_add_layernorm(_p(a), dt_of(a), _p(b), _p(g), _p(beta), _p(y), a.numel() // d, d,
                                        dt_of(out_dtype), _stream()), "xml_add_layernorm")
    return y


def q2c_scores(qn, cn, mask, out=None, combine=False):
    """K6.  qn (Nq, H) and cn (Nv, Lpad, H) L2-normalised; mask (Nv, Lpad) f32 -> out (Nq, Nv) f32."""
    _req(qn, "qn"); _req(cn, "cn", qn.dtype); _req(mask, "mask", torch.float32)
    nq, hidden = qn.shape
    nv, lpad, h2 = cn.shape
    assert h2 == hidden and tuple(mask.shape) == (nv, lpad)
    if out is None:
        assert not combine
        out = torch.empty((nq, nv), dtype=torch.float32, device=qn.device)
    _req(out, "out", torch.float32)
    check(_lib.load().xml_q2c_scores(_p(qn), _p(cn), _p(mask), _p(out), out.stride(0), nq, nv, lpad, hidden,
                                     int(combine), dt_of(qn), _stream()), "xml_q2c_scores")
    return out


class TiledRows(object):
    """Rows of a (rows, H) operand in K6's slice-major tile layout (xml_q2c_tile_rows): `data` is the flat tiled
    image, `rows` / `hidden` / `dtype` describe the row-major tensor it was made from."""

    def __init__(self, data, rows, hidden, shape, all_valid=False):
        self.data, self.rows, self.hidden, self.shape = data, int(rows), int(hidden), tuple(shape)
        self.dtype, self.device = data.dtype, data.device
        self.all_valid = bool(all_valid)     # every clip mask of the corpus is 1: K6 may skip the masks (5-slot ring)
        self.mask_bits = None                # (Nv, 4) int32: binary clip masks packed 128 bits per video (ragged corpora)
        self.plan = None                     # PackPlan: length-bucketed image (mask_bits are then per wave tile)

    def numel(self):
        return self.data.numel()

    def element_size(self):
        return self.data.element_size()

    def to_rows(self):
        """Back to the row-major tensor of the original shape (tests, the CPU baseline sample)."""
        per = 64 // self.data.element_size()
        t = self.data.view(-1, self.hidden // per, 256, per).permute(0, 2, 1, 3).reshape(-1, self.hidden)
        if self.plan is not None:            # un-bucket: packed row i came from original row row_map[i]
            rm = self.plan.row_map.long()
            out = torch.zeros((self.rows, self.hidden), dtype=t.dtype, device=t.device)
            ok = rm >= 0
            out[rm[ok]] = t[:rm.numel()][ok]
            return out.reshape(self.shape)
        return t[:self.rows].reshape(self.shape).contiguous()


_TILE_PATTERNS = None


def _tile_patterns():
    """Every multiset of video sizes (1..8 blocks of 16 clips) that fits the 16 blocks of a K6 tile: (P, 9) counts per
    size, (P,) blocks used, and per pattern the order in which its videos are laid out -- a sub-multiset that fills the
    left wave tile exactly (8 blocks) goes first when there is one, so that no video straddles the two wave tiles."""
    global _TILE_PATTERNS
    if _TILE_PATTERNS is None:
        pats = []

        def rec(rem, mx, cur):
            if cur:
                pats.append(list(cur))
            for part in range(min(mx, rem), 0, -1):
                cur.append(part)
                rec(rem - part, part, cur)
                cur.pop()
        rec(16, 8, [])
        cnt = np.zeros((len(pats), 9), dtype=np.int64)
        orders = []
        for i, pat in enumerate(pats):
            for x in pat:
                cnt[i, x] += 1
            best = None
            for bits in range(1 << len(pat)):                  # <= 2^16 subsets, once per process
                if sum(x for j, x in enumerate(pat) if bits >> j & 1) == 8:
                    best = bits
                    break
            if best is None:
                orders.append(list(pat))
            else:
                orders.append([x for j, x in enumerate(pat) if best >> j & 1] + [x for j, x in enumerate(pat) if not best >> j & 1])
        _TILE_PATTERNS = (cnt, (cnt * np.arange(9)).sum(1), orders)
    return _TILE_PATTERNS


class PackPlan(object):
    """Packed layout of a ragged corpus for K6 (xml_q2c_scores_packed): every video is padded to a multiple of 16 clips
    ("blocks") and the videos are laid back to back into the 16 blocks of a 256-row tile; a video may straddle the two
    wave tiles (8 blocks each) of a tile, never two tiles.
      row_map  (n_tiles * 256,) int32  original row (video * 128 + clip) of every packed row, -1 = zero row
      slot_ids (2 * n_tiles, 8) int32  per block: id >= 0 last block of video id, -1 the video continues in the next block,
                                       -2 unused, -3 (block 7 of an even wave tile) continues in the next wave tile
      padded_clips                     sum of the padded lengths; n_tiles * 256 = the clip rows K6 executes
    The tiles are filled by a greedy bin packing over the 794 size multisets that fit a tile: fullest pattern first, ties
    to the pattern that draws on the sizes holding most of the remaining blocks (the real TVR lengths, mean 51.4 clips:
    4 871 tiles = the lower bound ceil(blocks / 16); 1.114 x the valid clip rows against 1.31 x for 128 / 64 / 32 buckets)."""

    def __init__(self, masks):
        m = masks[0]
        for o in masks[1:]:
            m = torch.maximum(m, o)
        nv, l = m.shape
        assert l == 128
        dev = m.device
        pos = torch.arange(1, l + 1, device=dev, dtype=torch.float32)
        lens = ((m != 0).float() * pos).amax(1).cpu().numpy().astype(np.int64)      # last valid clip + 1 (one read-back)
        blocks = np.maximum((lens + 15) // 16, 1)
        pat_cnt, pat_fill, pat_order = _tile_patterns()
        counts = np.bincount(blocks, minlength=9).astype(np.int64)
        queues = {s: np.nonzero(blocks == s)[0] for s in range(1, 9)}             # ascending ids per size
        taken = np.zeros(9, dtype=np.int64)
        sizes = np.arange(9)
        seg_tile, seg_start, seg_nb, seg_vid = [], [], [], []
        n_tiles = 0
        while counts.sum() > 0:
            weight = counts * sizes
            score = pat_fill * 1000.0 + (pat_cnt * sizes * (weight / weight.sum())).sum(1) * 50.0
            score = np.where((pat_cnt <= counts).all(1), score, -1.0)
            i = int(score.argmax())
            used = pat_cnt[i]
            rep = max(1, int(min(counts[s] // used[s] for s in range(1, 9) if used[s])) // 4)
            order = np.asarray(pat_order[i], dtype=np.int64)
            start = np.concatenate([[0], np.cumsum(order)[:-1]])
            vid = np.empty((rep, len(order)), dtype=np.int64)
            for s in range(1, 9):
                if used[s]:
                    ids = queues[s][taken[s]:taken[s] + rep * used[s]].reshape(rep, used[s])
                    vid[:, order == s] = ids
                    taken[s] += rep * used[s]
            seg_tile.append(np.repeat(np.arange(n_tiles, n_tiles + rep), len(order)))
            seg_start.append(np.tile(start, rep))
            seg_nb.append(np.tile(order, rep))
            seg_vid.append(vid.reshape(-1))
            counts = counts - used * rep
            n_tiles += rep
        seg_tile, seg_start = np.concatenate(seg_tile), np.concatenate(seg_start)
        seg_nb, seg_vid = np.concatenate(seg_nb), np.concatenate(seg_vid)
        first = seg_tile * 16 + seg_start                                          # first block of every video
        codes = np.full(n_tiles * 16, -2, dtype=np.int64)
        seg_of_block = np.repeat(np.arange(len(seg_nb)), seg_nb)
        blk = first[seg_of_block] + (np.arange(seg_nb.sum()) - np.repeat(np.cumsum(seg_nb) - seg_nb, seg_nb))
        codes[blk] = -1
        codes[first + seg_nb - 1] = seg_vid
        straddles = (seg_start < 8) & (seg_start + seg_nb > 8)
        codes[seg_tile[straddles] * 16 + 7] = -3
        rows = np.full(n_tiles * 256, -1, dtype=np.int64)
        seg_of_row = np.repeat(np.arange(len(seg_nb)), seg_nb * 16)
        within = np.arange(seg_nb.sum() * 16) - np.repeat(np.cumsum(seg_nb * 16) - seg_nb * 16, seg_nb * 16)
        rows[first[seg_of_row] * 16 + within] = seg_vid[seg_of_row] * 128 + within
        self.row_map = torch.from_numpy(rows.astype(np.int32)).to(dev).contiguous()
        self.slot_ids = torch.from_numpy(codes.astype(np.int32).reshape(2 * n_tiles, 8)).to(dev).contiguous()
        self.n_tiles = n_tiles
        self.n_videos = nv
        self.n_straddles = int(straddles.sum())
        self.padded_clips = int(seg_nb.sum() * 16)

    def mask_bits(self, mask):
        """(nv, 128) binary f32 mask -> (2 * n_tiles, 4) int32: the masks of every wave tile's 128 packed columns."""
        flat = torch.cat([(mask != 0).reshape(-1), torch.zeros(1, dtype=torch.bool, device=mask.device)])
        rm = self.row_map.long()
        v = flat[torch.where(rm >= 0, rm, torch.full_like(rm, flat.numel() - 1))]
        w = (v.view(-1, 4, 32).to(torch.int64) << torch.arange(32, device=mask.device, dtype=torch.int64)).sum(-1)
        return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def q2c_pack_plan(masks):
    """PackPlan for the corpus with these per-modality (Nv, 128) clip masks, or None when packing does not apply
    (non-binary masks, full-length corpus, fewer than 5 % of the clip rows to save, XML_Q2C_NO_BUCKETS=1 for A/B runs)."""
    if os.environ.get("XML_Q2C_NO_BUCKETS") or os.environ.get("XML_Q2C_KEEP_MASKS") or os.environ.get("XML_Q2C_ROW_MAJOR"):
        return None
    for m in masks:
        if m.shape[1] != 128 or not bool(((m == 0) | (m == 1)).all()):
            return None
    if all(bool((m == 1).all()) for m in masks):
        return None                           # full-length corpus: the mask-free kernel on the plain tiles
    plan = PackPlan(masks)
    # the packed kernel runs a four-slot ring and a per-video epilogue: it has to save rows to pay (plain: 2 videos per tile)
    return plan if plan.n_tiles * 2 <= 0.95 * ((plan.n_videos + 1) // 2 * 2) else None


def q2c_tiled_ok(lpad, hidden, dtype):
    return dtype in _DT and bool(_lib.load().xml_q2c_tiled_ok(int(lpad), int(hidden), dt_of(dtype)))


def q2c_tile_rows(x):
    """(..., H) contiguous rows -> TiledRows (K6 operand layout)."""
    _req(x, "x")
    hidden = x.shape[-1]
    rows = x.numel() // hidden
    nbytes = _lib.load().xml_q2c_tiled_bytes(rows, hidden, dt_of(x))
    data = torch.empty(nbytes // x.element_size(), dtype=x.dtype, device=x.device)
    check(_lib.load().xml_q2c_tile_rows(_p(x), _p(data), rows, hidden, dt_of(x), _stream()), "xml_q2c_tile_rows")
    return TiledRows(data, rows, hidden, x.shape)


def q2c_tiled_numel(rows, hidden, dtype):
    """Elements of the K6 tile image of (rows, hidden) rows of `dtype`, 0 when the tiled kernel does not take the shape."""
    if not q2c_tiled_ok(128, hidden, dtype) or os.environ.get("XML_Q2C_ROW_MAJOR"):
        return 0
    return int(_lib.load().xml_q2c_tiled_bytes(rows, hidden, _DT[dtype])) // torch.empty(0, dtype=dtype).element_size()


def pack_q2c_corpus(feat1n, mask=None, plan=None, normalize=False, out=None):
    """Resident form of the similarity operand: slice-major tiles when the persistent kernel takes it, else as is.
    mask (Nv, Lpad): if every entry is 1 (full-length videos) the tiles are marked all_valid and K6 skips the masks.
    plan (q2c_pack_plan over ALL modalities' masks): the length-bucketed image instead.
    normalize: feat1n holds the UN-normalised clip features; F.normalize runs inside the tiling pass
    (xml_q2c_tile_rows_l2norm: bitwise the values of l2norm_rows followed by the tiling, one pass over the index less)."""
    lib = _lib.load()
    fused = normalize and lib.xml_q2c_tile_rows_l2norm_ok(feat1n.shape[2], dt_of(feat1n)) \
        and q2c_tiled_ok(feat1n.shape[1], feat1n.shape[2], feat1n.dtype) and not os.environ.get("XML_Q2C_ROW_MAJOR")
    if normalize and not fused:
        feat1n = l2norm_rows(feat1n)
    if plan is not None and q2c_tiled_ok(feat1n.shape[1], feat1n.shape[2], feat1n.dtype):
        _req(feat1n, "feat1n")
        nv, lpad, hidden = feat1n.shape
        rows_packed = plan.n_tiles * 256
        data = torch.empty(rows_packed * hidden, dtype=feat1n.dtype, device=feat1n.device)
        if fused:
            check(lib.xml_q2c_tile_rows_l2norm(_p(feat1n), _p(plan.row_map), _p(data), nv * lpad, rows_packed, hidden,
                                               dt_of(feat1n), _stream()), "xml_q2c_tile_rows_l2norm")
        else:
            check(lib.xml_q2c_tile_rows_gather(_p(feat1n), _p(plan.row_map), _p(data), rows_packed, hidden,
                                               dt_of(feat1n), _stream()), "xml_q2c_tile_rows_gather")
        t = TiledRows(data, nv * lpad, hidden, feat1n.shape)
        t.plan = plan
        t.mask_bits = plan.mask_bits(mask)
        return t
    if q2c_tiled_ok(feat1n.shape[1], feat1n.shape[2], feat1n.dtype) and not os.environ.get("XML_Q2C_ROW_MAJOR"):
        if fused:
            hidden = feat1n.shape[-1]
            rows = feat1n.numel() // hidden
            nbytes = lib.xml_q2c_tiled_bytes(rows, hidden, dt_of(feat1n))
            if out is not None:            # caller-owned tile buffer (inference.IndexStorage)
                _req(out, "out", feat1n.dtype)
                assert out.numel() == nbytes // feat1n.element_size()
                data = out
            else:
                data = torch.empty(nbytes // feat1n.element_size(), dtype=feat1n.dtype, device=feat1n.device)
            check(lib.xml_q2c_tile_rows_l2norm(_p(feat1n), None, _p(data), rows, nbytes // (hidden * feat1n.element_size()),
                                               hidden, dt_of(feat1n), _stream()), "xml_q2c_tile_rows_l2norm")
            t = TiledRows(data, rows, hidden, feat1n.shape)
        else:
            t = q2c_tile_rows(feat1n)         # (XML_Q2C_ROW_MAJOR=1: keep rows, for A/B measurements)
        keep = os.environ.get("XML_Q2C_KEEP_MASKS")      # 1: float masks (4-slot kernel), for A/B measurements
        t.all_valid = mask is not None and bool((mask == 1).all()) and not keep
        if mask is not None and not t.all_valid and not keep and mask.shape[1] == 128 \
                and bool(((mask == 0) | (mask == 1)).all()):
            w = (mask.view(-1, 4, 32) != 0).to(torch.int64) << torch.arange(32, device=mask.device, dtype=torch.int64)
            w = w.sum(-1)
            t.mask_bits = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
        return t
    return feat1n


def q2c_tile_rows_l2norm(x):
    """F.normalize(x, dim=-1) + the K6 tile layout in ONE pass (xml_q2c_tile_rows_l2norm; bitwise l2norm_rows then
    q2c_tile_rows), or None when the fused pass does not take the shape."""
    _req(x, "x")
    lib = _lib.load()
    hidden = x.shape[-1]
    if not lib.xml_q2c_tile_rows_l2norm_ok(hidden, dt_of(x)):
        return None
    rows = x.numel() // hidden
    nbytes = lib.xml_q2c_tiled_bytes(rows, hidden, dt_of(x))
    data = torch.empty(nbytes // x.element_size(), dtype=x.dtype, device=x.device)
    check(lib.xml_q2c_tile_rows_l2norm(_p(x), None, _p(data), rows, nbytes // (hidden * x.element_size()), hidden, dt_of(x),
                                       _stream()), "xml_q2c_tile_rows_l2norm")
    return TiledRows(data, rows, hidden, x.shape)


_pair_maps = {}


def q2c_tile_rows_l2norm_pair(q0, q1):
    """q2c_tile_rows_l2norm of BOTH modalities' query vectors in ONE launch when they are the two halves of one contiguous
    (2, nq, H) tensor -- what xml_modular_pool returns: the tiled image of 2 R rows (R = nq rounded up to the 256-row tile) IS
    the two modalities' images back to back, and a row map (cached per nq) sends source row m nq + i to destination row m R + i.
    Same values as two launches; one kernel fewer in a 50-query batch's chain.  None when the layout does not match."""
    if q0.shape != q1.shape or q0.dtype != q1.dtype or not (q0.is_contiguous() and q1.is_contiguous()):
        return None
    nq, hidden = q0.shape
    if q1.data_ptr() != q0.data_ptr() + nq * hidden * q0.element_size():
        return None
    lib = _lib.load()
    if not lib.xml_q2c_tile_rows_l2norm_ok(hidden, dt_of(q0)):
        return None
    r = (nq + 255) // 256 * 256
    key = (nq, str(q0.device))
    rmap = _pair_maps.get(key)
    if rmap is None:
        i = torch.arange(2 * r, dtype=torch.int32, device=q0.device)
        m, j = i // r, i % r
        rmap = torch.where(j < nq, m * nq + j, torch.full_like(i, -1)).contiguous()
        _pair_maps[key] = rmap
    es = q0.element_size()
    half = lib.xml_q2c_tiled_bytes(nq, hidden, dt_of(q0)) // es
    data = torch.empty(2 * half, dtype=q0.dtype, device=q0.device)
    check(lib.xml_q2c_tile_rows_l2norm(_p(q0), _p(rmap), _p(data), 2 * nq, 2 * r, hidden, dt_of(q0), _stream()),
          "xml_q2c_tile_rows_l2norm")
    return [TiledRows(data[:half], nq, hidden, q0.shape), TiledRows(data[half:], nq, hidden, q0.shape)]


def q2c_scores_fused(qn, cn, masks, out=None, normalize_q=False):
    """K6 for all modalities in one launch.  qn / cn / masks: lists (len 1 or 2) of (Nq,H), (Nv,Lpad,H), (Nv,Lpad) f32.
    out (Nq, Nv) f32 = mean over modalities of the masked max-over-clips cosine.
    cn[m] may be TiledRows (pack_q2c_corpus): the queries are tiled here and the tiled entry runs.
    normalize_q: qn holds the UN-normalised modular query vectors; on the tiled path F.normalize runs inside the tiling pass
    (one launch per modality instead of two -- 10 us of a 350 us 50-query batch), elsewhere as l2norm_rows first."""
    n_mod = len(qn)
    qt_pre = None
    if normalize_q:
        if isinstance(cn[0], TiledRows):
            qt_pre = q2c_tile_rows_l2norm_pair(qn[0], qn[1]) if n_mod == 2 else None      # (one launch for both modalities)
            if qt_pre is None:
                qt_pre = [q2c_tile_rows_l2norm(q.contiguous()) for q in qn]
            if any(t is None for t in qt_pre):
                qt_pre = None
        if qt_pre is None:
            qn = [l2norm_rows(q.contiguous()) for q in qn]
    if isinstance(cn[0], TiledRows):
        nq, hidden = qn[0].shape
        nv, lpad = masks[0].shape
        for m in range(n_mod):
            _req(qn[m], "qn"); _req(masks[m], "mask", torch.float32)
            assert isinstance(cn[m], TiledRows) and cn[m].shape == (nv, lpad, hidden) and cn[m].dtype == qn[m].dtype
        qt = qt_pre if qt_pre is not None else [q2c_tile_rows(q) for q in qn]
        if out is None:
            out = torch.empty((nq, nv), dtype=torch.float32, device=qn[0].device)
        _req(out, "out", torch.float32)
        j = 1 if n_mod > 1 else 0
        plan = cn[0].plan
        if plan is not None:                  # packed image: one masked maximum per video, original columns
            assert all(c.plan is plan for c in cn[:n_mod]) and plan.n_videos == nv
            check(_lib.load().xml_q2c_scores_packed(n_mod, _p(qt[0].data), _p(cn[0].data), _p(qt[j].data),
                                                    _p(cn[j].data), _p(out), out.stride(0), nq, plan.n_tiles,
                                                    _p(plan.slot_ids), _p(cn[0].mask_bits),
                                                    _p(cn[j].mask_bits), hidden, dt_of(qn[0]), _stream()),
                  "xml_q2c_scores_packed")
            return out
        bits = [c.mask_bits for c in cn[:n_mod]]
        if all(c.all_valid for c in cn[:n_mod]):
            mode, bits = 1, [None, None]
        elif all(b is not None or c.all_valid for b, c in zip(bits, cn[:n_mod])):
            mode = 2       # a fully valid modality next to a ragged one: all-ones bit rows
            bits = [b if b is not None else torch.full((nv, 4), -1, dtype=torch.int32, device=out.device) for b in bits]
        else:
            mode, bits = 0, [None, None]
        bits = (bits + [None])[:2] if n_mod == 1 else bits
        check(_lib.load().xml_q2c_scores_tiled(n_mod, _p(qt[0].data), _p(cn[0].data), _p(masks[0]), _p(qt[j].data),
                                               _p(cn[j].data), _p(masks[j]), _p(out), out.stride(0), nq, nv, lpad,
                                               hidden, dt_of(qn[0]), mode, _p(bits[0]), _p(bits[j]), _stream()),
              "xml_q2c_scores_tiled")
        return out
    for m in range(n_mod):
        _req(qn[m], "qn"); _req(cn[m], "cn", qn[m].dtype); _req(masks[m], "mask", torch.float32)
        assert qn[m].shape == qn[0].shape and cn[m].shape == cn[0].shape and tuple(masks[m].shape) == tuple(cn[0].shape[:2])
    nq, hidden = qn[0].shape
    nv, lpad, _ = cn[0].shape
    if out is None:
        out = torch.empty((nq, nv), dtype=torch.float32, device=qn[0].device)
    _req(out, "out", torch.float32)
    j = 1 if n_mod > 1 else 0
    check(_lib.load().xml_q2c_scores_fused(n_mod, _p(qn[0]), _p(cn[0]), _p(masks[0]), _p(qn[j]), _p(cn[j]),
                                           _p(masks[j]), _p(out), out.stride(0), nq, nv, lpad, hidden, dt_of(qn[0]),
                                           _stream()), "xml_q2c_scores_fused")
    return out


def topk_rows(scores, k, alpha=0.0, idx_in=None):
    """K8.  scores (rows, n) f32 -> (values (rows, k) f32 [exp(alpha*s) if alpha], indices (rows, k) int32)."""
    _req(scores, "scores", torch.float32)
    rows, n = scores.shape
    if idx_in is not None:
        _req(idx_in, "idx_in", torch.int32)
        assert idx_in.shape == scores.shape
    vals = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((rows, k), dtype=torch.int32, device=scores.device)
    lib = _lib.load()
    ws = _workspace(lib.xml_topk_rows_workspace_bytes(rows, n, k), scores.device)      # header contract: caller's scratch
    check(lib.xml_topk_rows(_p(scores), scores.stride(0), _p(idx_in), _p(vals), _p(idx), rows, n, k,
                            float(alpha), _p(ws), ws.numel(), _stream()), "xml_topk_rows")
    return vals, idx


def merge_shard_topk(recv_score, recv_id, k, alpha=0.0):
    """The owner's merge of a sharded pass behind the wire (xml_merge_shard_topk): recv_score / recv_id (world, rows, c)
    -- every shard's local top-c of the same query rows, source rank major -> (values (rows, k), ids (rows, k)) ordered
    like topk_rows (score desc, id asc)."""
    _req(recv_score, "recv_score", torch.float32); _req(recv_id, "recv_id", torch.int32)
    world, rows, c = recv_score.shape
    assert recv_id.shape == recv_score.shape
    vals = torch.empty((rows, k), dtype=torch.float32, device=recv_score.device)
    idx = torch.empty((rows, k), dtype=torch.int32, device=recv_score.device)
    lib = _lib.load()
    ws = _workspace(lib.xml_merge_shard_topk_workspace_bytes(world, rows, c), recv_score.device)
    check(lib.xml_merge_shard_topk(_p(recv_score), _p(recv_id), world, rows, c, int(k), float(alpha), _p(vals), _p(idx), _p(ws),
                                   ws.numel(), _stream()), "xml_merge_shard_topk")
    return vals, idx


Q2C_NORMALIZE_Q = True  # q2c_scores_fused(normalize_q=True) exists
RAGGED_ROWS = True     # convse_rerank(vid_len=) / moment_topk(pair_vid=, vid_len=) exist (xml_convse_rerank_ex, xml_moment_topk_ex)
MOMENT_SUMM = 8        # XML_MOMENT_SUMM


def convse_rerank(q_lin, feat2, masks, pair_vid, conv_w, l_ref, merged, ksize, softmax=True, zero_skipped=True,
                  pair_w=None, band=None, vid_len=None, out=None):
    """K7.  q_lin / feat2 / masks: lists over modalities (len 1 or 2).
    q_lin[m] (Nq, H); feat2[m] (Nv, Lpad, H); masks[m] (Nv, Lpad) f32; pair_vid (Nq, K) int32;
    conv_w flat f32 [st filters..., ed filters...].  Returns st, ed (Nq, K, Lpad) f32.
    zero_skipped=False leaves the rows of skipped pairs (pair_vid < 0) unwritten -- for callers that never read them
    (the sharded pass: K9 skips pairs of weight 0).
    band = (min_l, max_l) [+ pair_w (Nq, K) f32, the video weights]: also returns summ (Nq, K, 8) f32, the 8 largest banded
    row maxima of every pair (xml_convse_rerank_ex) -- hand it to moment_topk(..., summ=summ), which then reads the rows once.
    vid_len (Nv,) int32: ragged corpora -- valid clips per video; entries l >= vid_len[v] of st / ed are left UNWRITTEN (exact
    zeros of the masked softmax) for moment_topk(..., pair_vid=pair_vid, vid_len=vid_len), which does not read them."""
    n_mod = len(q_lin)
    for m in range(n_mod):
        _req(q_lin[m], "q_lin"); _req(feat2[m], "feat2", q_lin[m].dtype); _req(masks[m], "mask", torch.float32)
    _req(pair_vid, "pair_vid", torch.int32); _req(conv_w, "conv_w", torch.float32)
    nq, hidden = q_lin[0].shape
    nv, lpad, _ = feat2[0].shape
    kpairs = pair_vid.shape[1]
    d = ConvseDesc(nq=nq, nv=nv, kpairs=kpairs, lpad=lpad, l_ref=int(l_ref), hidden=hidden, n_mod=n_mod,
                   merged=int(merged), ksize=int(ksize), softmax=int(bool(softmax)) | (0 if zero_skipped else 2),
                   dt=dt_of(q_lin[0]))
    n_conv = 1 if merged else n_mod
    assert conv_w.numel() == 2 * n_conv * ksize
    lib = _lib.load()
    if out is not None:            # (st, ed) destination tensors of the caller
        st, ed = out
        _req(st, "st", torch.float32); _req(ed, "ed", torch.float32)
        assert tuple(st.shape) == tuple(ed.shape) == (nq, kpairs, lpad)
    else:
        st = torch.empty((nq, kpairs, lpad), dtype=torch.float32, device=pair_vid.device)
        ed = torch.empty_like(st)
    ws = _workspace(lib.xml_convse_rerank_workspace_bytes(ctypes.byref(d)), pair_vid.device)
    if band is not None or vid_len is not None:
        assert softmax, "candidate summaries / unwritten zero tails are those of probabilities"
        if pair_w is not None:
            _req(pair_w, "pair_w", torch.float32)
            assert tuple(pair_w.shape) == (nq, kpairs)
        if vid_len is not None:
            _req(vid_len, "vid_len", torch.int32)
            assert vid_len.numel() == nv
        summ = torch.empty((nq, kpairs, MOMENT_SUMM), dtype=torch.float32, device=pair_vid.device) if band is not None else None
        band = band if band is not None else (0, 1)
        one, sp = n_mod == 1, q_lin[0].dtype is F16S
        inv = lambda t: _p(t.inv) if sp else None          # noqa: E731
        check(lib.xml_convse_rerank_ex(ctypes.byref(d), _p(q_lin[0]), None if one else _p(q_lin[1]), inv(q_lin[0]),
                                       None if one else inv(q_lin[1]), _p(feat2[0]), None if one else _p(feat2[1]),
                                       inv(feat2[0]), None if one else inv(feat2[1]), _p(masks[0]),
                                       None if one else _p(masks[1]), _p(pair_vid), _p(conv_w), _p(pair_w), int(band[0]),
                                       int(band[1]), _p(vid_len), _p(st), _p(ed), _p(summ), _p(ws), ws.numel(), _stream()),
              "xml_convse_rerank_ex")
        return (st, ed, summ) if summ is not None else (st, ed)
    if q_lin[0].dtype is F16S:       # split-f16 rows on both sides: f32-grade similarities on the 16-bit pipe
        one = n_mod == 1
        check(lib.xml_convse_rerank_f16s(ctypes.byref(d), _p(q_lin[0]), None if one else _p(q_lin[1]), _p(q_lin[0].inv),
                                         None if one else _p(q_lin[1].inv), _p(feat2[0]), None if one else _p(feat2[1]),
                                         _p(feat2[0].inv), None if one else _p(feat2[1].inv), _p(masks[0]),
                                         None if one else _p(masks[1]), _p(pair_vid), _p(conv_w), _p(st), _p(ed), _p(ws),
                                         ws.numel(), _stream()), "xml_convse_rerank_f16s")
        return st, ed
    check(lib.xml_convse_rerank(ctypes.byref(d), _p(q_lin[0]), _p(q_lin[1]) if n_mod > 1 else None, _p(feat2[0]),
                                _p(feat2[1]) if n_mod > 1 else None, _p(masks[0]),
                                _p(masks[1]) if n_mod > 1 else None, _p(pair_vid), _p(conv_w), _p(st), _p(ed),
                                _p(ws), ws.numel(), _stream()), "xml_convse_rerank")
    return st, ed


def moment_topk(st, ed, w, l_ref, min_l, max_l, n_out, summ=None, pair_vid=None, vid_len=None):
    """K9/K10.  st, ed (Nq, K, Lpad) f32 probabilities; w (Nq, K) f32 or None.
    summ (Nq, K, 8) f32: the candidate summaries convse_rerank(..., band=(min_l, max_l), pair_w=w) returned for THESE rows.
    pair_vid (Nq, K) int32 + vid_len (Nv,) int32: the rows came from convse_rerank(..., vid_len=vid_len) -- entries beyond a
    video's valid length were not written and are not read.
    Returns (scores (Nq, n_out) f32 desc, flat (Nq, n_out) int32 = (r*l_ref + i)*l_ref + j, -1 = empty)."""
    _req(st, "st", torch.float32); _req(ed, "ed", torch.float32)
    if w is not None:
        _req(w, "w", torch.float32)
    nq, kpairs, lpad = st.shape
    if summ is not None:
        _req(summ, "summ", torch.float32)
        assert tuple(summ.shape) == (nq, kpairs, MOMENT_SUMM)
    assert (pair_vid is None) == (vid_len is None)
    if vid_len is not None:
        _req(pair_vid, "pair_vid", torch.int32); _req(vid_len, "vid_len", torch.int32)
        assert tuple(pair_vid.shape) == (nq, kpairs)
    sc = torch.empty((nq, n_out), dtype=torch.float32, device=st.device)
    fl = torch.empty((nq, n_out), dtype=torch.int32, device=st.device)
    check(_lib.load().xml_moment_topk_ex(_p(st), _p(ed), _p(w), _p(summ), _p(pair_vid), _p(vid_len), _p(sc), _p(fl), nq, kpairs,
                                         lpad, int(l_ref), int(min_l), int(max_l), int(n_out), _stream()), "xml_moment_topk_ex")
    return sc, fl


def moments_decode(scores, flat=None, top_idx=None, row_vid=None, meta2vid=None, l_ref=0, clip_length=1.5, seconds=True,
                   n=None, out=None, out_count=None):
    """K10.  (flat (Nq, n) int32, scores (Nq, n) f32) -> records (Nq, n, 4) int32 = xml_moment {vid i32, st f32, ed f32,
    score f32} (view the host copy as results.MOMENT_DTYPE) and count (Nq,) int32.
    top_idx (Nq, K) int32: video of local rank r; row_vid (Nq,) int32: the one video of each query (SVMR); meta2vid (Nv,)
    int32: meta index -> video2idx value.  flat=None: the VR list of top_idx[:, :n] / scores[:, :n].
    out / out_count: write into these rows of a larger result buffer ((Nq, >= n, 4) int32 / (Nq,) int32 views)."""
    _req(scores, "scores", torch.float32)
    nq = scores.shape[0]
    n = scores.shape[1] if n is None else int(n)
    src = flat if flat is not None else top_idx
    _req(src, "flat / top_idx", torch.int32)
    assert src.shape[0] == nq and src.stride(0) == scores.stride(0) and n <= scores.shape[1]
    k = 0
    if flat is not None and top_idx is not None:
        _req(top_idx, "top_idx", torch.int32)
        k = top_idx.shape[1]
    for t, name in ((row_vid, "row_vid"), (meta2vid, "meta2vid")):
        if t is not None:
            _req(t, name, torch.int32)
    if out is None:
        out = torch.empty((nq, n, 4), dtype=torch.int32, device=scores.device)
    if out_count is None:
        out_count = torch.empty((nq,), dtype=torch.int32, device=scores.device)
    assert out.dtype == torch.int32 and out.shape[0] == nq and out.shape[2] == 4 and out.stride(2) == 1 and out.stride(1) == 4
    assert out_count.dtype == torch.int32 and out_count.is_contiguous() and out_count.numel() == nq
    check(_lib.load().xml_moments_decode(_p(flat), _p(scores), _p(top_idx), _p(row_vid), _p(meta2vid), nq, n,
                                         scores.stride(0), k, int(l_ref), float(clip_length), 1 if seconds else 0,
                                         _p(out), out.stride(0) // 4, _p(out_count), _stream()), "xml_moments_decode")
    return out, out_count


# ---- exact-rank mode (bf16 K6 as a filter in front of f32 scores; include/xmlhip.h "Exact-rank mode") --------------------
def round_bf16_rows_err(y):
    """y (..., d) f32 L2-normalised rows -> (yb bf16 = rne(y), err (...) f32 = ||y - yb||_2 per row)."""
    _req(y, "y", torch.float32)
    d = y.shape[-1]
    yb = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
    err = torch.empty(y.shape[:-1], dtype=torch.float32, device=y.device)
    check(_lib.load().xml_round_bf16_rows_err(_p(y), _p(yb), _p(err), y.numel() // d, d, _stream()),
          "xml_round_bf16_rows_err")
    return yb, err


def q2c_rescore(qn, cn, masks, pair_vid):
    """Video-level scores of the listed pairs only.  qn / cn / masks: lists over modalities of (Nq, H), (Nv, Lpad, H)
    row-major L2-normalised, (Nv, Lpad) f32; pair_vid (Nq, K) int32 -> (Nq, K) f32 (-inf for ids outside [0, Nv))."""
    n_mod = len(qn)
    for m in range(n_mod):
        _req(qn[m], "qn"); _req(cn[m], "cn", qn[m].dtype); _req(masks[m], "mask", torch.float32)
        assert cn[m].shape == cn[0].shape and qn[m].shape == qn[0].shape and tuple(masks[m].shape) == tuple(cn[0].shape[:2])
    _req(pair_vid, "pair_vid", torch.int32)
    nq, hidden = qn[0].shape
    nv, lpad, _ = cn[0].shape
    kp = pair_vid.shape[1]
    assert pair_vid.shape[0] == nq
    out = torch.empty((nq, kp), dtype=torch.float32, device=pair_vid.device)
    lib = _lib.load()
    ws = _workspace(lib.xml_q2c_rescore_workspace_bytes(nq, nv, kp), pair_vid.device)
    j = 1 if n_mod > 1 else 0
    check(lib.xml_q2c_rescore(n_mod, _p(qn[0]), _p(qn[j]), _p(cn[0]), _p(cn[j]), _p(masks[0]), _p(masks[j]), _p(pair_vid),
                              _p(out), nq, nv, kp, lpad, hidden, dt_of(qn[0]), _p(ws), ws.numel(), _stream()),
          "xml_q2c_rescore")
    return out


def exact_certificate(filter_scores, top_val, eq, ec, slack, alpha, outside):
    """Per-query certificate of the exact-rank filter.  filter_scores (Nq, M) f32 desc (bf16 pass); top_val (Nq, K) f32 desc,
    raw re-scored values -- turned into exp(alpha * s) IN PLACE; eq: list of (Nq,) f32 per modality; ec: list of floats.
    Returns (fail (Nq,) int32, eps (Nq,) f32, thr (Nq,) f32 = raw T_k - eps, n_fail (1,) int32 device tensor)."""
    _req(filter_scores, "filter_scores", torch.float32); _req(top_val, "top_val", torch.float32)
    n_mod = len(eq)
    for e in eq:
        _req(e, "eq", torch.float32)
    nq, m = filter_scores.shape
    k = top_val.shape[1]
    dev = top_val.device
    fail = torch.empty((nq,), dtype=torch.int32, device=dev)
    eps = torch.empty((nq,), dtype=torch.float32, device=dev)
    thr = torch.empty((nq,), dtype=torch.float32, device=dev)
    n_fail = torch.zeros((1,), dtype=torch.int32, device=dev)
    j = 1 if n_mod > 1 else 0
    check(_lib.load().xml_exact_certificate(_p(filter_scores), m, _p(top_val), k, _p(eq[0]), _p(eq[j]), float(ec[0]),
                                            float(ec[j]), n_mod, float(slack), float(alpha), int(bool(outside)), _p(fail),
                                            _p(eps), _p(thr), _p(n_fail), nq, _stream()), "xml_exact_certificate")
    return fail, eps, thr, n_fail


# ---- packed variable-length sequences (the query encoder without its padding rows) -------------------------------------
def pack_plan(mask, rows=None):
    """Packing plan of a padded batch.  mask (n, lq) f32 -> (cu_seqlens (n + 1,) int32, src_row (n * lq,) int32, rows);
    rows = -1 when some mask row is not a non-empty prefix of ones.  (One 4-byte read-back: the launch shapes of the
    packed kernels depend on rows.)
    rows (host int): the caller KNOWS the number of valid tokens -- it built the masks on the host, like the reference's
    collate (start_end_dataset.py:346-370) -- and vouches that every mask row is a non-empty prefix of ones: no read-back, the
    pass stays asynchronous (the read-back is a host synchronisation: ~0.6 ms of idle device per 10 000-query pass)."""
    _req(mask, "mask", torch.float32)
    n, lq = mask.shape
    cu = torch.empty(n + 1, dtype=torch.int32, device=mask.device)
    src = torch.empty(n * lq, dtype=torch.int32, device=mask.device)
    status = torch.empty(2, dtype=torch.int32, device=mask.device)
    check(_lib.load().xml_pack_plan(_p(mask), n, lq, _p(cu), _p(src), _p(status), _stream()), "xml_pack_plan")
    if rows is not None:
        rows = int(rows)
        if not n <= rows <= n * lq:
            raise ValueError("pack_plan: %d valid tokens cannot be right for %d sequences of <= %d" % (rows, n, lq))
        return cu, src, rows
    return cu, src, int(status[0].item())


def linear_ln_relu_pos_packed(x, src_row, rows, lq, ln_in_g, ln_in_b, w, b, pos, ln_pos_g, ln_pos_b):
    """K1+K2 on packed tokens: x (n * lq, d_in) f32 or w.dtype is the PADDED batch, packed token i = x[src_row[i]] with
    positional row pos[src_row[i] % lq] -> (rows, hidden) w.dtype."""
    _req(x, "x"); _req(src_row, "src_row", torch.int32); _req(w, "w"); _req(pos, "pos", act_dtype(w.dtype))
    for t, nm in ((ln_in_g, "ln_in_g"), (ln_in_b, "ln_in_b"), (b, "b"), (ln_pos_g, "ln_pos_g"), (ln_pos_b, "ln_pos_b")):
        _req(t, nm, torch.float32)
    d_in, hidden = x.shape[1], w.shape[0]
    assert w.shape[1] == d_in and pos.shape[0] >= lq and pos.shape[1] == hidden and src_row.numel() >= rows
    lib = _lib.load()
    dt = dt_of(w)
    y = torch.empty((rows, hidden), dtype=act_dtype(w.dtype), device=x.device)
    ws = _workspace(lib.xml_linear_ln_relu_pos_packed_workspace_bytes(rows, d_in, hidden, dt), x.device)
    check(lib.xml_linear_ln_relu_pos_packed(_p(x), dt_of(x), _p(src_row), int(lq), _p(ln_in_g), _p(ln_in_b), _p(w), _p(b),
                                            _p(pos), _p(ln_pos_g), _p(ln_pos_b), _p(y), rows, d_in, hidden, dt, _p(ws),
                                            ws.numel(), _stream()), "xml_linear_ln_relu_pos_packed")
    return y


def attention_block_varlen(x, cu_seqlens, n, max_len, wqkv, bqkv, wo, bo, ln_g, ln_b, n_heads):
    """K3+K4 on packed tokens.  x (rows, H); cu_seqlens (n + 1,) int32 -> (rows, H)."""
    _req(x, "x"); _req(cu_seqlens, "cu_seqlens", torch.int32); _req_w(wqkv, "wqkv", x); _req_w(wo, "wo", x)
    for t, nm in ((bqkv, "bqkv"), (bo, "bo"), (ln_g, "ln_g"), (ln_b, "ln_b")):
        _req(t, nm, torch.float32)
    rows, hidden = x.shape
    assert cu_seqlens.numel() == n + 1
    lib = _lib.load()
    dt = dt_of(wqkv)
    y = torch.empty_like(x)
    ws = _workspace(lib.xml_attention_block_varlen_workspace_bytes(rows, hidden, dt), x.device)
    check(lib.xml_attention_block_varlen(_p(x), _p(cu_seqlens), _p(wqkv), _p(bqkv), _p(wo), _p(bo), _p(ln_g), _p(ln_b), _p(y),
                                         rows, n, int(max_len), hidden, n_heads, dt, _p(ws), ws.numel(), _stream()),
          "xml_attention_block_varlen")
    return y


def modular_pool_varlen(enc, cu_seqlens, n, max_len, w_m):
    """K5 on packed tokens.  enc (rows, H); w_m (n_mod, H) f32 -> (n_mod, n, H)."""
    _req(enc, "enc"); _req(cu_seqlens, "cu_seqlens", torch.int32); _req(w_m, "w_m", torch.float32)
    hidden = enc.shape[1]
    n_mod = w_m.shape[0]
    out = torch.empty((n_mod, n, hidden), dtype=enc.dtype, device=enc.device)
    check(_lib.load().xml_modular_pool_varlen(_p(enc), _p(cu_seqlens), _p(w_m), _p(out), n, int(max_len), hidden, n_mod,
                                              dt_of(enc), _stream()), "xml_modular_pool_varlen")
    return out


def select_ge_rows(scores, thr, cap=None):
    """Columns of every row of scores (rows, n) f32 that reach thr (rows,) f32.  cap None -> counts (rows,) int32;
    else -> (idx (rows, cap) int32 filled with -1 beyond each row's count, counts)."""
    _req(scores, "scores", torch.float32); _req(thr, "thr", torch.float32)
    rows, n = scores.shape
    cnt = torch.empty((rows,), dtype=torch.int32, device=scores.device)
    idx = None
    if cap is not None:
        idx = torch.full((rows, int(cap)), -1, dtype=torch.int32, device=scores.device)
    check(_lib.load().xml_select_ge_rows(_p(scores), scores.stride(0), _p(thr), _p(idx), int(cap or 0), _p(cnt), rows, n,
                                         _stream()), "xml_select_ge_rows")
    return cnt if idx is None else (idx, cnt)


# ---- split-f16 forms (XML_F16S; include/xmlhip.h "Exact-rank mode on the 16-bit pipe") ------------------------------------
def pack_weights_f16s(w_f32):
    """(n, k) f32 weight -> SplitWeight ((n, 3k) f16 [hi | hi | lo] + scale trailer)."""
    _req(w_f32, "w", torch.float32)
    n, k = w_f32.shape
    lib = _lib.load()
    data = torch.empty(int(lib.xml_pack_weights_f16s_bytes(n, k)), dtype=torch.uint8, device=w_f32.device)
    check(lib.xml_pack_weights_f16s(_p(w_f32), _p(data), n, k, _stream()), "xml_pack_weights_f16s")
    return SplitWeight(data, n, k)


def split_f16_rows(x, fixed_log2=None, want_hi=False, want_err=False):
    """x (..., k) f32 -> SplitRows [, hi plane (..., k) torch.float16] [, err (...) f32 = || x - hi / S ||_2 per row].
    fixed_log2 None: one power-of-two scale per row (row maximum -> [2^13, 2^14)); an int: that scale for every row
    (F16_UNIT_LOG2 for unit-norm rows -- what K6 / the re-score expect)."""
    _req(x, "x", torch.float32)
    k = x.shape[-1]
    rows = x.numel() // k
    data = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    inv = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device) if want_hi else None
    err = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device) if want_err else None
    check(_lib.load().xml_split_f16_rows(_p(x), _p(data), _p(inv), _p(hi), _p(err), rows, k,
                                         -1 if fixed_log2 is None else int(fixed_log2), _stream()), "xml_split_f16_rows")
    out = (SplitRows(data, inv),)
    if want_hi:
        out += (hi,)
    if want_err:
        out += (err,)
    return out[0] if len(out) == 1 else out


def unsplit_f16_rows(sr):
    """SplitRows -> f32 rows (hi + lo) / scale (tests, the CPU baseline's view of a split index)."""
    k = sr.shape[-1]
    x = torch.empty(sr.shape, dtype=torch.float32, device=sr.device)
    check(_lib.load().xml_unsplit_f16_rows(_p(sr.data), _p(sr.inv), _p(x), sr.data.numel() // k, k, _stream()),
          "xml_unsplit_f16_rows")
    return x
