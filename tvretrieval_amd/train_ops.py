"""Tensor-level wrappers over the training-step entries of the C ABI (include/xmlhip.h, "TRAINING STEP" block).

Same rules as ops.py: torch owns device memory and the stream, every function launches kernels of libxmlhip.so,
nothing falls back to eager torch.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import check
from .ops import _p, _req, _stream, _workspace, dt_of

F32 = torch.float32


def _r8(x):
    return (x + 7) // 8 * 8


def transpose(x, ld_out=None):
    """(B, R, C) or (R, C) -> (B, C, ld_out>=R) / (C, ld_out); padding columns are zero."""
    _req(x, "x")
    squeeze = x.dim() == 2
    if squeeze:
        x = x.unsqueeze(0)
    b, r, c = x.shape
    ld = r if ld_out is None else ld_out
    y = (torch.zeros if ld != r else torch.empty)((b, c, ld), dtype=x.dtype, device=x.device)
    check(_lib.load().xml_transpose_batched(_p(x), _p(y), b, r, c, ld, dt_of(x), _stream()), "xml_transpose_batched")
    return y[0] if squeeze else y


def transpose_segments(flat_f32, table, max_tiles):
    """table (n_ent, 6) int64 on the device: {src_off, n, k, dst address, ld_dst, col0} -> bf16 W^T blocks, one launch."""
    _req(flat_f32, "flat", F32); _req(table, "table", torch.int64)
    check(_lib.load().xml_transpose_segments(_p(flat_f32), _p(table), table.shape[0], int(max_tiles), _lib.XML_BF16,
                                             _stream()), "xml_transpose_segments")


def colsum(x, rows, cols, out=None):
    """sum over rows of x viewed as (rows, cols) -> f32 (cols,)."""
    _req(x, "x")
    acc = out is not None
    if out is None:
        out = torch.empty(cols, dtype=F32, device=x.device)
    check(_lib.load().xml_colsum(_p(x), dt_of(x), _p(out), rows, cols, int(acc), _stream()), "xml_colsum")
    return out


def relu_bwd(y, dy):
    _req(y, "y"); _req(dy, "dy", y.dtype)
    dx = torch.empty_like(dy)
    check(_lib.load().xml_relu_bwd(_p(y), _p(dy), _p(dx), y.numel(), dt_of(y), _stream()), "xml_relu_bwd")
    return dx


def add_inplace(y, x):
    _req(y, "y"); _req(x, "x")
    assert y.numel() == x.numel()
    check(_lib.load().xml_add_inplace(_p(y), dt_of(y), _p(x), dt_of(x), y.numel(), _stream()), "xml_add_inplace")
    return y


def layernorm_bwd(a, b, g, dy, need_dx=True, dg=None, dbeta=None):
    """-> (dx or None, dg f32, dbeta f32) for y = LN(a [+ b]) * g + beta; a/b/dy (..., d).
    dg / dbeta given: the kernel ACCUMULATES into them (gradient sinks); otherwise fresh zero-filled tensors."""
    _req(a, "a"); _req(g, "g", F32); _req(dy, "dy")
    d = a.shape[-1]
    rows = a.numel() // d
    dg = torch.zeros(d, dtype=F32, device=a.device) if dg is None else _req(dg, "dg", F32)
    dbeta = torch.zeros(d, dtype=F32, device=a.device) if dbeta is None else _req(dbeta, "dbeta", F32)
    wide = d > 1024 or d % 8 != 0
    dx = torch.empty(a.shape, dtype=dy.dtype, device=a.device) if need_dx else None
    ws = _workspace(max(rows * 16, _lib.load().xml_layernorm_bwd_partials_bytes(rows, d)), a.device) if wide else None
    check(_lib.load().xml_layernorm_bwd(_p(a), dt_of(a), _p(b), _p(g), _p(dy), _p(dx), _p(dg), _p(dbeta), rows, d,
                                        dt_of(dy), _p(ws), 0 if ws is None else ws.numel(), _stream()),
          "xml_layernorm_bwd")
    return dx, dg, dbeta


def layernorm_drop_supported(d, dt, need_dx, has_b, p_in):
    """Shapes xml_add_layernorm_drop / xml_layernorm_bwd_drop serve (callers keep separate dropout launches otherwise)."""
    if d % 8 or d > 4096 or dt not in (F32, torch.bfloat16):
        return False
    if d > 1024:
        return dt == torch.bfloat16 and not need_dx and not has_b and p_in == 0
    return True


def add_layernorm_drop(a, b, g, beta, out_dtype, p_in, seed_in, p_out, seed_out):
    """y = drop_out(LN(drop_in(a) + b) * g + beta) in one launch (xml_dropout's masks for the two seeds)."""
    _req(a, "a"); _req(g, "g", F32); _req(beta, "beta", F32)
    d = a.shape[-1]
    rows = a.numel() // d
    if b is not None:
        _req(b, "b", out_dtype)
    y = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    check(_lib.load().xml_add_layernorm_drop(_p(a), dt_of(a), _p(b), _p(g), _p(beta), _p(y), rows, d, dt_of(y),
                                             float(p_in), int(seed_in), float(p_out), int(seed_out), _seed_base_ptr(),
                                             _stream()), "xml_add_layernorm_drop")
    return y


def layernorm_bwd_drop(a, b, g, dy, p_in, seed_in, p_out, seed_out, need_dx=True, dg=None, dbeta=None):
    """Backward of add_layernorm_drop -> (dx or None, dxa or None, dg, dbeta); dxa (gradient of a) only when p_in > 0."""
    _req(a, "a"); _req(g, "g", F32); _req(dy, "dy")
    d = a.shape[-1]
    rows = a.numel() // d
    dg = torch.zeros(d, dtype=F32, device=a.device) if dg is None else _req(dg, "dg", F32)
    dbeta = torch.zeros(d, dtype=F32, device=a.device) if dbeta is None else _req(dbeta, "dbeta", F32)
    dx = torch.empty(a.shape, dtype=dy.dtype, device=a.device) if need_dx else None
    dxa = torch.empty(a.shape, dtype=dy.dtype, device=a.device) if (dx is not None and p_in > 0) else None
    nws = _lib.load().xml_layernorm_bwd_partials_bytes(rows, d)      # scratch for the per-workgroup column sums (0: none used)
    ws = _workspace(nws, a.device) if nws else None
    check(_lib.load().xml_layernorm_bwd_drop_ws(_p(a), dt_of(a), _p(b), _p(g), _p(dy), _p(dx), _p(dxa), _p(dg), _p(dbeta),
                                                rows, d, dt_of(dy), float(p_in), int(seed_in), float(p_out), int(seed_out),
                                                _seed_base_ptr(), _p(ws), nws, _stream()), "xml_layernorm_bwd_drop_ws")
    return dx, dxa, dg, dbeta


def gemm_batched(a, b, scale=1.0, out_f32=False):
    """out[z] = scale * a[z] @ b[z]^T ; a (B, M, K), b (B, N, K) -> (B, M, N) (2-D inputs: B = 1, 2-D output)."""
    _req(a, "a"); _req(b, "b", a.dtype)
    squeeze = a.dim() == 2
    if squeeze:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
    bt, m, k = a.shape
    n = b.shape[1]
    assert b.shape[0] == bt and b.shape[2] == k, "gemm_batched: shape mismatch"
    out = torch.empty((bt, m, n), dtype=F32 if out_f32 else a.dtype, device=a.device)
    check(_lib.load().xml_gemm_batched(_p(a), _p(b), _p(out), bt, m, n, k, float(scale), int(out_f32), dt_of(a),
                                       _stream()), "xml_gemm_batched")
    return out[0] if squeeze else out


def split_heads(x, heads, want=True, want_t=False, col0=0, width=None):
    """x (N, L, ld) -> [(N*heads, L8, dh)], [(N*heads, dh, L8)] taking columns [col0, col0 + width)."""
    _req(x, "x")
    n, l, ld = x.shape
    width = ld - col0 if width is None else width
    dh = width // heads
    l8 = _r8(l)
    dst = torch.empty((n * heads, l8, dh), dtype=x.dtype, device=x.device) if want else None
    dst_t = torch.empty((n * heads, dh, l8), dtype=x.dtype, device=x.device) if want_t else None
    check(_lib.load().xml_split_heads(_p(x), ld, col0, n, l, l8, heads, dh, _p(dst), _p(dst_t), dt_of(x), _stream()),
          "xml_split_heads")
    return dst, dst_t


def merge_heads(xh, n, l, heads, out=None, col0=0):
    """(N*heads, L8, dh) -> (N, L, heads*dh), or into columns [col0, col0 + heads*dh) of `out` (N, L, ld)."""
    _req(xh, "xh")
    l8, dh = xh.shape[1], xh.shape[2]
    if out is None:
        out = torch.empty((n, l, heads * dh), dtype=xh.dtype, device=xh.device)
    _req(out, "out", xh.dtype)
    check(_lib.load().xml_merge_heads(_p(xh), _p(out), out.shape[-1], col0, n, l, l8, heads, dh, dt_of(xh), _stream()),
          "xml_merge_heads")
    return out


DISABLE_GEMM_TN = False               # tests / A-B measurements: transposes + split-K NT GEMM for the weight gradients


def gemm_tn_supported(rows, n, k, dtype):
    return not DISABLE_GEMM_TN and bool(_lib.load().xml_gemm_tn_supported(int(rows), int(n), int(k), dt_of(dtype)))


def gemm_tn(a, b, colsum=False, out=None, colsum_out=None):
    """a (rows, N), b (rows, K) -> a^T b (N, K) f32, or None when the shape / dtype is not served (caller falls back).
    colsum=True: -> (a^T b, column sums of a (N,) f32) from the same launch (a layer's weight and bias gradients).
    out (N, K) f32 [+ colsum_out (N,) f32]: ACCUMULATE into these buffers (persistent .grad views of the optimizer's flat
    gradient buffer) instead of returning fresh tensors -- no fill, no add afterwards; returns True."""
    _req(a, "a"); _req(b, "b", a.dtype)
    rows, n = a.shape
    k = b.shape[1]
    if DISABLE_GEMM_TN or not _lib.load().xml_gemm_tn_supported(rows, n, k, dt_of(a)):
        return None
    if out is not None:
        _req(out, "out", F32)
        assert out.numel() == n * k
        if colsum_out is not None:
            _req(colsum_out, "colsum_out", F32)
            assert colsum_out.numel() == n
        check(_lib.load().xml_gemm_tn(_p(a), _p(b), _p(out), _p(colsum_out), rows, n, k, dt_of(a), 1, _stream()),
              "xml_gemm_tn")
        return True
    buf = torch.empty(n * k + (n if colsum else 0), dtype=F32, device=a.device)      # back to back: one fill inside the entry
    out = buf[:n * k].view(n, k)
    cs = buf[n * k:] if colsum else None
    check(_lib.load().xml_gemm_tn(_p(a), _p(b), _p(out), _p(cs), rows, n, k, dt_of(a), 0, _stream()), "xml_gemm_tn")
    return (out, cs) if colsum else out


DISABLE_FUSED_ATTENTION = False      # tests / A-B measurements: keep the unfused chain for bf16 too

# Device-resident base seed (int64 (1,) tensor) that the dropout kernels ADD to the seed they are given, or None.  A training
# step captured into a HIP graph (train.GraphedTrainStep) sets it: the per-site seeds frozen into the graph's nodes are
# constants, the base seed is advanced by a node of the graph itself, so every replay draws fresh masks.
SEED_BASE = None


def _seed_base_ptr():
    return None if SEED_BASE is None else ctypes.c_void_p(SEED_BASE.data_ptr())


def attention_train_supported(lq, lk, hidden, heads, dtype):
    if DISABLE_FUSED_ATTENTION:
        return False
    return bool(_lib.load().xml_attention_train_supported(int(lq), int(lk), int(hidden), int(heads), dt_of(dtype)))


def _col_ptr(x, col0):
    return ctypes.c_void_p(x.data_ptr() + col0 * x.element_size())


def attention_train_fwd(q, k, v, q_mask, k_mask, heads, hidden, p_drop, seed, q_col=0, k_col=0, v_col=0):
    """Fused attention forward (bf16).  q (N, Lq, ldq), k / v (N, Lk, ld): head blocks start at column *_col of each tensor
    (a fused QKV tensor is passed three times with q_col / k_col / v_col = 0 / H / 2H).  -> (N, Lq, hidden)."""
    _req(q, "q"); _req(k, "k", q.dtype); _req(v, "v", q.dtype)
    n, lq, lk = q.shape[0], q.shape[1], k.shape[1]
    out = torch.empty((n, lq, hidden), dtype=q.dtype, device=q.device)
    check(_lib.load().xml_attention_train_fwd(_col_ptr(q, q_col), q.shape[2], _col_ptr(k, k_col), k.shape[2],
                                              _col_ptr(v, v_col), v.shape[2], _p(q_mask), _p(k_mask), _p(out), hidden, n, lq,
                                              lk, hidden, heads, float(p_drop), int(seed), _seed_base_ptr(), dt_of(q),
                                              _stream()), "xml_attention_train_fwd")
    return out


def attention_train_bwd(q, k, v, q_mask, k_mask, dout, dq, dk, dv, heads, hidden, p_drop, seed, q_col=0, k_col=0, v_col=0,
                        dq_col=0, dk_col=0, dv_col=0):
    """Fused attention backward (bf16): writes the head blocks of dq / dk / dv (column offsets d*_col) from dout (N, Lq, H)."""
    _req(dout, "dout", q.dtype)
    n, lq, lk = q.shape[0], q.shape[1], k.shape[1]
    check(_lib.load().xml_attention_train_bwd(_col_ptr(q, q_col), q.shape[2], _col_ptr(k, k_col), k.shape[2],
                                              _col_ptr(v, v_col), v.shape[2], _p(q_mask), _p(k_mask), _p(dout), dout.shape[2],
                                              _col_ptr(dq, dq_col), dq.shape[2], _col_ptr(dk, dk_col), dk.shape[2],
                                              _col_ptr(dv, dv_col), dv.shape[2], n, lq, lk, hidden, heads, float(p_drop),
                                              int(seed), _seed_base_ptr(), dt_of(q), _stream()), "xml_attention_train_bwd")


def attn_softmax_fwd(s, q_mask, k_mask, n, heads, lq, lk, dh, dtype, want_t=False):
    lq8, lk8 = s.shape[1], s.shape[2]
    p = torch.empty((n * heads, lq8, lk8), dtype=dtype, device=s.device)
    pt = torch.empty((n * heads, lk8, lq8), dtype=dtype, device=s.device) if want_t else None
    check(_lib.load().xml_attn_softmax(_p(s), None, _p(q_mask), _p(k_mask), _p(p), _p(pt), None, None, n, heads, lq,
                                       lk, lq8, lk8, math.sqrt(dh), dt_of(dtype), _stream()), "xml_attn_softmax")
    return p, pt


def attn_softmax_bwd(s, dp, q_mask, k_mask, n, heads, lq, lk, dh, dtype):
    lq8, lk8 = s.shape[1], s.shape[2]
    ds = torch.empty((n * heads, lq8, lk8), dtype=dtype, device=s.device)
    dst = torch.empty((n * heads, lk8, lq8), dtype=dtype, device=s.device)
    check(_lib.load().xml_attn_softmax(_p(s), _p(dp), _p(q_mask), _p(k_mask), None, None, _p(ds), _p(dst), n, heads,
                                       lq, lk, lq8, lk8, math.sqrt(dh), dt_of(dtype), _stream()), "xml_attn_softmax")
    return ds, dst


def modular_pool_bwd(enc, mask, wm, dout, dwm=None):
    """dwm given: accumulated into (a gradient sink); otherwise a fresh zero-filled tensor."""
    _req(enc, "enc"); _req(mask, "mask", F32); _req(wm, "wm", F32); _req(dout, "dout", enc.dtype)
    n, lq, hidden = enc.shape
    n_mod = wm.shape[0]
    denc = torch.empty_like(enc)
    dwm = torch.zeros_like(wm) if dwm is None else _req(dwm, "dwm", F32)
    check(_lib.load().xml_modular_pool_bwd(_p(enc), _p(mask), _p(wm), _p(dout), _p(denc), _p(dwm), n, lq, hidden,
                                           n_mod, dt_of(enc), _stream()), "xml_modular_pool_bwd")
    return denc, dwm


def l2norm_bwd(x, dy):
    _req(x, "x"); _req(dy, "dy", F32)
    d = x.shape[-1]
    dx = torch.empty_like(x)
    check(_lib.load().xml_l2norm_bwd(_p(x), _p(dy), _p(dx), x.numel() // d, d, dt_of(x), _stream()), "xml_l2norm_bwd")
    return dx


def q2c_scores_bwd(qn, cn, mask, dscores, scale=1.0):
    _req(qn, "qn"); _req(cn, "cn", qn.dtype); _req(mask, "mask", F32); _req(dscores, "dscores", F32)
    nq, hidden = qn.shape
    nv, l, _ = cn.shape
    dqn = torch.empty((nq, hidden), dtype=F32, device=qn.device)
    dcn = torch.empty((nv, l, hidden), dtype=F32, device=qn.device)
    check(_lib.load().xml_q2c_scores_bwd(_p(qn), _p(cn), _p(mask), _p(dscores), dscores.stride(0), float(scale), _p(dqn), _p(dcn),
                                         nq, nv, l, hidden, dt_of(qn), _stream()), "xml_q2c_scores_bwd")
    return dqn, dcn


def q2c_scores_l2norm_bwd_supported(nq, nv, l, hidden, dtype):
    return bool(_lib.load().xml_q2c_scores_l2norm_bwd_supported(nq, nv, l, hidden, _lib.XML_F32 if dtype == F32 else
                                                                 (_lib.XML_BF16 if dtype == torch.bfloat16 else -1)))


def q2c_scores_arg(qn, cn, mask, out=None, combine=False):
    """In-batch video-level scores with the arg-max clips: qn (Nq, H), cn (Nv, L, H) normalised rows, mask (Nv, L) f32, all
    unpadded (L <= 128) -> (out (Nq, Nv) f32 [averaged into `out` when combine], arg (Nq, Nv) int32)."""
    _req(qn, "qn"); _req(cn, "cn", qn.dtype); _req(mask, "mask", F32)
    nq, hidden = qn.shape
    nv, l, _ = cn.shape
    assert mask.shape == (nv, l) and cn.shape[2] == hidden
    if out is None:
        assert not combine
        out = torch.empty((nq, nv), dtype=F32, device=qn.device)
    _req(out, "out", F32)
    arg = torch.empty((nq, nv), dtype=torch.int32, device=qn.device)
    check(_lib.load().xml_q2c_scores_arg(_p(qn), _p(cn), _p(mask), _p(out), out.stride(0), _p(arg), arg.stride(0), nq, nv, l,
                                         hidden, int(combine), dt_of(qn), _stream()), "xml_q2c_scores_arg")
    return out, arg


def q2c_scores_l2norm_bwd(query, feat, qn, cn_p, mask_p, dscores, scale=1.0, arg=None):
    """VideoLevelScoresFn backward of one modality in one launch -> (dquery, dfeat) in the activation dtype.
    arg: the (Nq, Nv) int32 arg-max clips of q2c_scores_arg, or None (re-derived per pair with a gradient)."""
    _req(query, "query"); _req(feat, "feat", query.dtype); _req(qn, "qn", query.dtype); _req(cn_p, "cn", query.dtype)
    _req(mask_p, "mask", F32); _req(dscores, "dscores", F32)
    nq, hidden = query.shape
    nv, l, _ = feat.shape
    lpad = cn_p.shape[1]
    assert cn_p.shape == (nv, lpad, hidden) and mask_p.shape == (nv, lpad) and dscores.shape == (nq, nv) and dscores.stride(1) == 1
    dq, df = torch.empty_like(query), torch.empty_like(feat)
    if arg is not None:
        _req(arg, "arg", torch.int32)
        assert arg.shape == (nq, nv)
    check(_lib.load().xml_q2c_scores_l2norm_bwd(_p(query), _p(feat), _p(qn), _p(cn_p), _p(mask_p), _p(dscores),
                                                dscores.stride(0), float(scale), _p(dq), _p(df), nq, nv, l, lpad, hidden,
                                                _p(arg), 0 if arg is None else arg.stride(0), dt_of(query), _stream()),
          "xml_q2c_scores_l2norm_bwd")
    return dq, df


def q2c_scores_l2norm_bwd_multi(sets, dscores, scale=1.0):
    """The same for both modalities in one launch.  sets: list (1 or 2) of (query, feat, qn, cn, mask, arg) with the shapes of
    q2c_scores_l2norm_bwd (arg may be None) -> list of (dquery, dfeat)."""
    _req(dscores, "dscores", F32)
    n_mod = len(sets)
    nq, hidden = sets[0][0].shape
    nv = sets[0][1].shape[0]
    outs, cols = [], [[] for _ in range(10)]
    for query, feat, qn, cn_p, mask_p, arg in sets:
        _req(query, "query"); _req(feat, "feat", query.dtype); _req(qn, "qn", query.dtype); _req(cn_p, "cn", query.dtype)
        _req(mask_p, "mask", F32)
        assert query.shape == (nq, hidden) and feat.shape[0] == nv and feat.shape[2] == hidden and query.dtype == sets[0][0].dtype
        l, lpad = feat.shape[1], cn_p.shape[1]
        assert cn_p.shape == (nv, lpad, hidden) and mask_p.shape == (nv, lpad)
        if arg is not None:
            _req(arg, "arg", torch.int32)
            assert arg.shape == (nq, nv) and arg.stride(0) == nv
        dq, df = torch.empty_like(query), torch.empty_like(feat)
        outs.append((dq, df))
        for c, v in zip(cols, (query, feat, qn, cn_p, mask_p, dq, df, arg)):
            c.append(0 if v is None else v.data_ptr())
        cols[8].append(l); cols[9].append(lpad)
    assert dscores.shape == (nq, nv) and dscores.stride(1) == 1
    vp = lambda vals: (ctypes.c_void_p * n_mod)(*vals)      # noqa: E731
    ip = lambda vals: (ctypes.c_int * n_mod)(*vals)         # noqa: E731
    check(_lib.load().xml_q2c_scores_l2norm_bwd_multi(n_mod, vp(cols[0]), vp(cols[1]), vp(cols[2]), vp(cols[3]), vp(cols[4]),
                                                      _p(dscores), dscores.stride(0), float(scale), vp(cols[5]), vp(cols[6]),
                                                      nq, nv, ip(cols[8]), ip(cols[9]), hidden, vp(cols[7]), nv,
                                                      dt_of(sets[0][0]), _stream()), "xml_q2c_scores_l2norm_bwd_multi")
    return outs


def loss_combine(st_ed, rank2, w_st_ed, w_neg_ctx, w_neg_q):
    """-> (parts (4,) f32 [weighted st_ed, neg_ctx, neg_q, sum], overall 0-d f32); st_ed 0-d / rank2 (2,) f32 or None."""
    ref = st_ed if st_ed is not None else rank2
    for t, nm in ((st_ed, "st_ed"), (rank2, "rank2")):
        if t is not None:
            _req(t, nm, F32)
    parts = torch.empty(4, dtype=F32, device=ref.device)
    overall = torch.empty((), dtype=F32, device=ref.device)
    check(_lib.load().xml_loss_combine(_p(st_ed), _p(rank2), float(w_st_ed), float(w_neg_ctx), float(w_neg_q), _p(parts),
                                       _p(overall), _stream()), "xml_loss_combine")
    return parts, overall


def loss_combine_bwd(g, w_st_ed, w_neg_ctx, w_neg_q, want_st_ed, want_rank):
    _req(g, "g", F32)
    d0 = torch.empty((), dtype=F32, device=g.device) if want_st_ed else None
    d1 = torch.empty(2, dtype=F32, device=g.device) if want_rank else None
    check(_lib.load().xml_loss_combine_bwd(_p(g), float(w_st_ed), float(w_neg_ctx), float(w_neg_q), _p(d0), _p(d1),
                                           _stream()), "xml_loss_combine_bwd")
    return d0, d1


def pair_sim(q, f2):
    _req(q, "q"); _req(f2, "f2", q.dtype)
    n, l, hidden = f2.shape
    sim = torch.empty((n, l), dtype=F32, device=q.device)
    check(_lib.load().xml_pair_sim(_p(q), _p(f2), _p(sim), n, l, hidden, dt_of(q), _stream()), "xml_pair_sim")
    return sim


def pair_sim_bwd(q, f2, dsim):
    _req(q, "q"); _req(f2, "f2", q.dtype); _req(dsim, "dsim", F32)
    n, l, hidden = f2.shape
    dq, df2 = torch.empty_like(q), torch.empty_like(f2)
    check(_lib.load().xml_pair_sim_bwd(_p(q), _p(f2), _p(dsim), _p(dq), _p(df2), n, l, hidden, dt_of(q), _stream()),
          "xml_pair_sim_bwd")
    return dq, df2


def span_loss(sims, conv_w, masks, st_ed, merged, ks, gout=None):
    """sims: 1 or 2 (N, L) f32; conv_w flat f32 [st filters | ed filters]; masks like sims; st_ed (N, 2) int64.
    gout None -> loss (0-d f32).  gout (1,) f32 -> (dsims list, dconv_w)."""
    for t in sims:
        _req(t, "sim", F32)
    for t in masks:
        _req(t, "mask", F32)
    _req(conv_w, "conv_w", F32); _req(st_ed, "st_ed", torch.int64)
    n, l = sims[0].shape
    n_sim = len(sims)
    s1 = sims[1] if n_sim > 1 else None
    m1 = masks[1] if len(masks) > 1 else None
    lib = _lib.load()
    if gout is None:
        loss = torch.empty(1, dtype=F32, device=sims[0].device)
        check(lib.xml_span_loss(_p(sims[0]), _p(s1), _p(conv_w), _p(masks[0]), _p(m1), _p(st_ed), int(merged), n_sim,
                                ks, n, l, None, _p(loss), None, None, None, _stream()), "xml_span_loss")
        return loss[0]
    _req(gout, "gout", F32)
    if n_sim == 2:          # one allocation [dsim0 | dsim1 | dconv_w]: xml_span_loss zero-fills it with one launch
        buf = torch.empty(2 * n * l + conv_w.numel(), dtype=F32, device=sims[0].device)
        dsims = [buf[:n * l].view(n, l), buf[n * l:2 * n * l].view(n, l)]
        dconv = buf[2 * n * l:]
    else:
        dsims = [torch.empty_like(s) for s in sims]
        dconv = torch.empty_like(conv_w)
    check(lib.xml_span_loss(_p(sims[0]), _p(s1), _p(conv_w), _p(masks[0]), _p(m1), _p(st_ed), int(merged), n_sim, ks, n,
                            l, _p(gout), None, _p(dsims[0]), _p(dsims[1]) if n_sim > 1 else None, _p(dconv),
                            _stream()), "xml_span_loss")
    return dsims, dconv


def rank_loss(scores, ranks_ctx, ranks_q, margin, lse, gout=None):
    """scores (N, N) f32, ranks int32 (N,).  gout None -> losses (2,) f32; else dscores (N, N)."""
    _req(scores, "scores", F32); _req(ranks_ctx, "ranks_ctx", torch.int32); _req(ranks_q, "ranks_q", torch.int32)
    n = scores.shape[0]
    assert scores.shape == (n, n)
    lib = _lib.load()
    if gout is None:
        losses = torch.empty(2, dtype=F32, device=scores.device)
        check(lib.xml_rank_loss(_p(scores), _p(ranks_ctx), _p(ranks_q), float(margin), int(lse), n, None, _p(losses),
                                None, _stream()), "xml_rank_loss")
        return losses
    _req(gout, "gout", F32)
    ds = torch.empty_like(scores)
    check(lib.xml_rank_loss(_p(scores), _p(ranks_ctx), _p(ranks_q), float(margin), int(lse), n, _p(gout), None, _p(ds),
                            _stream()), "xml_rank_loss")
    return ds


def dropout(x, p, seed, out=None):
    """y = mask(seed) * x / (1 - p); the same call on a gradient is the backward pass."""
    _req(x, "x")
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().xml_dropout(_p(x), _p(y), x.numel(), float(p), int(seed), _seed_base_ptr(), dt_of(x), _stream()),
          "xml_dropout")
    return y


def clip_grad_norm(flat_g, max_norm):
    """In-place global-norm clip of the flat gradient buffer (torch.nn.utils.clip_grad_norm_ semantics)."""
    _req(flat_g, "flat_g", F32)
    ws = torch.empty(1, dtype=F32, device=flat_g.device)
    check(_lib.load().xml_clip_grad_norm(_p(flat_g), flat_g.numel(), float(max_norm), _p(ws), _stream()),
          "xml_clip_grad_norm")
    return flat_g


def bert_adam_step(p, g, m, v, seg_off, seg_lr, seg_wd, norms, lr_mult, b1, b2, eps, max_grad_norm, seg_active=None,
                   seg_lr_mult=None, norm_ws=None):
    """norm_ws: ceil(p.numel() / 1024) f32 of scratch (XML_ADAM_NORM_BLOCK; atomic-free gradient norms) or None."""
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (seg_lr, "seg_lr"), (seg_wd, "seg_wd"), (norms, "norms")):
        _req(t, nm, F32)
    if norm_ws is not None:
        _req(norm_ws, "norm_ws", F32)
        assert norm_ws.numel() >= (p.numel() + 1023) // 1024, "bert_adam_step: norm_ws too small"
    _req(seg_off, "seg_off", torch.int64)
    if seg_active is not None:
        _req(seg_active, "seg_active", torch.uint8)
    if seg_lr_mult is not None:
        _req(seg_lr_mult, "seg_lr_mult", F32)
    check(_lib.load().xml_bert_adam_step(_p(p), _p(g), _p(m), _p(v), _p(seg_off), _p(seg_lr), _p(seg_wd),
                                         seg_lr.numel(), p.numel(), float(lr_mult), float(b1), float(b2), float(eps),
                                         float(max_grad_norm), _p(norms), _p(seg_active), _p(seg_lr_mult), _p(norm_ws),
                                         _stream()),
          "xml_bert_adam_step")
