"""torch.autograd.Function nodes whose forward AND backward are libxmlhip.so kernels.

The reference trains with torch autograd over eager ops (xml/train.py:78-95).  Here autograd is only the tape:
each node below records what its hand-written backward kernel needs and launches it.  Parameters are the f32
masters of model_xml.XML (so `.grad` is f32, as the reference's); activations and their gradients use the
compute dtype (f32 or bf16).  Dropout is not applied (see train.py).

Nodes (reference op -> node):
  nn.Linear (+ReLU)                         LinearFn          dX = dY W, dW = dY^T X (MFMA GEMMs), db = colsum
  nn.LayerNorm(a [+ b])                     LayerNormFn
  BertSelfAttention core                    AttentionCoreFn   softmax(QK^T/sqrt(d) + mask) V per head
  get_modularized_queries                   ModularPoolFn
  get_video_level_scores                    VideoLevelScoresFn
  einsum("bd,bld->bl")                      PairSimFn
  conv predictors + CE                      SpanLossFn
  get_video_level_loss                      RankLossFn
"""
import torch

from . import ops
from . import train_ops as T

F32 = torch.float32


def _r8(x):
    return (x + 7) // 8 * 8


def _packed(w, dtype):
    w = w.detach()
    return w.contiguous() if dtype == F32 else ops.pack_weights(w.float().contiguous(), dtype)


def _weights(params, dtype, ctx=None):
    """Compute-dtype (sum N, K) weight of the row-concatenated f32 parameters: the optimizer's copy of this step when it is
    current (BertAdam.refresh_shadows), else converted now.
    ctx: the autograd context that will SAVE the returned tensor for its backward.  The optimizer's copy is a view of a buffer
    that the next refresh overwrites in place (through ctypes: no autograd version bump), so the context remembers which
    refresh it saw and its backward (_shadow_guard) refuses to run on a later one."""
    if ctx is not None:
        ctx.shadow_gen = None
    if dtype != F32:
        reg = getattr(params[0], "_xml_sink", None)
        if reg is not None:
            v = reg.opt.shadow_w(params, dtype)
            if v is not None:
                if ctx is not None:
                    ctx.shadow_gen = (reg.opt, reg.opt.shadow_generation())
                return v
    if len(params) == 1:
        return _packed(params[0], dtype)
    wf = _adjacent(params, "flat_p")      # usually back to back in the optimizer's flat parameter buffer: one view, no cat
    wcat = wf.view(-1, params[0].shape[1]) if wf is not None else torch.cat([p.detach() for p in params], 0)
    return _packed(wcat, dtype)


def _shadow_guard(ctx):
    """forward A, optimizer step + refresh B, backward A: the weights A saved were a view of the per-step shadow buffer and
    now hold B's values -- dX would silently use the new weights.  Unsupported; say so instead."""
    g = getattr(ctx, "shadow_gen", None)
    if g is not None and g[0].shadow_generation() != g[1]:
        raise RuntimeError("backward of a forward pass that ran before the last optimizer step: its saved compute-dtype "
                           "weights were the optimizer's per-step shadow copies, which that step has overwritten "
                           "(train.SHADOW_WEIGHTS = False keeps private copies)")


def _weights_t(params, w):
    """(K, ceil8(N)) transpose of the compute-dtype weight w of `params` (the B operand of dX = dY W): the optimizer's
    per-step copy when current, else transposed now."""
    n = w.shape[0]
    if w.dtype != F32:
        reg = getattr(params[0], "_xml_sink", None)
        if reg is not None:
            t = reg.opt.shadow_t(params, w.dtype)
            if t is not None:
                return t
    return T.transpose(w, _r8(n)) if n % 8 else T.transpose(w)


# ---- gradient sinks ---------------------------------------------------------------------------------------------------
# BertAdam lays every parameter's .grad into ONE flat f32 buffer that it zeroes once per step (train.BertAdam._flatten) and
# registers a GradSink on the parameter.  A backward node whose kernels ACCUMULATE (f32 atomics: the weight-gradient GEMM,
# LayerNorm / pooling parameter gradients, column sums) then adds straight into that view and returns None for the
# parameter instead of a fresh tensor -- autograd's AccumulateGrad would otherwise run `p.grad += dW` per parameter (96
# elementwise launches per step at the C5 shape) on top of the 70 fills of the temporaries.  The sink tells the optimizer
# that the gradient is there (what the post-accumulate hook does on the ordinary path: "has ever received a gradient",
# data-parallel bucket bookkeeping).
FUSED_LOSS_TAIL = True         # A-B runs / tests: False = the separate q2c_scores_bwd + l2norm_bwd launches, torch loss sum
USE_GRAD_SINKS = True          # tests / A-B runs: False = every node returns its gradients to autograd


class GradSink(object):
    def __init__(self, opt, index, offset):
        self.opt, self.index, self.offset = opt, index, offset      # offset in elements into opt.flat_g / opt.flat_p


def _sink(p):
    """p's persistent .grad view inside the optimizer's flat buffer, or None (no optimizer, view replaced, sinks off)."""
    if not USE_GRAD_SINKS or p is None:
        return None
    reg = getattr(p, "_xml_sink", None)
    g = p.grad
    if reg is None or g is None or g.dtype != F32 or g.data_ptr() != reg.opt.flat_g.data_ptr() + 4 * reg.offset:
        return None
    return g


def _claim(ctx, params, positions, extra_ok=True):
    """Forward-time decision for the listed parameters of a node (positions = their indices in the node's inputs): True =
    this node's backward accumulates their gradients straight into the sinks and returns None for them.
    "The gradient of parameter i is complete" is still signalled by autograd: the parameter's AccumulateGrad node runs --
    and its post-accumulate hook fires -- once EVERY node that uses the parameter has run its backward, whether those
    returned tensors or None (checked on the GPU by tests/rccl_world1_check.py: reducer == plain path; a parameter shared
    by two nodes, like the context positional table, is reported once, after both)."""
    live = [p for p, pos in zip(params, positions) if p is not None and ctx.needs_input_grad[pos]]
    return bool(live) and extra_ok and all(_sink(p) is not None for p in live)


def _adjacent(params, flat_attr):
    """The parameters' slices of the optimizer's flat buffer as ONE tensor if they lie back to back in the given order
    (e.g. query / key / value weights of a layer within one parameter group), else None."""
    regs = [getattr(p, "_xml_sink", None) for p in params]
    if any(r is None for r in regs) or any(r.opt is not regs[0].opt for r in regs):
        return None
    flat = getattr(regs[0].opt, flat_attr)
    off = regs[0].offset
    for p, r in zip(params, regs):
        if r.offset != off:
            return None
        # ... and the live tensor must still BE that slice: after model.to() / .float(), a second optimizer or a stale one
        # the registered offsets survive while p.data (or p.grad) points elsewhere -- the flat view would be stale values
        t = p.data if flat_attr == "flat_p" else p.grad
        if t is None or t.dtype != flat.dtype or t.data_ptr() != flat.data_ptr() + flat.element_size() * r.offset:
            return None
        off += p.numel()
    return flat[regs[0].offset:off]


_HOOK_ON_UNDEFINED_GRAD = None


def hook_fires_on_undefined_grad():
    """One-off self-check of what the gradient sinks rely on: AccumulateGrad still runs the parameter's post-accumulate hook
    when the node feeding it returned None (true on torch 2.10; older 2.x returned early before the hook -- sunk parameters
    would then never be reported to the optimizer / the bucket reducer, silently)."""
    global _HOOK_ON_UNDEFINED_GRAD
    if _HOOK_ON_UNDEFINED_GRAD is None:
        fired = []
        p = torch.nn.Parameter(torch.zeros(2))
        p.register_post_accumulate_grad_hook(lambda _p: fired.append(1))

        class _NoneGrad(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x.clone()

            @staticmethod
            def backward(ctx, dy):
                return dy, None

        x = torch.ones(2, requires_grad=True)
        _NoneGrad.apply(x, p).sum().backward()
        _HOOK_ON_UNDEFINED_GRAD = bool(fired)
    return _HOOK_ON_UNDEFINED_GRAD


class LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b); x (..., K) compute dtype, W (N, K) / b (N) f32 masters."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        w = _weights((weight,), x.dtype, ctx)
        y = ops.linear(x.contiguous(), w, None if bias is None else bias.detach().float().contiguous(), relu=relu)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        n, k = w.shape
        ctx.sunk = _claim(ctx, (weight, bias), (1, 2), ctx.needs_input_grad[1] and
                          T.gemm_tn_supported(x.numel() // k, n, k, x.dtype))
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _shadow_guard(ctx)
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = T.relu_bwd(y, dy)
        n, k = w.shape
        rows = x.numel() // k
        dy2, x2 = dy.view(rows, n), x.contiguous().view(rows, k)
        dx = dw = db = None
        weight, bias = ctx.params
        if ctx.needs_input_grad[0]:
            wt = _weights_t((weight,), w)                                          # (K, N[8])
            a = dy2
            if n % 8:                                                              # pad the reduction dim
                a = torch.zeros((rows, _r8(n)), dtype=dy.dtype, device=dy.device)
                a[:, :n] = dy2
            dx = ops.linear(a, wt).view(x.shape)                                   # dX = dY W
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.sunk:
            gw, gb = _sink(weight), (_sink(bias) if want_db else None)
            if gw is not None and (gb is not None or not want_db):
                assert T.gemm_tn(dy2, x2, out=gw, colsum_out=gb)                   # straight into the flat .grad buffer
                return dx, None, None, None
        if ctx.needs_input_grad[1]:
            dw = T.gemm_tn(dy2, x2, colsum=want_db)                               # dW = dY^T X (+ db = column sums of dY)
            if dw is not None and want_db:
                dw, db = dw
            if dw is None:
                r8 = _r8(rows)
                dyt = T.transpose(dy2, r8)                                        # (N, rows8)
                xt = T.transpose(x2, r8)                                          # (K, rows8)
                dw = T.gemm_batched(dyt, xt, out_f32=True)
        if want_db and db is None:
            db = T.colsum(dy2, rows, n)
        return dx, dw, db, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in training mode; the mask is regenerated from (seed, index) in the backward pass."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return T.dropout(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return T.dropout(dy.contiguous(), ctx.p, ctx.seed), None, None


class LayerNormFn(torch.autograd.Function):
    """y = LN(a [+ b]) * g + beta; a may be raw f32 features (no gradient), b optional residual."""

    @staticmethod
    def forward(ctx, a, b, g, beta, out_dtype, drop_in=None, drop_out=None):
        """drop_in / drop_out: (p, seed) of the dropout site in front of `a` / behind the LayerNorm, applied inside the
        LayerNorm kernels (xml_add_layernorm_drop) -- the caller checks train_ops.layernorm_drop_supported first."""
        gf, bf = g.detach().float().contiguous(), beta.detach().float().contiguous()
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        drop_in = drop_in if drop_in and drop_in[0] > 0 else None
        drop_out = drop_out if drop_out and drop_out[0] > 0 else None
        ctx.drop = (drop_in or (0.0, 0)) + (drop_out or (0.0, 0)) if (drop_in or drop_out) else None
        if ctx.drop:
            y = T.add_layernorm_drop(a, b, gf, bf, out_dtype, *ctx.drop)
        else:
            y = ops.add_layernorm(a, b, gf, bf, out_dtype=out_dtype)
        ctx.params = (g, beta)
        ctx.sunk = _claim(ctx, (g, beta), (2, 3), ctx.needs_input_grad[2] and ctx.needs_input_grad[3])
        ctx.save_for_backward(a, b, gf)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, g = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0] or (b is not None and ctx.needs_input_grad[1])
        pg, pbeta = ctx.params
        sg, sb = (_sink(pg), _sink(pbeta)) if ctx.sunk else (None, None)
        sunk = sg is not None and sb is not None
        dxa = None
        if ctx.drop:
            dx, dxa, dg, dbeta = T.layernorm_bwd_drop(a, b, g, dy.contiguous(), *ctx.drop, need_dx=need_dx,
                                                      dg=sg.view(-1) if sunk else None, dbeta=sb.view(-1) if sunk else None)
        else:
            dx, dg, dbeta = T.layernorm_bwd(a, b, g, dy.contiguous(), need_dx=need_dx, dg=sg.view(-1) if sunk else None,
                                            dbeta=sb.view(-1) if sunk else None)
        if sunk:
            dg = dbeta = None
        da = db = None
        if ctx.needs_input_grad[0]:
            da = dx if dxa is None else dxa
            da = da if da.dtype == a.dtype else ops.convert(da, a.dtype)
        if b is not None and ctx.needs_input_grad[1]:
            db = dx if dx.dtype == b.dtype else ops.convert(dx, b.dtype)
        return da, db, dg, dbeta, None, None, None


class AttentionCoreFn(torch.autograd.Function):
    """Multi-head softmax(Q K^T / sqrt(dh) + (1 - q_mask (x) k_mask) * -1e4) V (xml/model_components.py:266-303,
    dropout on the probabilities omitted).  q (N, Lq, H), k / v (N, Lk, H); masks f32 (q_mask may be None)."""

    @staticmethod
    def forward(ctx, q, k, v, q_mask, k_mask, heads, p_drop=0.0, seed=0):
        n, lq, hidden = q.shape
        lk = k.shape[1]
        dh = hidden // heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.heads = heads
        ctx.drop = (p_drop, seed)
        ctx.fused = T.attention_train_supported(lq, lk, hidden, heads, q.dtype)
        if ctx.fused:       # bf16: one launch, nothing but q / k / v kept for the backward pass (attention_train.hip)
            ctx.save_for_backward(q, k, v, None, q_mask, k_mask)
            return T.attention_train_fwd(q, k, v, q_mask, k_mask, heads, hidden, p_drop, seed)
        qh, _ = T.split_heads(q, heads)
        kh, _ = T.split_heads(k, heads)
        _, vht = T.split_heads(v, heads, want=False, want_t=True)
        s = T.gemm_batched(qh, kh, out_f32=True)                       # (N*h, lq8, lk8)
        p, _ = T.attn_softmax_fwd(s, q_mask, k_mask, n, heads, lq, lk, dh, q.dtype)
        if p_drop > 0:
            T.dropout(p, p_drop, seed, out=p)                          # attention_probs dropout, :297
        oh = T.gemm_batched(p, vht)                                    # (N*h, lq8, dh)
        out = T.merge_heads(oh, n, lq, heads)
        ctx.save_for_backward(q, k, v, s, q_mask, k_mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, s, q_mask, k_mask = ctx.saved_tensors
        heads = ctx.heads
        n, lq, hidden = q.shape
        lk = k.shape[1]
        dh = hidden // heads
        p_drop, seed = ctx.drop
        if ctx.fused:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            T.attention_train_bwd(q, k, v, q_mask, k_mask, dout.contiguous(), dq, dk, dv, heads, hidden, p_drop, seed)
            return dq, dk, dv, None, None, None, None, None
        doh, doht = T.split_heads(dout.contiguous(), heads, want=True, want_t=True)
        vh, _ = T.split_heads(v, heads)
        dp = T.gemm_batched(doh, vh, out_f32=True)                     # dP = dO V^T   (w.r.t. the dropped probs)
        if p_drop > 0:
            p, _ = T.attn_softmax_fwd(s, q_mask, k_mask, n, heads, lq, lk, dh, q.dtype)
            T.dropout(p, p_drop, seed, out=p)
            pt = T.transpose(p)                                        # dropped P^T for dV
            T.dropout(dp, p_drop, seed, out=dp)                        # through the dropout: same mask, same scale
        else:
            _, pt = T.attn_softmax_fwd(s, q_mask, k_mask, n, heads, lq, lk, dh, q.dtype, want_t=True)
        ds, dst = T.attn_softmax_bwd(s, dp, q_mask, k_mask, n, heads, lq, lk, dh, q.dtype)
        dvh = T.gemm_batched(pt, doht)                                 # dV = P^T dO      (lk8, dh)
        _, qht = T.split_heads(q, heads, want=False, want_t=True)
        _, kht = T.split_heads(k, heads, want=False, want_t=True)
        dqh = T.gemm_batched(ds, kht)                                  # dQ = dS K        (lq8, dh)
        dkh = T.gemm_batched(dst, qht)                                 # dK = dS^T Q      (lk8, dh)
        return (T.merge_heads(dqh, n, lq, heads), T.merge_heads(dkh, n, lk, heads), T.merge_heads(dvh, n, lk, heads),
                None, None, None, None, None)


class QkvFn(torch.autograd.Function):
    """Several projections of the same input as ONE GEMM on the stacked weight: the three of BertSelfAttention
    (xml/model_components.py:266-272), [Wq; Wk; Wv]: x (N, L, H) -> (N, L, 3H), or key + value of the cross attention
    (xml/model_xml.py:357-373), x -> (N, L, 2H).  apply(x, w0, b0, w1, b1, ...), all weights (H, K).  Backward: one dX GEMM,
    one split-K dW GEMM (nH x K), one column sum; the input is transposed once instead of n times."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = tuple(wb[0::2]), tuple(wb[1::2])
        assert len(ws) == len(bs) >= 2 and all(w_.shape == ws[0].shape for w_ in ws)
        w = _weights(ws, x.dtype, ctx)
        bf = _adjacent(bs, "flat_p")
        b = bf.detach() if bf is not None else torch.cat([b_.detach() for b_ in bs], 0).float().contiguous()
        ctx.params = tuple(wb)
        k = w.shape[1]
        ctx.sunk = _claim(ctx, ctx.params, tuple(range(1, 1 + len(wb))),
                          all(ctx.needs_input_grad[1:]) and _adjacent(ws, "flat_g") is not None and
                          _adjacent(bs, "flat_g") is not None and
                          T.gemm_tn_supported(x.numel() // k, w.shape[0], k, x.dtype))
        ctx.save_for_backward(x, w)
        return ops.linear(x.contiguous(), w, b)

    @staticmethod
    def backward(ctx, dy):
        return QkvFn._backward(ctx, dy, None)

    @staticmethod
    def _backward(ctx, dy, dres):
        _shadow_guard(ctx)
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        nh, k = w.shape
        ws, bs = ctx.params[0::2], ctx.params[1::2]
        h = nh // len(ws)
        rows = x.numel() // k
        dy2, x2 = dy.view(rows, nh), x.contiguous().view(rows, k)
        dx = None
        if ctx.needs_input_grad[0]:
            # dres (QkvResFn): the gradient that reached x through the residual connection, added in the dX GEMM's epilogue
            add = None if dres is None else dres.contiguous().view(rows, k)
            if add is not None and (add.dtype != dy2.dtype or w.dtype is ops.F16S):
                dx = ops.linear(dy2, _weights_t(ws, w)).view(x.shape) + dres
            else:
                dx = ops.linear(dy2, _weights_t(ws, w), addend=add).view(x.shape)
        if ctx.sunk:
            if all(_sink(p) is not None for p in ctx.params):
                gw, gb = _adjacent(ws, "flat_g"), _adjacent(bs, "flat_g")
                assert T.gemm_tn(dy2, x2, out=gw, colsum_out=gb)
                return (dx,) + (None,) * len(ctx.params)
        dw = T.gemm_tn(dy2, x2, colsum=True)                                               # (nH, K) and the n bias gradients
        if dw is None:
            r8 = _r8(rows)
            dw = T.gemm_batched(T.transpose(dy2, r8), T.transpose(x2, r8), out_f32=True)
            db = T.colsum(dy2, rows, nh)
        else:
            dw, db = dw
        out = [dx]
        for i in range(len(ws)):
            out += [dw[i * h:(i + 1) * h], db[i * h:(i + 1) * h]]
        return tuple(out)


class QkvResFn(torch.autograd.Function):
    """QkvFn for an input that ALSO feeds the block's residual connection (BertAttention: x -> [Wq; Wk; Wv] and x + dense(..)
    into the LayerNorm, xml/model_components.py:201-216): returns (projection, x) -- the second output is x itself, to be used
    as the residual operand -- so that both gradients of x arrive at THIS node and the residual's rides into the dX GEMM as its
    epilogue addend; as two consumers of x autograd summed them with a launch of its own per block (13 per step)."""

    @staticmethod
    def forward(ctx, x, *wb):
        y = QkvFn.forward(ctx, x, *wb)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        if dy is None:                       # (only the residual was used)
            return (dres,) + (None,) * len(ctx.params)
        return QkvFn._backward(ctx, dy, dres)


class AttentionKvFn(torch.autograd.Function):
    """AttentionCoreFn with the keys and values as the column blocks of ONE (N, Lk, 2H) tensor (the stacked key / value
    projection of the cross attention, QkvFn) -- fused bf16 kernels only: the caller checks attention_train_supported."""

    @staticmethod
    def forward(ctx, q, kv, q_mask, k_mask, heads, p_drop=0.0, seed=0):
        q, kv = q.contiguous(), kv.contiguous()
        hidden = q.shape[2]
        assert kv.shape[2] == 2 * hidden and T.attention_train_supported(q.shape[1], kv.shape[1], hidden, heads, q.dtype)
        ctx.cfg = (heads, p_drop, seed)
        ctx.save_for_backward(q, kv, q_mask, k_mask)
        return T.attention_train_fwd(q, kv, kv, q_mask, k_mask, heads, hidden, p_drop, seed, 0, 0, hidden)

    @staticmethod
    def backward(ctx, dout):
        q, kv, q_mask, k_mask = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        hidden = q.shape[2]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        T.attention_train_bwd(q, kv, kv, q_mask, k_mask, dout.contiguous(), dq, dkv, dkv, heads, hidden, p_drop, seed,
                              0, 0, hidden, 0, 0, hidden)
        return dq, dkv, None, None, None, None, None


class AttentionQkvFn(torch.autograd.Function):
    """AttentionCoreFn on a fused (N, L, 3H) projection tensor (self-attention: q, k, v are its column blocks)."""

    @staticmethod
    def forward(ctx, qkv, k_mask, heads, p_drop=0.0, seed=0):
        n, l, h3 = qkv.shape
        hidden = h3 // 3
        dh = hidden // heads
        qkv = qkv.contiguous()
        ctx.cfg = (heads, p_drop, seed)
        ctx.fused = T.attention_train_supported(l, l, hidden, heads, qkv.dtype)
        if ctx.fused:
            ctx.save_for_backward(qkv, None, k_mask)
            return T.attention_train_fwd(qkv, qkv, qkv, None, k_mask, heads, hidden, p_drop, seed, 0, hidden, 2 * hidden)
        qh, _ = T.split_heads(qkv, heads, col0=0, width=hidden)
        kh, _ = T.split_heads(qkv, heads, col0=hidden, width=hidden)
        _, vht = T.split_heads(qkv, heads, want=False, want_t=True, col0=2 * hidden, width=hidden)
        s = T.gemm_batched(qh, kh, out_f32=True)
        p, _ = T.attn_softmax_fwd(s, None, k_mask, n, heads, l, l, dh, qkv.dtype)
        if p_drop > 0:
            T.dropout(p, p_drop, seed, out=p)
        out = T.merge_heads(T.gemm_batched(p, vht), n, l, heads)
        ctx.save_for_backward(qkv, s, k_mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, s, k_mask = ctx.saved_tensors
        heads, p_drop, seed = ctx.cfg
        n, l, h3 = qkv.shape
        hidden = h3 // 3
        dh = hidden // heads
        if ctx.fused:
            dqkv = torch.empty_like(qkv)
            T.attention_train_bwd(qkv, qkv, qkv, None, k_mask, dout.contiguous(), dqkv, dqkv, dqkv, heads, hidden, p_drop, seed,
                                  0, hidden, 2 * hidden, 0, hidden, 2 * hidden)
            return dqkv, None, None, None, None
        doh, doht = T.split_heads(dout.contiguous(), heads, want=True, want_t=True)
        vh, _ = T.split_heads(qkv, heads, col0=2 * hidden, width=hidden)
        dp = T.gemm_batched(doh, vh, out_f32=True)
        if p_drop > 0:
            p, _ = T.attn_softmax_fwd(s, None, k_mask, n, heads, l, l, dh, qkv.dtype)
            T.dropout(p, p_drop, seed, out=p)
            pt = T.transpose(p)
            T.dropout(dp, p_drop, seed, out=dp)
        else:
            _, pt = T.attn_softmax_fwd(s, None, k_mask, n, heads, l, l, dh, qkv.dtype, want_t=True)
        ds, dst = T.attn_softmax_bwd(s, dp, None, k_mask, n, heads, l, l, dh, qkv.dtype)
        _, qht = T.split_heads(qkv, heads, want=False, want_t=True, col0=0, width=hidden)
        _, kht = T.split_heads(qkv, heads, want=False, want_t=True, col0=hidden, width=hidden)
        dqkv = torch.empty_like(qkv)
        T.merge_heads(T.gemm_batched(ds, kht), n, l, heads, out=dqkv, col0=0)              # dQ = dS K
        T.merge_heads(T.gemm_batched(dst, qht), n, l, heads, out=dqkv, col0=hidden)        # dK = dS^T Q
        T.merge_heads(T.gemm_batched(pt, doht), n, l, heads, out=dqkv, col0=2 * hidden)    # dV = P^T dO
        return dqkv, None, None, None, None


class ModularPoolFn(torch.autograd.Function):
    """get_modularized_queries (xml/model_xml.py:410-423) -> (n_mod, N, H)."""

    @staticmethod
    def forward(ctx, enc, mask, wm):
        wmf = wm.detach().float().contiguous()
        enc = enc.contiguous()
        out = ops.modular_pool(enc, mask, wmf)
        ctx.params = (wm,)
        ctx.sunk = _claim(ctx, (wm,), (2,))
        ctx.save_for_backward(enc, mask, wmf)
        return out

    @staticmethod
    def backward(ctx, dout):
        enc, mask, wmf = ctx.saved_tensors
        wm, = ctx.params
        sw = _sink(wm) if ctx.sunk else None
        denc, dwm = T.modular_pool_bwd(enc, mask, wmf, dout.contiguous(), dwm=sw)
        if sw is not None:
            dwm = None
        return denc, None, dwm


class VideoLevelScoresFn(torch.autograd.Function):
    """get_video_level_scores per modality, averaged (xml/model_xml.py:436-453,572-574): F.normalize both sides,
    cosine vs every clip, mask_logits, max over clips -> (Nq, Nv) f32.
    forward(n_mod, queries..., feat1s..., masks...)."""

    @staticmethod
    def forward(ctx, n_mod, *rest):
        saved = []
        scores = None
        for i in range(n_mod):
            query, feat1, mask = rest[i].contiguous(), rest[n_mod + i].contiguous(), rest[2 * n_mod + i]
            n, l, hidden = feat1.shape
            lpad = (l + 15) // 16 * 16
            qn = ops.l2norm_rows(query)
            cn = ops.l2norm_rows(feat1)
            if (FUSED_LOSS_TAIL and l <= 128 and hidden % 8 == 0 and
                    T.q2c_scores_l2norm_bwd_supported(query.shape[0], n, l, hidden, query.dtype)):
                # training form: unpadded clips, the arg-max clip kept for the one-launch backward (loss_tail.hip)
                mk = mask.contiguous()
                if scores is None:
                    scores, arg = T.q2c_scores_arg(qn, cn, mk)
                else:
                    _, arg = T.q2c_scores_arg(qn, cn, mk, out=scores, combine=True)
                saved += [query, feat1, qn, cn, mk, arg]
                continue
            if lpad != l:
                cn_p = torch.zeros((n, lpad, hidden), dtype=cn.dtype, device=cn.device)
                mk_p = torch.zeros((n, lpad), dtype=F32, device=cn.device)
                cn_p[:, :l] = cn
                mk_p[:, :l] = mask
            else:
                cn_p, mk_p = cn, mask.contiguous()
            if scores is None:
                scores = ops.q2c_scores(qn, cn_p, mk_p)
            else:
                ops.q2c_scores(qn, cn_p, mk_p, out=scores, combine=True)      # (a + b) / 2
            saved += [query, feat1, qn, cn_p, mk_p, None]
        assert n_mod in (1, 2)
        ctx.n_mod = n_mod
        ctx.save_for_backward(*saved)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        n_mod = ctx.n_mod
        dscores = dscores.contiguous()
        dq, df = [], []
        sets = [ctx.saved_tensors[6 * i:6 * i + 6] for i in range(n_mod)]
        if n_mod == 2 and all(s_[5] is not None for s_ in sets) and sets[0][0].shape == sets[1][0].shape and \
                sets[0][1].shape[0] == sets[1][1].shape[0]:
            res = T.q2c_scores_l2norm_bwd_multi(sets, dscores, scale=1.0 / n_mod)      # both modalities, one launch
            return (None,) + tuple(r[0] for r in res) + tuple(r[1] for r in res) + (None,) * n_mod
        for i in range(n_mod):
            query, feat1, qn, cn_p, mk_p, arg = ctx.saved_tensors[6 * i:6 * i + 6]
            if arg is not None or (FUSED_LOSS_TAIL and T.q2c_scores_l2norm_bwd_supported(
                    query.shape[0], feat1.shape[0], feat1.shape[1], feat1.shape[2], query.dtype)):
                dq_i, df_i = T.q2c_scores_l2norm_bwd(query, feat1, qn, cn_p, mk_p, dscores, scale=1.0 / n_mod, arg=arg)
                dq.append(dq_i)
                df.append(df_i)
                continue
            dqn, dcn = T.q2c_scores_bwd(qn, cn_p, mk_p, dscores, scale=1.0 / n_mod)
            if cn_p.shape[1] != feat1.shape[1]:
                dcn = dcn[:, :feat1.shape[1]].contiguous()
            dq.append(T.l2norm_bwd(query, dqn))
            df.append(T.l2norm_bwd(feat1, dcn))
        return (None,) + tuple(dq) + tuple(df) + (None,) * n_mod


class CombineLossFn(torch.autograd.Function):
    """loss = lw_st_ed * loss_st_ed + lw_neg_ctx * loss_neg_ctx + lw_neg_q * loss_neg_q (xml/model_xml.py:241-251) in one
    launch each way.  forward(st_ed 0-d or None, rank_losses (2,) or None, weights) -> (overall 0-d,
    parts (4,) [the three weighted terms, their sum], not differentiable)."""

    @staticmethod
    def forward(ctx, st_ed, rank2, w_st_ed, w_neg_ctx, w_neg_q):
        ctx.w = (float(w_st_ed), float(w_neg_ctx), float(w_neg_q))
        ctx.has = (st_ed is not None, rank2 is not None)
        parts, overall = T.loss_combine(None if st_ed is None else st_ed.contiguous(),
                                        None if rank2 is None else rank2.contiguous(), *ctx.w)
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)
        return overall, parts

    @staticmethod
    def backward(ctx, g, _g_parts):
        d0, d1 = T.loss_combine_bwd(g.float().contiguous(), *ctx.w, ctx.has[0] and ctx.needs_input_grad[0],
                                    ctx.has[1] and ctx.needs_input_grad[1])
        return d0, d1, None, None, None


class PairSimFn(torch.autograd.Function):
    """sim[b][l] = q[b] . f2[b][l]   (cross=False branch, xml/model_xml.py:478-479,532) -> f32."""

    @staticmethod
    def forward(ctx, q, f2):
        q, f2 = q.contiguous(), f2.contiguous()
        ctx.save_for_backward(q, f2)
        return T.pair_sim(q, f2)

    @staticmethod
    def backward(ctx, dsim):
        q, f2 = ctx.saved_tensors
        return T.pair_sim_bwd(q, f2, dsim.contiguous())


class SpanLossFn(torch.autograd.Function):
    """conv1d start / end predictors + mask_logits + F.cross_entropy on (N, L) similarities.
    forward(merged, ks, st_ed, n_sim, sims..., masks..., filters...) with filters = st filters then ed filters,
    each an nn.Conv1d weight (1, 1, ks)."""

    @staticmethod
    def forward(ctx, merged, ks, st_ed, n_sim, *rest):
        sims = [t.contiguous() for t in rest[:n_sim]]
        masks = [t.contiguous() for t in rest[n_sim:2 * n_sim]]
        filters = rest[2 * n_sim:]
        conv_w = torch.cat([f.detach().float().reshape(-1) for f in filters]).contiguous()
        ctx.cfg = (merged, ks, n_sim, [f.shape for f in filters])
        ctx.save_for_backward(conv_w, st_ed, *sims, *masks)
        return T.span_loss(sims, conv_w, masks, st_ed, merged, ks)

    @staticmethod
    def backward(ctx, gout):
        merged, ks, n_sim, fshapes = ctx.cfg
        conv_w, st_ed = ctx.saved_tensors[:2]
        sims = list(ctx.saved_tensors[2:2 + n_sim])
        masks = list(ctx.saved_tensors[2 + n_sim:])
        dsims, dconv = T.span_loss(sims, conv_w, masks, st_ed, merged, ks, gout=gout.reshape(1).float().contiguous())
        dfilters = [dconv[i * ks:(i + 1) * ks].reshape(s) for i, s in enumerate(fshapes)]
        return (None, None, None, None) + tuple(dsims) + (None,) * n_sim + tuple(dfilters)


class RankLossFn(torch.autograd.Function):
    """get_video_level_loss (xml/model_xml.py:588-637) -> (2,) f32 [loss_neg_ctx, loss_neg_q] (unweighted)."""

    @staticmethod
    def forward(ctx, scores, ranks_ctx, ranks_q, margin, lse):
        scores = scores.contiguous()
        ctx.cfg = (margin, lse)
        ctx.save_for_backward(scores, ranks_ctx, ranks_q)
        return T.rank_loss(scores, ranks_ctx, ranks_q, margin, lse)

    @staticmethod
    def backward(ctx, gout):
        scores, ranks_ctx, ranks_q = ctx.saved_tensors
        margin, lse = ctx.cfg
        return T.rank_loss(scores, ranks_ctx, ranks_q, margin, lse, gout=gout.float().contiguous()), None, None, None, None
