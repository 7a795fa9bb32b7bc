"""One XML training step on HIP kernels: forward graph, hand-written backward, BertAdam, gradient all-reduce.

Mirrors the reference's training inner loop (xml/train.py:62-111): `loss, loss_dict = model(**inputs)`;
`optimizer.zero_grad(); loss.backward(); [clip_grad_norm_]; optimizer.step()`.

  * `xml_forward_train(model, ...)` is XML.forward (xml/model_xml.py:212-251) assembled from the autograd nodes of
    autograd.py -- every forward and backward op is a libxmlhip.so kernel; torch autograd only keeps the tape.
  * `BertAdam` has the reference's constructor and schedule semantics (xml/optimization.py:219-338) but owns ONE flat
    f32 buffer for parameters / gradients / moments and updates it with two kernel launches per step.
  * `allreduce_gradients` averages the flat gradient buffer over ranks in size-bounded buckets (RCCL over xGMI,
    one process per GPU); the reference has no multi-process training (it wraps nn.DataParallel, xml/train.py:236).

Dropout: in `model.train()` mode the four dropout sites of the reference (LinearLayer input, positional encoding
output, attention probabilities, BertSelfOutput; xml/model_components.py:88,151,239,297,315) are applied by a
counter-based mask kernel (xml_dropout) seeded from torch's CPU generator -- statistically, not bitwise, the
reference's stream.  Parity is pinned in `model.eval()` mode on the training-step fixtures
(tests/golden/train_step_*.npz), which the reference also produced in eval mode.
"""
import math

import numpy as np
import torch

from . import ops
from . import train_ops as T
from . import autograd as _ag
from .autograd import (AttentionCoreFn, AttentionKvFn, AttentionQkvFn, CombineLossFn, DropoutFn, LayerNormFn, LinearFn, ModularPoolFn, PairSimFn, QkvResFn, RankLossFn,
                       QkvFn, SpanLossFn, VideoLevelScoresFn)

F32 = torch.float32


class _PosTableFn(torch.autograd.Function):
    """position_embeddings(arange(L)) broadcast over the batch (xml/model_components.py:83-87) in compute dtype;
    backward = column sums over the batch."""

    @staticmethod
    def forward(ctx, weight, n, seq_len, dtype):
        if seq_len > weight.shape[0]:
            raise IndexError("sequence length %d exceeds the positional table (%d)" % (seq_len, weight.shape[0]))
        from . import ops
        rows = weight.detach()[:seq_len].contiguous()
        rows = rows if dtype == F32 else ops.pack_weights(rows.float(), dtype)
        from .autograd import _claim
        ctx.shape = tuple(weight.shape)
        ctx.n, ctx.seq_len = n, seq_len
        ctx.params = (weight,)
        ctx.sunk = _claim(ctx, (weight,), (0,))
        return rows.unsqueeze(0).expand(n, seq_len, rows.shape[1]).contiguous()

    @staticmethod
    def backward(ctx, dy):
        from .autograd import _sink
        hidden = ctx.shape[1]
        weight, = ctx.params
        sw = _sink(weight) if ctx.sunk else None
        if sw is not None:          # column sums accumulated straight into the flat .grad buffer
            T.colsum(dy.contiguous(), ctx.n, ctx.seq_len * hidden, out=sw.view(-1)[:ctx.seq_len * hidden])
            return None, None, None, None
        dw = torch.zeros(ctx.shape, dtype=F32, device=dy.device)
        T.colsum(dy.contiguous(), ctx.n, ctx.seq_len * hidden, out=dw.view(-1)[:ctx.seq_len * hidden])
        return dw, None, None, None


_SITE = [0]      # dropout sites visited since the capture of a GraphedTrainStep began


def _seed():
    """A fresh 62-bit dropout seed from torch's CPU generator (so torch.manual_seed makes a run repeatable).  While a
    training step is being captured into a HIP graph the seed is a per-site constant instead; the kernels add the
    device-resident base seed that the graph itself advances (train_ops.SEED_BASE)."""
    if T.SEED_BASE is not None:
        _SITE[0] += 1
        return (_SITE[0] * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _drop(x, module):
    """module: an nn.Dropout holder of the parameter tree; identity in eval mode or with p == 0."""
    if not module.training or module.p <= 0:
        return x
    return DropoutFn.apply(x, float(module.p), _seed())


SHADOW_WEIGHTS = True    # bf16 steps: weights converted / transposed once per step by the optimizer (BertAdam.refresh_shadows)
FUSE_KV = True           # cross attention (bf16): key + value projections stacked into one GEMM (QkvFn) / one attention operand
FUSE_DROPOUT = True      # dropout sites next to a LayerNorm run inside the LayerNorm kernels (same masks, 22 launches fewer)


def _site(module):
    return (float(module.p), _seed()) if (module is not None and module.training and module.p > 0) else None


def _ln(a, b, norm, dt, drop_in=None, drop_out=None):
    """drop_out(LayerNorm(drop_in(a) + b)); drop_in / drop_out: the nn.Dropout holders of the two sites, or None."""
    si, so = _site(drop_in), _site(drop_out)
    need_dx = a.requires_grad or (b is not None and b.requires_grad)
    if (si or so) and FUSE_DROPOUT and T.layernorm_drop_supported(a.shape[-1], dt, need_dx, b is not None,
                                                                  si[0] if si else 0.0):
        return LayerNormFn.apply(a, b, norm.weight, norm.bias, dt, si, so)
    if si:
        a = DropoutFn.apply(a, *si)
    y = LayerNormFn.apply(a, b, norm.weight, norm.bias, dt)
    return DropoutFn.apply(y, *so) if so else y


def _probs_drop(sa):
    return (float(sa.dropout.p), _seed()) if (sa.training and sa.dropout.p > 0) else (0.0, 0)


def _bert_attention(mod, x, key_mask, dt):
    """BertAttention = BertSelfAttention + BertSelfOutput (xml/model_components.py:201-216,313-317)."""
    sa, so = mod.self, mod.output
    wb = (sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias)
    if RESIDUAL_THROUGH_QKV and x.requires_grad and x.is_cuda:
        qkv, x = QkvResFn.apply(x, *wb)      # (x: the same values, its gradient routed through the projection's backward node)
    else:
        qkv = QkvFn.apply(x, *wb)
    a = AttentionQkvFn.apply(qkv, key_mask, sa.num_attention_heads, *_probs_drop(sa))
    return _ln(LinearFn.apply(a, so.dense.weight, so.dense.bias, False), x, so.LayerNorm, dt, drop_in=so.dropout)


def _encode_input(model, feat, mask, proj, enc, pos):
    """encode_input (xml/model_xml.py:377-392)."""
    dt = model.compute_dtype
    if feat.dtype not in (F32, dt):
        feat = feat.float()
    n, seq_len = feat.shape[:2]
    x = _ln(feat.contiguous(), None, proj.LayerNorm, dt, drop_out=proj.net[0])
    x = LinearFn.apply(x, proj.net[1].weight, proj.net[1].bias, True)
    p = _PosTableFn.apply(pos.position_embeddings.weight, n, seq_len, dt)
    x = _ln(x, p, pos.LayerNorm, dt, drop_out=pos.dropout)
    return _bert_attention(enc, x, mask, dt)


def _cross_context(model, main, main_mask, side, side_mask, cross, norm, self_att):
    """cross_context_encoder (xml/model_xml.py:357-373)."""
    dt = model.compute_dtype
    q = LinearFn.apply(main, cross.query.weight, cross.query.bias, False)
    heads, hidden = cross.num_attention_heads, main.shape[2]
    if FUSE_KV and main.is_cuda and T.attention_train_supported(main.shape[1], side.shape[1], hidden, heads, dt):
        # key and value projections of the other stream as one GEMM each way (and one gradient fewer to add into `side`)
        kv = QkvFn.apply(side, cross.key.weight, cross.key.bias, cross.value.weight, cross.value.bias)
        a = AttentionKvFn.apply(q, kv, main_mask, side_mask, heads, *_probs_drop(cross))
    else:
        k = LinearFn.apply(side, cross.key.weight, cross.key.bias, False)
        v = LinearFn.apply(side, cross.value.weight, cross.value.bias, False)
        a = AttentionCoreFn.apply(q, k, v, main_mask, side_mask, heads, *_probs_drop(cross))
    res = LayerNormFn.apply(a, main, norm.weight, norm.bias, dt)
    return _bert_attention(self_att, res, main_mask, dt)


RESIDUAL_THROUGH_QKV = True     # the residual gradient of a BertAttention block as the addend of its dX GEMM (autograd.QkvResFn)
QUERY_FIRST = False         # measured and not kept (see xml_forward_train): 4.13 vs 3.96 ms per captured step
PARALLEL_BRANCHES = True      # video / subtitle branches of the training graph on two HIP streams
_SIDE_STREAMS = {}


def _side_stream(dev, which=0):
    key = (torch.device(dev).index, which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def encode_context_train(model, video_feat, video_mask, sub_feat, sub_mask):
    """encode_context (xml/model_xml.py:331-355,297-329) with gradient tape."""
    cfg = model.config
    dt = model.compute_dtype
    if cfg.cross_att:
        enc_v = lambda: _encode_input(model, video_feat, video_mask, model.video_input_proj, model.video_encoder1,    # noqa: E731
                                      model.ctx_pos_embed)
        enc_s = lambda: _encode_input(model, sub_feat, sub_mask, model.sub_input_proj, model.sub_encoder1,           # noqa: E731
                                      model.ctx_pos_embed)
        if PARALLEL_BRANCHES and video_feat.is_cuda:
            # The video and the subtitle stream are independent until the cross-attention, and again after it.  At the C5
            # shape one projection is 150 tiles of 256 x 256 on 256 CUs: issued on TWO HIP streams, the two branches' kernels
            # share the chip (and autograd runs each backward node on its forward stream, so the backward pass overlaps the
            # same way).  In a captured step (GraphedTrainStep) the two streams are parallel branches of the graph.
            dev = video_feat.device
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(main)
            # The caller's tensors were allocated on ITS stream but are read by kernels on the side stream, in this pass and --
            # as saved tensors -- in the backward pass.  Tell the caching allocator: a caller that drops its batch before the
            # backward kernels have run (a loop that loads the next batch, a test that passes temporaries) would otherwise get
            # the block back at once and overwrite it under the side stream's feet (seen as an intermittently wrong
            # sub_input_proj.LayerNorm.weight gradient: dgamma from overwritten features, dbeta right).
            if not torch.cuda.is_current_stream_capturing():
                for t in (sub_feat, sub_mask, video_mask):
                    if t is not None and t.is_cuda:
                        t.record_stream(side)
            ev = enc_v()
            with torch.cuda.stream(side):
                es = enc_s()
            main.wait_stream(side)
            side.wait_stream(main)
            es.record_stream(main)
            ev.record_stream(side)
            xv = _cross_context(model, ev, video_mask, es, sub_mask, model.video_cross_att, model.video_cross_layernorm,
                                model.video_encoder2)
            with torch.cuda.stream(side):
                xs = _cross_context(model, es, sub_mask, ev, video_mask, model.sub_cross_att, model.sub_cross_layernorm,
                                    model.sub_encoder2)
            main.wait_stream(side)
            xs.record_stream(main)
            return ev, xv, es, xs
        ev, es = enc_v(), enc_s()
        xv = _cross_context(model, ev, video_mask, es, sub_mask, model.video_cross_att, model.video_cross_layernorm,
                            model.video_encoder2)
        xs = _cross_context(model, es, sub_mask, ev, video_mask, model.sub_cross_att, model.sub_cross_layernorm,
                            model.sub_encoder2)
        return ev, xv, es, xs
    out = []
    for name, use, feat, mask in (("video", model.use_video, video_feat, video_mask),
                                  ("sub", model.use_sub, sub_feat, sub_mask)):
        if not use:
            out += [None, None]
            continue
        f1 = _encode_input(model, feat, mask, getattr(model, name + "_input_proj"), getattr(model, name + "_encoder1"),
                           model.ctx_pos_embed)
        f2 = _bert_attention(getattr(model, name + "_encoder2"), f1, mask, dt)
        f2 = _bert_attention(getattr(model, name + "_encoder3"), f2, mask, dt)
        out += [f1, f2]
    return tuple(out)


def draw_negative_ranks(model, bsz):
    """The two torch.randint draws of get_neg_scores (xml/model_xml.py:608-624), in the reference's call order
    (negative contexts first, then negative queries), on the CPU generator like the reference."""
    cfg = model.config
    hi = min(1 + cfg.hard_pool_size, bsz) if cfg.use_hard_negative else bsz
    return torch.randint(1, hi, size=(bsz,)), torch.randint(1, hi, size=(bsz,))


def xml_forward_train(model, query_feat, query_mask, video_feat, video_mask, sub_feat, sub_mask, st_ed_indices,
                      neg_ctx_rank=None, neg_q_rank=None, as_tensors=False):
    """XML.forward (xml/model_xml.py:212-251) -> (loss 0-d tensor with grad_fn, loss dict of floats).
    as_tensors=True: the dict holds detached 0-d device tensors instead of floats (no host synchronisation: what a
    captured step needs)."""
    cfg = model.config
    dev = query_feat.device
    if model.compute_dtype is ops.F16S:
        raise ValueError("compute_dtype=ops.F16S is inference-only (the exact-rank mode's split-f16 model): train in "
                         "torch.bfloat16 or torch.float32 and load the checkpoint into an ops.F16S model")
    fm = lambda m: None if m is None else m.float().contiguous()       # noqa: E731
    query_mask, video_mask, sub_mask = fm(query_mask), fm(video_mask), fm(sub_mask)
    if SHADOW_WEIGHTS and model.compute_dtype == torch.bfloat16 and torch.is_grad_enabled():
        reg = getattr(next(iter(model.parameters())), "_xml_sink", None)
        if reg is not None:             # an optimizer owns the parameters: its bf16 / transposed weight copies for this step
            # (two-stream pass: the transposed copies are not needed before the backward pass -- onto the side stream, which
            # carries the shorter subtitle branch)
            two = PARALLEL_BRANCHES and cfg.cross_att and video_feat is not None and video_feat.is_cuda
            reg.opt.refresh_shadows(model.compute_dtype, transpose_stream=_side_stream(video_feat.device) if two else None)
    # QUERY_FIRST (off): the query encoder BEFORE the context branches.  The backward pass runs its nodes latest-created first,
    # so the two-stream context backward would start right behind the loss chain and the query side's ~15 small backward kernels
    # -- leaves that only the optimizer waits for -- would follow the video branch instead of standing between the loss chain and
    # the fork.  Measured, same box: 4.13 vs 3.96 ms per captured step -- the small kernels then end the step alone on the chip.
    def encode_query_side():
        e = _encode_input(model, query_feat, query_mask, model.query_input_proj, model.query_encoder, model.query_pos_embed)
        return ModularPoolFn.apply(e, query_mask, model.modular_vector_mapping.weight)
    mq = encode_query_side() if QUERY_FIRST else None
    v1, v2, s1, s2 = encode_context_train(model, video_feat, video_mask, sub_feat, sub_mask)
    # (the query encoder on a THIRD stream was measured and not kept: 5.54 vs 5.04 ms per captured step -- its small kernels
    # then interleave with the two context branches and break up their pairing; round 4: the same BEHIND the subtitle branch
    # on the side stream, which finishes ~100 us ahead of the video branch: 4.99-5.01 vs 4.32-4.33 ms, same box)
    if mq is None:
        mq = encode_query_side()
    # (unbind, not mq[0] / mq[1]: its backward is one stack instead of two zero-fills, two copies and an add)
    video_query, sub_query = mq.unbind(0) if mq.shape[0] == 2 else (mq[0], mq[0])

    names = [n for n, u in (("video", model.use_video), ("sub", model.use_sub)) if u]
    qs = dict(video=video_query, sub=sub_query)
    f1 = dict(video=v1, sub=s1)
    f2 = dict(video=v2, sub=s2)
    ms = dict(video=video_mask, sub=sub_mask)
    bsz = query_feat.shape[0]
    zero = torch.zeros((), dtype=F32, device=dev)

    loss_st_ed = zero
    if cfg.lw_st_ed != 0:
        sims = []
        for n in names:
            lin = getattr(model, n + "_query_linear")
            sims.append(PairSimFn.apply(LinearFn.apply(qs[n], lin.weight, lin.bias, False), f2[n]))
        merged = bool(cfg.merge_two_stream and len(names) == 2)
        if merged:
            masks = [ms["video"], ms["video"]]
            filters = [model.merged_st_predictor.weight, model.merged_ed_predictor.weight]
        else:
            masks = [ms[n] for n in names]
            filters = [getattr(model, n + "_st_predictor").weight for n in names] + \
                      [getattr(model, n + "_ed_predictor").weight for n in names]
        loss_st_ed = SpanLossFn.apply(merged, cfg.conv_kernel_size, st_ed_indices.long().contiguous(), len(sims),
                                      *sims, *masks, *filters)

    loss_neg_ctx = loss_neg_q = zero
    losses = None
    if cfg.lw_neg_ctx != 0 or cfg.lw_neg_q != 0:
        q2c = VideoLevelScoresFn.apply(len(names), *[qs[n] for n in names], *[f1[n] for n in names],
                                       *[ms[n] for n in names])
        if neg_ctx_rank is None or neg_q_rank is None:
            neg_ctx_rank, neg_q_rank = draw_negative_ranks(model, bsz)
        to_dev = lambda r: torch.as_tensor(r).to(device=dev, dtype=torch.int32).contiguous()   # noqa: E731
        if cfg.ranking_loss_type not in ("hinge", "lse"):
            raise NotImplementedError("Only support 'hinge' and 'lse'")
        losses = RankLossFn.apply(q2c, to_dev(neg_ctx_rank), to_dev(neg_q_rank), float(cfg.margin),
                                  cfg.ranking_loss_type == "lse")
        loss_neg_ctx, loss_neg_q = losses[0], losses[1]

    if _ag.FUSED_LOSS_TAIL and dev.type == "cuda" and (cfg.lw_st_ed != 0 or losses is not None):
        # the weighted sum and its backward as one launch each (a dozen scalar torch kernels in the serial middle of the step)
        loss, parts = CombineLossFn.apply(loss_st_ed if cfg.lw_st_ed != 0 else None, losses, cfg.lw_st_ed, cfg.lw_neg_ctx,
                                          cfg.lw_neg_q)
        if as_tensors:
            return loss, {"loss_st_ed": parts[0], "loss_neg_ctx": parts[1], "loss_neg_q": parts[2], "loss_overall": parts[3]}
        vals = parts.tolist()                # one host read for the four numbers
        return loss, {"loss_st_ed": vals[0], "loss_neg_ctx": vals[1], "loss_neg_q": vals[2], "loss_overall": vals[3]}
    loss_st_ed = cfg.lw_st_ed * loss_st_ed
    loss_neg_ctx = cfg.lw_neg_ctx * loss_neg_ctx
    loss_neg_q = cfg.lw_neg_q * loss_neg_q
    loss = loss_st_ed + loss_neg_ctx + loss_neg_q
    if as_tensors:
        f = lambda t: t.detach() if torch.is_tensor(t) else torch.full((), float(t), dtype=F32, device=dev)   # noqa: E731
    else:
        f = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)      # noqa: E731
    return loss, {"loss_st_ed": f(loss_st_ed), "loss_neg_ctx": f(loss_neg_ctx), "loss_neg_q": f(loss_neg_q),
                  "loss_overall": f(loss)}


# ---------------------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------------------
def _sched_warmup_linear(progress, warmup):
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


def _sched_warmup_constant(progress, warmup):
    return progress / warmup if progress < warmup else 1.0


def _sched_warmup_cosine(progress, warmup, cycles=0.5):
    if progress < warmup:
        return progress / warmup
    progress = (progress - warmup) / (1 - warmup)
    return 0.5 * (1.0 + math.cos(math.pi * cycles * 2 * progress))


_SCHEDULES = {None: None, "none": None, "warmup_linear": _sched_warmup_linear,
              "warmup_constant": _sched_warmup_constant, "warmup_cosine": _sched_warmup_cosine}


class BertAdam(object):
    """BERT Adam with decoupled weight decay, per-tensor gradient clipping and no bias correction
    (xml/optimization.py:219-338), fused over a flat buffer.

    `params`: iterable of parameters or of param-group dicts ({"params": [...], "weight_decay": ..., "lr": ...}) as in
    the reference (xml/train.py:355-362).  At construction every parameter is re-pointed into one flat f32 device
    buffer (`.data` becomes a view) and gets a persistent `.grad` view into a flat gradient buffer, so that
    zero_grad / all-reduce / step each touch one allocation."""

    def __init__(self, params, lr, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in _SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        params = list(params)
        groups = params if params and isinstance(params[0], dict) else [{"params": params}]
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            self.param_groups.append(g)
        self.defaults = dict(lr=lr, warmup=max(warmup, 0.0), t_total=float(t_total), schedule=schedule, b1=b1, b2=b2,
                             e=e, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.step_count = 0
        self._flatten()

    def _flatten(self):
        plist, lrs, wds = [], [], []
        for g in self.param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda or p.dtype != F32:
                    raise RuntimeError("BertAdam: parameters must be f32 tensors on the GPU (no CPU path)")
                plist.append(p)
                lrs.append(g["lr"])
                wds.append(g["weight_decay"])
        if not plist:
            raise ValueError("optimizer got an empty parameter list")
        dev = plist[0].device
        offs = [0]
        for p in plist:
            offs.append(offs[-1] + (p.numel() + 3) // 4 * 4)        # keep every tensor 16-byte aligned
        total = offs[-1]
        self.flat_p = torch.zeros(total, dtype=F32, device=dev)
        self.flat_g = torch.zeros(total, dtype=F32, device=dev)
        self.flat_m = torch.zeros(total, dtype=F32, device=dev)
        self.flat_v = torch.zeros(total, dtype=F32, device=dev)
        with torch.no_grad():
            for p, o in zip(plist, offs):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.params = plist
        self._offs = offs
        self._shadow = None                  # compute-dtype copies of the weights, refreshed once per step (refresh_shadows)
        self.seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_lr = torch.tensor(lrs, dtype=F32, device=dev)
        self.seg_wd = torch.tensor(wds, dtype=F32, device=dev)
        self.norms = torch.zeros(len(plist), dtype=F32, device=dev)
        self.norm_ws = torch.zeros((total + 1023) // 1024, dtype=F32, device=dev)      # per-block partial sums of g^2
        # `if p.grad is None: continue` + per-tensor state['step'] (xml/optimization.py:289-291,325-330): a tensor takes
        # part in a step once it has EVER received a gradient (the reference's pinned torch 1.4 zero_grad() zeroes
        # existing .grad tensors in place, it does not reset them to None); its schedule counts its own steps.
        # Gradients land in persistent views of the flat buffer, so "received a gradient" is observed by a
        # post-accumulate hook per parameter (host flag only, no device work).
        self._touched = [False] * len(plist)
        self._reducer = None                 # GradientReducer (data-parallel runs): all-reduce overlapped with backward
        self.seg_steps = [0] * len(plist)
        self._active_key = None
        self._active_dev = None
        from . import autograd as _ag
        from .autograd import GradSink
        if _ag.USE_GRAD_SINKS and not _ag.hook_fires_on_undefined_grad():
            import warnings
            warnings.warn("this torch does not run post-accumulate hooks for undefined gradients: gradient sinks disabled "
                          "(backward nodes return their parameter gradients to autograd)")
            _ag.USE_GRAD_SINKS = False
        for i, (p, o) in enumerate(zip(plist, offs)):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touch(i))
            p._xml_sink = GradSink(self, i, o)      # backward kernels may accumulate straight into p.grad (autograd.py)

    def _touch(self, i):
        """Post-accumulate hook of parameter i: every node that uses it has run its backward (its share of the gradient is
        in the flat buffer -- added by AccumulateGrad, or already accumulated there by the node itself, autograd.py sinks)."""
        self._touched[i] = True
        if self._reducer is not None:
            self._reducer.grad_ready(i)

    # ---- weight shadows ------------------------------------------------------------------------------------------------
    # The bf16 training step used to convert every f32 master weight to bf16 in the forward pass and transpose it again for
    # the dX GEMM of the backward pass: 24 + 21 launches, 0.33 ms of the 5 ms step at the C5 shape, all of them pure
    # functions of the flat parameter buffer.  refresh_shadows (called by xml_forward_train at the start of a step) makes
    # ONE bf16 copy of the whole buffer and ONE launch that writes every transposed matrix the backward nodes have asked for
    # (shadow_t registers a matrix the first time it is wanted; it is served from the following refresh on).
    # A shadow is only handed out while it is current: refreshed since the last step(), and no parameter it covers has been
    # written through torch since (Tensor._version -- load_state_dict, manual edits); otherwise the caller packs afresh.
    def refresh_shadows(self, dtype, transpose_stream=None):
        """transpose_stream: a stream that the caller forks from and joins back into the current one BEFORE the backward pass
        (encode_context_train's side stream): the transposed copies -- read by the dX GEMMs of the backward pass only -- are
        written there, off the head of the step's critical path."""
        sh = self._shadow
        if sh is None or sh["dtype"] != dtype:
            sh = self._shadow = dict(dtype=dtype, flat=torch.empty(self.flat_p.numel(), dtype=dtype, device=self.flat_p.device),
                                     t={}, ents={}, table=None, tables=[], max_tiles=0, dirty=False, fresh=False, versions=None,
                                     gen=0)
        # a new generation whenever the masters may have moved since the last refresh (step(), a write through torch); a
        # second forward on the SAME weights (gradient accumulation) re-packs the same values and keeps the generation.
        # (A captured refresh is replayed with its own forward AND backward: nothing to guard.)
        if not torch.cuda.is_current_stream_capturing() and \
                not (sh["fresh"] and sh["versions"] == [p._version for p in self.params]):
            sh["gen"] += 1
        ops.pack_weights(self.flat_p, dtype, out=sh["flat"])
        if sh["dirty"] and not torch.cuda.is_current_stream_capturing():
            rows, tiles = [], 0
            for idx, ent in sh["t"].items():
                col = 0
                for i in idx:
                    n, k = self.params[i].shape
                    rows.append([self._offs[i], n, k, ent["buf"].data_ptr(), ent["buf"].shape[1], col])
                    tiles = max(tiles, ((n + 63) // 64) * ((k + 63) // 64))
                    col += n
                ent["ready"] = True
            sh["table"] = torch.tensor(rows, dtype=torch.int64, device=self.flat_p.device)
            sh["tables"].append(sh["table"])      # a captured step keeps reading the table it was captured with
            sh["max_tiles"], sh["dirty"] = tiles, False
        if sh["table"] is not None:
            if transpose_stream is not None:
                transpose_stream.wait_stream(torch.cuda.current_stream(self.flat_p.device))      # the masters as of now
                with torch.cuda.stream(transpose_stream):
                    T.transpose_segments(self.flat_p, sh["table"], sh["max_tiles"])
            else:
                T.transpose_segments(self.flat_p, sh["table"], sh["max_tiles"])
        sh["versions"] = [p._version for p in self.params]
        sh["fresh"] = True

    def _shadow_entry(self, params):
        """Resolve (once) the parameter tuple of a node: its indices here when the tensors lie back to back in the flat
        buffer, in this order; False otherwise."""
        sh = self._shadow
        key = tuple(map(id, params))
        ent = sh["ents"].get(key)
        if ent is None:
            ent = False
            regs = [getattr(p, "_xml_sink", None) for p in params]
            if all(r is not None and r.opt is self for r in regs):
                idx = tuple(r.index for r in regs)
                off, ok = self._offs[idx[0]], True
                for p, i in zip(params, idx):
                    ok = ok and self._offs[i] == off and self.params[i] is p
                    off += p.numel()
                if ok:
                    base = self.flat_p.data_ptr()
                    ent = dict(params=tuple(params), idx=idx, ptrs=tuple(base + 4 * self._offs[i] for i in idx),
                               view=sh["flat"][self._offs[idx[0]]:off].view(-1, params[0].shape[-1]), t=None)
            sh["ents"][key] = ent
        return ent

    @staticmethod
    def _shadow_current(sh, ent):
        # refreshed since the last step(), the tensors still ARE their slices of the flat buffer (model.to() / .float() move
        # them away), and nothing has written them through torch since the refresh
        vers = sh["versions"]
        for p, i, ptr in zip(ent["params"], ent["idx"], ent["ptrs"]):
            if p._version != vers[i] or p.data_ptr() != ptr:
                return False
        return sh["fresh"]

    def shadow_generation(self):
        """Which refresh the shadow buffers hold (autograd._shadow_guard: a backward must see the refresh its forward saw)."""
        return self._shadow["gen"] if self._shadow is not None else 0

    def shadow_w(self, params, dtype):
        """The compute-dtype copy (sum N, K) of the row-concatenated parameters, or None when there is no current one."""
        sh = self._shadow
        if sh is None or not sh["fresh"] or sh["dtype"] != dtype:
            return None
        ent = self._shadow_entry(params)
        return ent["view"] if ent and self._shadow_current(sh, ent) else None

    def shadow_t(self, params, dtype):
        """The transposed copy (K, ceil8(sum N)) of the row-concatenated 2-D parameters, or None -- registering it for the
        refreshes to come when it is not kept yet."""
        sh = self._shadow
        if sh is None or sh["dtype"] != dtype:
            return None
        ent = self._shadow_entry(params)
        if not ent:
            return None
        t = sh["t"].get(ent["idx"])
        if t is None:
            if not torch.cuda.is_current_stream_capturing():
                n, k = sum(p.shape[0] for p in params), params[0].shape[1]
                sh["t"][ent["idx"]] = dict(buf=torch.zeros((k, (n + 7) // 8 * 8), dtype=dtype, device=self.flat_p.device),
                                           ready=False)
                sh["dirty"] = True
            return None
        return t["buf"] if t["ready"] and self._shadow_current(sh, ent) else None

    def zero_grad(self):
        if self._reducer is not None:
            self._reducer.begin()
        self.flat_g.zero_()
        base = self.flat_g.data_ptr()
        for p, o in zip(self.params, self._offs):           # (host offsets: seg_off.tolist() was a device sync per step)
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def lr_multiplier(self, step=None):
        """_LRSchedule.get_lr (xml/optimization.py:40-55)."""
        d = self.defaults
        fn = _SCHEDULES[d["schedule"]]
        if fn is None or d["t_total"] < 0:
            return 1.0
        return fn(float(self.step_count if step is None else step) / d["t_total"], d["warmup"])

    def get_lr(self):
        """BertAdam.get_lr (xml/optimization.py:255-267): [0] before any tensor has stepped, else lr * schedule(step) per
        tensor (its own step count)."""
        it = iter(range(len(self.params)))
        out = []
        for g in self.param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                i = next(it)
                if self.seg_steps[i] == 0:        # `if len(state) == 0: return [0]` -- a tensor that has not stepped yet
                    return [0]
                out.append(g["lr"] * self.lr_multiplier(self.seg_steps[i]))
        return out

    def _step_plan(self):
        """Host side of a step: which tensors take part (`if p.grad is None: continue`) and their schedule multipliers
        (per-tensor step counts, xml/optimization.py:289-291,325-330)."""
        # no hook has ever fired = gradients are being written into .grad by hand (no autograd): every tensor "has a grad"
        active = self._touched if any(self._touched) else [True] * len(self._touched)
        per = {st: self.lr_multiplier(st) for st in {s_ for s_, a in zip(self.seg_steps, active) if a}}
        mults = [per[s_] if a else 0.0 for s_, a in zip(self.seg_steps, active)]
        return active, mults

    def _active_mask(self, active):
        if all(active):
            return None
        key = tuple(active)
        if key != self._active_key:      # uploaded only when the set of participating tensors changes
            self._active_key = key
            self._active_dev = torch.tensor([1 if a else 0 for a in active], dtype=torch.uint8, device=self.flat_p.device)
        return self._active_dev

    def _commit_step(self, active):
        for i, a in enumerate(active):
            if a:
                self.seg_steps[i] += 1
        self.step_count += 1
        if self._shadow is not None:
            self._shadow["fresh"] = False       # the masters have moved
        from .model_xml import _PackedMixin
        _PackedMixin.bump_generation()          # cached low-precision weight copies are stale now

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        d = self.defaults
        active, mults = self._step_plan()
        seg_active = self._active_mask(active)
        distinct = {m for m, a in zip(mults, active) if a}
        seg_mult = None
        if len(distinct) <= 1:               # every participating tensor is at the same step: one scalar multiplier
            mult = distinct.pop() if distinct else self.lr_multiplier(0)
        else:                                # tensors that joined later run their own warm-up
            mult = 0.0
            seg_mult = torch.tensor(mults, dtype=F32, device=self.flat_p.device)
        T.bert_adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.seg_off, self.seg_lr, self.seg_wd,
                         self.norms, mult, d["b1"], d["b2"], d["e"], d["max_grad_norm"], seg_active, seg_mult,
                         norm_ws=self.norm_ws)
        self._commit_step(active)
        return loss


class GraphedTrainStep(object):
    """One training iteration of the reference's loop (xml/train.py:78-95: forward, zero_grad, backward, [clip], step)
    captured ONCE into a HIP graph and replayed.

    Why: after the kernel work of rounds 2-3 the C5-shape step (batch 128, bf16) is ~300 launches and ~6 ms of kernel time,
    and the Python / autograd / ctypes side of issuing them costs about as much: the step had become CPU-bound.  One graph
    launch replaces all of it.  What makes the capture faithful:
      * inputs live in static device buffers (`__call__` copies the batch in; shapes are fixed at construction);
      * the in-batch negatives of get_neg_scores are drawn per step on the CPU generator, in the reference's order
        (draw_negative_ranks), and copied into static index tensors before the replay;
      * dropout seeds frozen into the graph's nodes are per-site constants; the kernels add a device-resident base seed
        that the FIRST node of the graph advances (train_ops.SEED_BASE): fresh masks on every replay, forward and backward
        of one replay see the same ones;
      * the learning-rate schedule reaches the optimizer kernel as a per-tensor multiplier array in device memory, written
        by the host before each replay (per-tensor step counts, warm-up, `p.grad is None` semantics: BertAdam._step_plan);
      * every fill inside the captured entries is a kernel (a memset NODE did not re-run on replay, ROCm 7.2).
    The warm-up steps run on copies of the optimizer state, which is restored before the capture: constructing the object
    does not train.  Data-parallel runs: attach the GradientReducer BEFORE constructing this object; its bucketed
    all-reduces are captured too (see __init__).  Not supported: changing which tensors receive gradients after the capture
    (set_train_st_ed: re-create the object).
    Returns (loss, loss_dict) as 0-d DEVICE tensors; reading them synchronises."""

    def __init__(self, model, optimizer, batch, grad_clip=-1, warmup_steps=2):
        # data-parallel runs: with a GradientReducer attached, its bucketed all-reduces (xml_rccl_allreduce_avg_f32 on the
        # reducer's side stream, ordered by events behind the backward kernels of each bucket) are captured as parallel
        # branches of the same graph -- the overlap with backward is part of every replay.  Every rank must construct the
        # object at the same point (the warm-up steps and the capture issue collectives).
        self.model, self.opt, self.grad_clip = model, optimizer, grad_clip
        dev = optimizer.flat_p.device
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in batch.items()
                       if k not in ("neg_ctx_rank", "neg_q_rank")}
        n = self.static["query_feat"].shape[0]
        # the host's per-step inputs -- the two negative-rank vectors and the schedule multipliers -- live in ONE device buffer
        # and arrive in ONE copy from pinned memory (three pageable copies, each behind its own host work, left the device
        # idle for ~125 us between two replays)
        nseg = len(optimizer.params)
        self._host_in = torch.ones(2 * n + nseg, dtype=torch.int32, device=dev)
        self.neg_ctx, self.neg_q = self._host_in[:n], self._host_in[n:2 * n]
        self.lr_mult = self._host_in[2 * n:].view(F32)
        self.lr_mult.fill_(1.0)
        pin = dev.type == "cuda"
        self._stage = [torch.ones(2 * n + nseg, dtype=torch.int32, pin_memory=pin) for _ in range(4)]
        self._stage_np = [t.numpy() for t in self._stage]      # (host-side fills go through numpy: ~1 us per slice, not ~8)
        self._stage_ev = [None] * 4
        self._stage_i = 0
        self._one = torch.ones((), dtype=F32, device=dev)
        self.seed_base = torch.zeros(1, dtype=torch.int64, device=dev)
        # ---- warm-up on the side stream (workspaces, allocator pools, packed-weight caches), state restored afterwards
        keep = [t.clone() for t in (optimizer.flat_p, optimizer.flat_m, optimizer.flat_v)]
        keep_host = (list(optimizer.seg_steps), optimizer.step_count, list(optimizer._touched))
        cpu_rng = torch.get_rng_state()
        # data-parallel capture: one stream (see GradientReducer._reduce) -- the branch streams are switched off for the
        # warm-up and the capture of THIS object; what is lost is the 10 % the two-stream capture gains on one GPU
        global PARALLEL_BRANCHES
        self._branches_were = PARALLEL_BRANCHES
        T.SEED_BASE = None
        try:        # (everything that can fail -- warm-up OOM, a collective error, the capture -- restores the module switches)
            if optimizer._reducer is not None:
                PARALLEL_BRANCHES = False
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup_steps)):
                    self._set_ranks(None, None)
                    self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self._restore(keep)
            optimizer.seg_steps, optimizer.step_count = list(keep_host[0]), keep_host[1]
            touched_after_warmup = list(optimizer._touched)     # which tensors this graph's backward reaches
            torch.set_rng_state(cpu_rng)
            # ---- capture
            self.active = touched_after_warmup if any(touched_after_warmup) else [True] * len(touched_after_warmup)
            self.seg_active = optimizer._active_mask(self.active)
            self._active_np = np.asarray(self.active, dtype=bool)
            self.graph = torch.cuda.CUDAGraph()
            T.SEED_BASE = self.seed_base
            _SITE[0] = 0
            with torch.cuda.graph(self.graph):
                self.seed_base.add_(0x2545F4914F6CDD1D)          # first node: this replay's base seed
                self.loss, self.parts = self._body(captured=True)
        finally:
            T.SEED_BASE = None
            PARALLEL_BRANCHES = self._branches_were
        optimizer._touched = [a or b for a, b in zip(keep_host[2], self.active)]
        torch.cuda.synchronize(dev)
        # the capture itself executed nothing; bring the optimizer state back in any case (allocator reuse)
        self._restore(keep)

    def _restore(self, keep):
        """Optimizer state back to the copies taken before the warm-up -- and the per-step weight shadows marked stale: they
        hold the compute-dtype weights of the LAST WARM-UP STEP (one update away from the restored masters), the copy below
        does not bump the parameters' versions, and during the capture refresh_shadows only RECORDS a refresh.  Without this an
        eager forward between construction and the first replay (a validation loss) would silently read those weights."""
        opt = self.opt
        for dst, src in zip((opt.flat_p, opt.flat_m, opt.flat_v), keep):
            dst.copy_(src)
        if opt._shadow is not None:
            opt._shadow["fresh"] = False

    def _set_ranks(self, neg_ctx_rank, neg_q_rank, lr_mults=None):
        """Stages this replay's host inputs and sends them in one copy.  lr_mults None: the multipliers stay as they are."""
        n = self.neg_ctx.shape[0]
        if neg_ctx_rank is None or neg_q_rank is None:
            neg_ctx_rank, neg_q_rank = draw_negative_ranks(self.model, n)
        i = self._stage_i
        self._stage_i = (i + 1) % len(self._stage)
        if self._stage_ev[i] is not None:
            self._stage_ev[i].synchronize()        # the copy that last read this pinned buffer (four calls ago) is done
        h, hn = self._stage[i], self._stage_np[i]
        hn[:n] = neg_ctx_rank.detach().cpu().numpy() if torch.is_tensor(neg_ctx_rank) else np.asarray(neg_ctx_rank)
        hn[n:2 * n] = neg_q_rank.detach().cpu().numpy() if torch.is_tensor(neg_q_rank) else np.asarray(neg_q_rank)
        if lr_mults is None:
            self._host_in[:2 * n].copy_(h[:2 * n], non_blocking=True)
        else:
            hn[2 * n:].view(np.float32)[:] = lr_mults
            self._host_in.copy_(h, non_blocking=True)
        if h.is_pinned():
            self._stage_ev[i] = torch.cuda.Event()
            self._stage_ev[i].record()

    def _body(self, captured=False):
        opt = self.opt
        loss, parts = xml_forward_train(self.model, neg_ctx_rank=self.neg_ctx, neg_q_rank=self.neg_q, as_tensors=True,
                                        **self.static)
        if opt._reducer is not None:
            opt._reducer.begin()
        opt.flat_g.zero_()
        loss.backward(self._one)            # (a resident 1.0: no ones_like fill node in front of the backward pass)
        if opt._reducer is not None:
            opt._reducer.finish()           # leftover buckets + join the all-reduce stream
        if self.grad_clip != -1:
            T.clip_grad_norm(opt.flat_g, self.grad_clip)
        d = opt.defaults
        if captured:
            T.bert_adam_step(opt.flat_p, opt.flat_g, opt.flat_m, opt.flat_v, opt.seg_off, opt.seg_lr, opt.seg_wd, opt.norms,
                             0.0, d["b1"], d["b2"], d["e"], d["max_grad_norm"], self.seg_active, self.lr_mult,
                             norm_ws=opt.norm_ws)
        else:
            opt.step()
        return loss.detach(), parts

    def __call__(self, batch=None, neg_ctx_rank=None, neg_q_rank=None):
        """batch: dict of the XML.forward keyword arguments with the shapes given at construction (None: keep the resident
        batch).  neg_*_rank: inject the in-batch negatives (parity tests); default: drawn like the reference draws them."""
        if batch is not None:
            for k, v in batch.items():
                if k in self.static and torch.is_tensor(v):
                    if tuple(v.shape) != tuple(self.static[k].shape):
                        raise ValueError("GraphedTrainStep was captured for %s %s, got %s" % (k, tuple(self.static[k].shape),
                                                                                              tuple(v.shape)))
                    self.static[k].copy_(v, non_blocking=True)
        opt = self.opt
        # the set of tensors this graph updates was frozen at capture; each runs its own schedule step (xml/optimization.py:325-330)
        steps = np.asarray(opt.seg_steps)
        mults = np.zeros(len(steps), np.float32)
        for st in np.unique(steps[self._active_np]):      # (one value once every tensor has stepped equally often)
            mults[self._active_np & (steps == st)] = opt.lr_multiplier(int(st))
        self._set_ranks(neg_ctx_rank if neg_ctx_rank is not None else (batch or {}).get("neg_ctx_rank"),
                        neg_q_rank if neg_q_rank is not None else (batch or {}).get("neg_q_rank"), mults)
        self.graph.replay()
        opt._commit_step(self.active)
        return self.loss, self.parts


class GradientReducer(object):
    """Data-parallel gradient averaging OVERLAPPED with backward (the reference's loop,
    xml/train.py:81-85, has no counterpart -- its nn.DataParallel wrapper is dead code).

    The flat gradient buffer is cut into buckets of ~bucket_bytes along tensor boundaries.  BertAdam's per-parameter
    post-accumulate hooks report every gradient the moment autograd has written it; when the last tensor of a bucket
    has reported, the bucket's all-reduce(AVG) is issued on a side stream (ordered behind the backward kernels enqueued
    so far by an event) while backward keeps running on the main stream.  finish() -- called by allreduce_gradients()
    after loss.backward() -- reduces the buckets that never became complete (tensors without a gradient this step) and
    makes the main stream wait for the side stream.  Every rank issues the same buckets in the same order: a bucket is
    flushed the moment it is complete, and every rank runs the same autograd graph, so completion order is the same on all
    of them; the leftovers go in descending order.
    GPU ranks reduce through libxmlhip's xml_rccl_allreduce_avg_f32 on an ncclComm_t of their own; other devices (the
    gloo tests) through torch.distributed."""

    def __init__(self, optimizer, group=None, bucket_bytes=24 << 20):
        import torch.distributed as dist
        self.opt, self.group = optimizer, group
        self.world = dist.get_world_size(group)
        offs = optimizer.seg_off.tolist()
        self.buckets = []                    # (lo, hi, first_seg, n_segs)
        lo_seg = 0
        for s in range(len(offs) - 1):
            if (offs[s + 1] - offs[lo_seg]) * 4 >= bucket_bytes or s == len(offs) - 2:
                self.buckets.append((offs[lo_seg], offs[s + 1], lo_seg, s + 1 - lo_seg))
                lo_seg = s + 1
        self.seg_bucket = [b for b, (_, _, _, n) in enumerate(self.buckets) for _ in range(n)]
        flat = optimizer.flat_g
        self.cuda = flat.is_cuda
        self.comm = None
        if self.cuda and dist.get_backend(group) == "nccl":
            from .rccl import RcclComm
            self.comm = RcclComm(group)
        self.side = torch.cuda.Stream(flat.device) if self.cuda else None
        self.works = []
        optimizer._reducer = self
        self.begin()

    def begin(self):
        self.count = [0] * len(self.buckets)
        self.reduced = [False] * len(self.buckets)
        self.works = []
        # the stream the step is issued on (zero_grad / the captured body call begin() from it).  A bucket's last hook may
        # fire with a BRANCH stream current (PARALLEL_BRANCHES: AccumulateGrad of a subtitle-branch parameter runs on that
        # branch's stream), so "the current stream" inside the hook is not enough to order the bucket behind the gradient
        # kernels the main stream still has in flight
        self.main = torch.cuda.current_stream(self.opt.flat_g.device) if self.cuda else None

    def _reduce(self, b):
        import torch.distributed as dist
        self.reduced[b] = True
        lo, hi = self.buckets[b][:2]
        view = self.opt.flat_g[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(view.device))
            self.side.wait_event(ev)
            # (eager steps only.  Inside a stream capture the same extra edge -- an event recorded on the capturing origin
            # stream from a hook that runs on a branch stream -- produced a graph whose replays lose gradient updates
            # (tests/rccl_world1_check.py: parameters 1.4e-2 off after three steps, ROCm 7.2); GraphedTrainStep therefore
            # captures a data-parallel step on ONE stream, where the event on the current stream above orders everything.)
            if self.main is not None and not torch.cuda.is_current_stream_capturing():
                self.side.wait_stream(self.main)
            # the branch streams of the training graph (PARALLEL_BRANCHES) write gradients too: a node that accumulated
            # straight into the flat buffer (gradient sink) hands no tensor to autograd, so nothing else orders its kernel
            # before this bucket's all-reduce
            for (dev_index, _), st in list(_SIDE_STREAMS.items()):
                if dev_index == view.device.index:
                    self.side.wait_stream(st)
            with torch.cuda.stream(self.side):
                if self.comm is not None:
                    self.comm.allreduce_avg_(view)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
        else:
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))

    def grad_ready(self, seg):
        b = self.seg_bucket[seg]
        if self.reduced[b]:
            # a second backward() before step() (gradient accumulation) would add LOCAL gradients on top of the already
            # averaged bucket, and finish() would have nothing left to reduce: the ranks would diverge silently
            raise RuntimeError("GradientReducer: gradient for a bucket that was already all-reduced in this step -- run "
                               "one backward per optimizer.zero_grad() (for gradient accumulation, detach the reducer and "
                               "call allreduce_gradients() once after the last backward)")
        self.count[b] += 1
        if self.count[b] >= self.buckets[b][3]:
            # flushed in COMPLETION order.  Every rank runs the same autograd graph, so the order is the same everywhere;
            # (a fixed descending order made every bucket wait for the last one, which holds all biases and LayerNorm
            # parameters -- the reference's [decay, no_decay] groups -- and completes only when backward ends)
            self._reduce(b)

    def finish(self):
        for b in range(len(self.buckets) - 1, -1, -1):      # what never became complete (tensors without a gradient this step)
            if not self.reduced[b]:
                self._reduce(b)
        if self.cuda:
            torch.cuda.current_stream(self.opt.flat_g.device).wait_stream(self.side)
        for w, view in self.works:
            w.wait()
            view.div_(self.world)
        self.works = []


def allreduce_gradients(optimizer, group=None, bucket_bytes=64 << 20):
    """Average the flat gradient buffer over ranks.  With a GradientReducer attached to the optimizer most buckets were
    already reduced under backward and this only flushes the rest and joins the side stream.  Without one: a few large
    all-reduces (64 MiB buckets by default -- ring collectives over xGMI are per-link bound, so fewer and larger beats
    per-tensor), issued back-to-back after backward and awaited together.  No-op without an initialised process group or
    at world size 1."""
    import torch.distributed as dist
    from . import dist as xdist
    if not (dist.is_available() and dist.is_initialized()):
        return
    if getattr(optimizer, "_reducer", None) is not None:
        optimizer._reducer.finish()
        return
    if dist.get_world_size(group) == 1 and xdist.SKIP_TRIVIAL_COLLECTIVES:
        return
    world = dist.get_world_size(group)
    flat = optimizer.flat_g
    step = max(bucket_bytes // 4, 1)
    # RCCL reduces with AVG directly; gloo (CPU tests) has no AVG, so sum and scale there
    avg = flat.is_cuda
    works = []
    for o in range(0, flat.numel(), step):
        works.append(dist.all_reduce(flat[o:o + step], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group,
                                     async_op=True))
    for w in works:
        w.wait()
    if not avg:
        flat.div_(world)


def train_step(model, optimizer, batch, grad_clip=-1, group=None):
    """One iteration of the reference's loop (xml/train.py:78-95): forward, zero_grad, backward, [global clip],
    all-reduce (multi-process only), optimizer step.  `batch` = dict of the XML.forward keyword arguments."""
    loss, loss_dict = xml_forward_train(model, **batch)
    optimizer.zero_grad()
    loss.backward()
    allreduce_gradients(optimizer, group)
    if grad_clip != -1:                                   # nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        T.clip_grad_norm(optimizer.flat_g, grad_clip)
    optimizer.step()
    return loss, loss_dict
