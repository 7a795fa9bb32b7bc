"""GPU parity tests of the training step: every hand-written backward kernel against torch
autograd on the same op (fp32, tolerances in the tests), BertAdam against the oracle restatement, and the whole
step (loss, every parameter gradient, parameters after three optimizer steps) against the golden vectors captured
from the reference (tests/golden/train_step_*.npz)."""
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import xml_oracle as O
from test_gpu_kernels import DEV

pytestmark = pytest.mark.gpu

F32 = torch.float32


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def lens_mask(n, l, seed=0, lo=1):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, l + 1, (n,), generator=g)
    lens[0] = l
    return (torch.arange(l)[None, :] < lens[:, None]).float().to(DEV)


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-12))


def check(name, got, want, tol):
    e = rel_err(got, want)
    assert e <= tol, "%s: max err / max|ref| = %.3e > %.1e" % (name, e, tol)


def run_pair(fn_hip, fn_ref, inputs, needs_grad, tol=2e-5, gout_seed=99):
    """Run both with fresh leaves, the same upstream gradient; compare outputs and all input gradients."""
    outs = []
    for fn, dt in ((fn_hip, F32), (fn_ref, torch.float64)):      # the torch reference runs in float64
        leaves = [t.clone().to(dt).requires_grad_(ng) if t is not None and t.is_floating_point() else t
                  for t, ng in zip(inputs, needs_grad)]
        y = fn(*leaves)
        g = rnd(*y.shape, seed=gout_seed) if y.dim() else torch.ones((), device=DEV)
        y.backward(g.to(y.dtype))
        outs.append((y, [l.grad if (l is not None and ng) else None for l, ng in zip(leaves, needs_grad)]))
    (y0, g0), (y1, g1) = outs
    check("forward", y0, y1, tol)
    for i, (a, b) in enumerate(zip(g0, g1)):
        if b is not None:
            assert a is not None, "missing gradient %d" % i
            check("grad[%d]" % i, a, b, tol)


@pytest.mark.parametrize("rows,k,n,relu", [(37, 64, 48, False), (144, 96, 128, True), (300, 3072, 128, True),
                                            (6, 128, 2, False)])
def test_linear_fn(rows, k, n, relu):
    from tvretrieval_amd.autograd import LinearFn
    x, w, b = rnd(rows, k, seed=1), rnd(n, k, seed=2, scale=0.1), rnd(n, seed=3, scale=0.1)
    ref = lambda x, w, b: F.relu(F.linear(x, w, b)) if relu else F.linear(x, w, b)    # noqa: E731
    run_pair(lambda x, w, b: LinearFn.apply(x, w, b, relu), ref, [x, w, b], [True, True, True], tol=5e-5)


@pytest.mark.parametrize("rows,d,resid", [(50, 128, True), (33, 768, False), (20, 3072, False), (9, 200, True)])
def test_layernorm_fn(rows, d, resid):
    from tvretrieval_amd.autograd import LayerNormFn
    a, b = rnd(rows, d, seed=1), (rnd(rows, d, seed=2) if resid else None)
    g, beta = 1 + rnd(d, seed=3, scale=0.2), rnd(d, seed=4, scale=0.2)

    def ref(a, b, g, beta):
        return F.layer_norm(a + b if b is not None else a, (d,), g, beta, 1e-5)
    needs_a = d <= 1024
    run_pair(lambda a, b, g, beta: LayerNormFn.apply(a, b, g, beta, F32), ref, [a, b, g, beta],
             [needs_a, resid, True, True], tol=5e-5)


def test_narrow_layernorm_backward_parameters_only():
    """d <= 1024 with dx = NULL (the 768-d subtitle / query input LayerNorm: raw features need no gradient): dgamma / dbeta
    equal those of the call that also writes dx; with and without the output dropout site."""
    from tvretrieval_amd import train_ops as TO
    rows, d = 3000, 768
    a = rnd(rows, d, seed=1) * 2 + 0.3
    dy = rnd(rows, d, seed=2).to(torch.bfloat16)
    g = 1 + rnd(d, seed=3, scale=0.2)
    dx0, dg0, db0 = TO.layernorm_bwd(a, None, g, dy, need_dx=False)
    dx1, dg1, db1 = TO.layernorm_bwd(a, None, g, dy, need_dx=True)
    assert dx0 is None and dx1 is not None
    check("dgamma", dg0, dg1, 2e-6)
    check("dbeta", db0, db1, 2e-6)
    r0 = TO.layernorm_bwd_drop(a, None, g, dy, 0.0, 0, 0.1, 99, need_dx=False)
    r1 = TO.layernorm_bwd_drop(a, None, g, dy, 0.0, 0, 0.1, 99, need_dx=True)
    assert r0[0] is None and r0[1] is None and r1[0] is not None
    check("dgamma (dropout site)", r0[2], r1[2], 2e-6)
    check("dbeta (dropout site)", r0[3], r1[3], 2e-6)


@pytest.mark.parametrize("rows,d,a_dt", [(5000, 3072, F32), (37, 3072, torch.bfloat16), (301, 2056, F32), (64, 4096, F32),
                                         (12800, 3072, F32), (1030, 1032, torch.bfloat16)])
def test_wide_layernorm_backward_parameters_only(rows, d, a_dt):
    """the input LayerNorm of the 3072-d video features needs dgamma / dbeta only: xml_layernorm_bwd's one-pass kernel
    (dx = NULL, bf16 dy) == float64 reference on the same values, and == the two-kernel path that also writes dx."""
    from tvretrieval_amd import train_ops as TO
    a = (rnd(rows, d, seed=1) * 2 + 0.3).to(a_dt)
    dy = rnd(rows, d, seed=2).to(torch.bfloat16)
    g = 1 + rnd(d, seed=3, scale=0.2)
    dx0, dg, db = TO.layernorm_bwd(a, None, g, dy, need_dx=False)
    assert dx0 is None
    x = a.double()
    xh = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    check("dgamma", dg, (dy.double() * xh).sum(0), 2e-5)
    check("dbeta", db, dy.double().sum(0), 2e-5)
    dx1, dg1, db1 = TO.layernorm_bwd(a, None, g, dy, need_dx=True)
    assert dx1 is not None
    check("dgamma vs two-kernel path", dg, dg1, 2e-5)
    check("dbeta vs two-kernel path", db, db1, 2e-5)
    # gradient sinks: the sums are ADDED to what dg / dbeta hold (from 1024 rows on through per-workgroup partials in the
    # scratch + a combining launch, below that with atomics)
    sink_g, sink_b = torch.full((d,), 2.0, device=DEV), torch.full((d,), -3.0, device=DEV)
    TO.layernorm_bwd(a, None, g, dy, need_dx=False, dg=sink_g, dbeta=sink_b)
    check("dgamma sink", sink_g, (dy.double() * xh).sum(0) + 2.0, 2e-5)
    check("dbeta sink", sink_b, dy.double().sum(0) - 3.0, 2e-5)


def ref_attention(q, k, v, qm, km, heads):
    n, lq, hsz = q.shape
    lk = k.shape[1]
    dh = hsz // heads
    mask = km[:, None, :] if qm is None else qm[:, :, None] * km[:, None, :]
    add = (1 - mask.unsqueeze(1)) * -10000.0
    sp = lambda t, l: t.view(n, l, heads, dh).permute(0, 2, 1, 3)     # noqa: E731
    s = torch.matmul(sp(q, lq), sp(k, lk).transpose(-1, -2)) / math.sqrt(dh) + add
    o = torch.matmul(torch.softmax(s, -1), sp(v, lk))
    return o.permute(0, 2, 1, 3).reshape(n, lq, hsz)


@pytest.mark.parametrize("n,lq,lk,hsz,heads,cross", [(3, 24, 24, 128, 4, False), (2, 13, 21, 128, 4, True),
                                                     (2, 100, 100, 768, 4, False), (2, 128, 128, 256, 4, True)])
def test_attention_core_fn(n, lq, lk, hsz, heads, cross):
    from tvretrieval_amd.autograd import AttentionCoreFn
    q, k, v = rnd(n, lq, hsz, seed=1), rnd(n, lk, hsz, seed=2), rnd(n, lk, hsz, seed=3)
    km = lens_mask(n, lk, seed=4, lo=3)
    qm = lens_mask(n, lq, seed=5, lo=3) if cross else None
    # fully masked query rows compute softmax(s/sqrt(d) - 1e4): fp32 keeps ~1e-3 of s there (in the reference too),
    # so those rows are compared separately at that resolution and excluded from the tight comparison
    keep = 1.0 if qm is None else qm[:, :, None]
    run_pair(lambda q, k, v: AttentionCoreFn.apply(q, k, v, qm, km, heads) * keep,
             lambda q, k, v: ref_attention(q, k, v, qm, km, heads) * keep, [q, k, v], [True, True, True], tol=5e-5)
    if qm is not None:
        with torch.no_grad():
            check("masked rows", AttentionCoreFn.apply(q, k, v, qm, km, heads),
                  ref_attention(q.double(), k.double(), v.double(), qm, km, heads), 2e-3)


@pytest.mark.parametrize("n_mod", [1, 2])
@pytest.mark.parametrize("n,l,h", [(5, 11, 128), (9, 30, 768), (3, 128, 200), (2, 17, 2048), (4, 13, 100)])
def test_modular_pool_fn(n_mod, n, l, h):
    """h % 8 == 0 (<= 2048): the 16-byte backward kernel (loss_tail.hip); h = 100: the scalar one."""
    from tvretrieval_amd.autograd import ModularPoolFn
    enc, wm, mask = rnd(n, l, h, seed=1), rnd(n_mod, h, seed=2, scale=0.3), lens_mask(n, l, seed=3, lo=2)

    def ref(enc, wm):
        sc = F.linear(enc, wm)
        sc = torch.softmax(O.mask_logits(sc, mask.unsqueeze(2)), dim=1)
        return torch.einsum("blm,bld->mbd", sc, enc)
    run_pair(lambda enc, wm: ModularPoolFn.apply(enc, mask, wm), ref, [enc, wm], [True, True], tol=5e-5)


@pytest.mark.parametrize("n_mod,l", [(1, 24), (2, 19), (2, 32)])
def test_video_level_scores_fn(n_mod, l):
    from tvretrieval_amd.autograd import VideoLevelScoresFn
    n, h = 7, 128
    qs = [rnd(n, h, seed=10 + i) for i in range(n_mod)]
    fs = [rnd(n, l, h, seed=20 + i) for i in range(n_mod)]
    ms = [lens_mask(n, l, seed=30, lo=2) for _ in range(n_mod)]

    def ref(*t):
        tot = 0
        for i in range(n_mod):
            q, c = F.normalize(t[i], dim=-1), F.normalize(t[n_mod + i], dim=-1)
            s = torch.einsum("md,nld->mln", q, c)
            s = O.mask_logits(s, ms[i].transpose(0, 1).unsqueeze(0))
            tot = tot + torch.max(s, dim=1)[0]
        return tot / n_mod
    run_pair(lambda *t: VideoLevelScoresFn.apply(n_mod, *t, *ms), ref, qs + fs, [True] * (2 * n_mod), tol=5e-5)


@pytest.mark.parametrize("nq,nv,l,h,dt,dense", [(128, 128, 100, 768, torch.bfloat16, False), (40, 24, 19, 128, F32, False),
                                                 (16, 9, 32, 256, F32, True), (7, 300, 5, 64, torch.bfloat16, True)])
def test_video_level_scores_backward_one_launch_vs_separate_kernels(nq, nv, l, h, dt, dense):
    """xml_q2c_scores_l2norm_bwd == xml_q2c_scores_bwd (f32 atomics) -> slice -> xml_l2norm_bwd on the same saved tensors:
    a ranking-loss-like sparse dscores (<= 3 pairs per row / column) and a dense one; clips padded to a multiple of 16;
    masked clips; an all-zero clip row (F.normalize clamps its norm)."""
    from tvretrieval_amd import ops, train_ops as TO
    lpad = (l + 15) // 16 * 16
    query, feat = rnd(nq, h, seed=1).to(dt), rnd(nv, l, h, seed=2).to(dt)
    feat[1, 0] = 0
    mask = lens_mask(nv, l, seed=3, lo=2)
    qn, cn = ops.l2norm_rows(query), ops.l2norm_rows(feat)
    cn_p = torch.zeros(nv, lpad, h, dtype=dt, device=DEV)
    mk_p = torch.zeros(nv, lpad, device=DEV)
    cn_p[:, :l], mk_p[:, :l] = cn, mask
    g = torch.Generator().manual_seed(7)
    if dense:
        ds = torch.randn(nq, nv, generator=g)
    else:
        ds = torch.zeros(nq, nv)
        for m in range(nq):
            for n in torch.randint(0, nv, (3,), generator=g).tolist():
                ds[m, n] += float(torch.randn((), generator=g))
    ds = ds.to(DEV)
    assert TO.q2c_scores_l2norm_bwd_supported(nq, nv, l, h, dt)
    dq, df = TO.q2c_scores_l2norm_bwd(query, feat, qn, cn_p, mk_p, ds, scale=0.5)
    # ... and with the arg-max clips kept by the forward kernel (unpadded operands): the same rows, the same values
    _, arg = TO.q2c_scores_arg(qn, cn, mask.contiguous())
    dq2, df2 = TO.q2c_scores_l2norm_bwd(query, feat, qn, cn, mask.contiguous(), ds, scale=0.5, arg=arg)
    check("dquery (arg kept)", dq2, dq, 1e-6)
    check("dfeat (arg kept)", df2, df, 1e-6)
    # ... and both modalities in one launch (the second one: other values, no kept arg-max, padded clips)
    query_b, feat_b = rnd(nq, h, seed=11).to(dt), rnd(nv, l, h, seed=12).to(dt)
    qn_b, cn_b = ops.l2norm_rows(query_b), ops.l2norm_rows(feat_b)
    cn_bp = torch.zeros(nv, lpad, h, dtype=dt, device=DEV)
    cn_bp[:, :l] = cn_b
    want_b = TO.q2c_scores_l2norm_bwd(query_b, feat_b, qn_b, cn_bp, mk_p, ds, scale=0.5)
    got = TO.q2c_scores_l2norm_bwd_multi([(query, feat, qn, cn, mask.contiguous(), arg), (query_b, feat_b, qn_b, cn_bp, mk_p, None)],
                                         ds, scale=0.5)
    assert torch.equal(got[0][0], dq2) and torch.equal(got[0][1], df2)
    assert torch.equal(got[1][0], want_b[0]) and torch.equal(got[1][1], want_b[1])
    dqn, dcn = TO.q2c_scores_bwd(qn, cn_p, mk_p, ds, scale=0.5)
    want_q, want_f = TO.l2norm_bwd(query, dqn), TO.l2norm_bwd(feat, dcn[:, :l].contiguous())
    tol = 2e-5 if dt == F32 else 8e-3           # bf16: one rounding of the outputs, different f32 summation orders
    check("dquery", dq, want_q, tol)
    check("dfeat", df, want_f, tol)
    assert torch.equal(df == 0, want_f == 0) or dt != F32      # the same rows are touched
    assert float(df.float().abs().sum()) > 0


@pytest.mark.parametrize("nq,nv,l,h,dt", [(128, 128, 100, 768, torch.bfloat16), (40, 24, 19, 128, F32), (7, 300, 5, 64, torch.bfloat16),
                                           (130, 9, 128, 256, F32), (1, 1, 1, 8, F32)])
def test_q2c_scores_arg_forward(nq, nv, l, h, dt):
    """xml_q2c_scores_arg (unpadded clips, arg-max kept) == xml_q2c_scores on the zero-padded operands, incl. combine; the
    arg-max is the float64 arg-max wherever the top two clips are further apart than the f32 summation noise; a video whose
    clips are all masked out reports clip 0."""
    from tvretrieval_amd import ops, train_ops as TO
    lpad = (l + 15) // 16 * 16
    qn = ops.l2norm_rows(rnd(nq, h, seed=1).to(dt))
    cn = ops.l2norm_rows(rnd(nv, l, h, seed=2).to(dt))
    mask = lens_mask(nv, l, seed=3, lo=1)
    if nv > 2:
        mask[2] = 0
    cn_p = torch.zeros(nv, lpad, h, dtype=dt, device=DEV)
    mk_p = torch.zeros(nv, lpad, device=DEV)
    cn_p[:, :l], mk_p[:, :l] = cn, mask
    want = ops.q2c_scores(qn, cn_p, mk_p)
    got, arg = TO.q2c_scores_arg(qn, cn, mask.contiguous())
    assert float((got - want).abs().max()) <= 2e-6
    s = torch.einsum("md,nld->mnl", qn.double(), cn.double())
    s = s * mask.double()[None] + (1 - mask.double()[None]) * -1e10
    top2 = torch.topk(s, min(2, l), dim=-1)
    clear = (top2.values[..., 0] - top2.values[..., -1] > 1e-5) if l > 1 else torch.ones_like(s[..., 0], dtype=torch.bool)
    assert torch.equal(arg.long()[clear], top2.indices[..., 0][clear])
    assert float(clear.float().mean()) > 0.85 or nv <= 2
    assert int(arg.min()) >= 0 and int(arg.max()) < l
    if nv > 2:
        assert int(arg[:, 2].abs().max()) == 0 and float(got[:, 2].max()) == -1e10
    # combine: (previous + this) / 2, as xml_q2c_scores
    prev = rnd(nq, nv, seed=9)
    a, b = prev.clone(), prev.clone()
    ops.q2c_scores(qn, cn_p, mk_p, out=a, combine=True)
    TO.q2c_scores_arg(qn, cn, mask.contiguous(), out=b, combine=True)
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())) or nv > 2
    if nv > 2:
        keep = torch.ones(nv, dtype=torch.bool, device=DEV)
        keep[2] = False
        assert float((a - b)[:, keep].abs().max()) <= 2e-6


def test_loss_combine_fn():
    from tvretrieval_amd.autograd import CombineLossFn
    a = torch.tensor(0.73, device=DEV, requires_grad=True)
    r = torch.tensor([1.25, -0.5], device=DEV, requires_grad=True)
    loss, parts = CombineLossFn.apply(a, r, 0.01, 1.0, 2.0)
    assert not parts.requires_grad
    (loss * 3.0).backward()
    want = [0.01 * 0.73, 1.25, -1.0]
    assert torch.allclose(parts.cpu(), torch.tensor(want + [sum(want)]), rtol=1e-6)
    assert abs(float(loss) - sum(want)) < 1e-6
    assert abs(float(a.grad) - 0.03) < 1e-7 and torch.allclose(r.grad.cpu(), torch.tensor([3.0, 6.0]))
    loss2, parts2 = CombineLossFn.apply(None, r, 0.0, 1.0, 1.0)
    assert abs(float(loss2) - 0.75) < 1e-6 and float(parts2[0]) == 0.0


def test_train_step_fused_loss_tail_equals_separate_launches():
    """autograd.FUSED_LOSS_TAIL on / off on the golden batch (fp32, eval): same loss, same gradients."""
    import tvretrieval_amd.autograd as AG
    import tvretrieval_amd.train as TR
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    m = build_train_model(cfg, d)
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]), neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    res = []
    try:
        for fused in (True, False):
            AG.FUSED_LOSS_TAIL = fused
            m.zero_grad()
            loss, parts = TR.xml_forward_train(m, **batch)
            loss.backward()
            res.append((float(loss), parts, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        AG.FUSED_LOSS_TAIL = True
    assert abs(res[0][0] - res[1][0]) < 1e-6 and abs(res[0][0] - float(d["loss"])) < 5e-5
    for k in res[1][1]:
        assert abs(res[0][1][k] - res[1][1][k]) < 1e-6, k
    for n, g0 in res[0][2].items():
        if not n.endswith(".key.bias"):
            check(n, g0, res[1][2][n], 5e-5)


def test_pair_sim_fn():
    from tvretrieval_amd.autograd import PairSimFn
    q, f2 = rnd(6, 128, seed=1), rnd(6, 23, 128, seed=2)
    run_pair(lambda q, f2: PairSimFn.apply(q, f2), lambda q, f2: torch.einsum("bd,bld->bl", q, f2), [q, f2],
             [True, True], tol=2e-5)


@pytest.mark.parametrize("merged,n_sim", [(True, 2), (False, 2), (False, 1)])
def test_span_loss_fn(merged, n_sim):
    from tvretrieval_amd.autograd import SpanLossFn
    n, l, ks = 9, 37, 5
    sims = [rnd(n, l, seed=1 + i, scale=3.0) for i in range(n_sim)]
    mask = lens_mask(n, l, seed=5, lo=4)
    masks = [mask] * n_sim
    n_filt = 1 if merged else n_sim
    filters = [rnd(1, 1, ks, seed=10 + i, scale=0.5) for i in range(2 * n_filt)]
    lens = mask.sum(1).long().cpu()
    g = torch.Generator().manual_seed(7)
    st = torch.stack([torch.randint(0, int(x), (1,), generator=g)[0] for x in lens])
    ed = torch.stack([torch.randint(int(s), int(x), (1,), generator=g)[0] for s, x in zip(st, lens)])
    st_ed = torch.stack([st, ed], 1).to(DEV)

    def ref(*t):
        sims_, filt = t[:n_sim], t[n_sim:]
        conv = lambda s, w: F.conv1d(s.unsqueeze(1), w, padding=ks // 2).squeeze(1)     # noqa: E731
        if merged:
            s = (sims_[0] + sims_[1]) / 2
            lst, led = O.mask_logits(conv(s, filt[0]), mask), O.mask_logits(conv(s, filt[1]), mask)
        else:
            lst = sum(O.mask_logits(conv(sims_[i], filt[i]), mask) for i in range(n_sim)) / n_sim
            led = sum(O.mask_logits(conv(sims_[i], filt[n_sim + i]), mask) for i in range(n_sim)) / n_sim
        return F.cross_entropy(lst, st_ed[:, 0]) + F.cross_entropy(led, st_ed[:, 1])
    run_pair(lambda *t: SpanLossFn.apply(merged, ks, st_ed, n_sim, *t[:n_sim], *masks, *t[n_sim:]), ref,
             sims + filters, [True] * (n_sim + len(filters)), tol=5e-5)


@pytest.mark.parametrize("lse", [False, True])
def test_rank_loss_fn(lse):
    from tvretrieval_amd.autograd import RankLossFn
    n = 17
    scores = rnd(n, n, seed=1, scale=0.3)
    g = torch.Generator().manual_seed(3)
    rc, rq = torch.randint(1, n, (n,), generator=g), torch.randint(1, n, (n,), generator=g)

    def ref(scores):
        ar = torch.arange(n, device=DEV)
        pos = scores[ar, ar]
        masked = scores.detach().clone()
        masked[ar, ar] = 999

        def neg(sc, scm, r):
            order = torch.sort(scm, descending=True, dim=1)[1]
            return sc[ar, order[ar, r.to(DEV)]]

        def rl(p, ng):
            return torch.log1p(torch.exp(ng - p)).sum() / n if lse else torch.clamp(0.1 + ng - p, min=0).sum() / n
        return torch.stack([rl(pos, neg(scores, masked, rc)), rl(pos, neg(scores.t(), masked.t(), rq))])
    run_pair(lambda s: RankLossFn.apply(s, rc.to(DEV).int(), rq.to(DEV).int(), 0.1, lse), ref, [scores], [True],
             tol=2e-5)


def test_bert_adam_kernel_vs_oracle():
    from tvretrieval_amd.train import BertAdam
    shapes = [(128, 96), (128,), (1, 1, 5), (40, 128), (3,), (257, 33), (300, 128), (4096,), (2, 4096), (1,)]
    ps = [torch.nn.Parameter(rnd(*s, seed=i, scale=0.05)) for i, s in enumerate(shapes)]
    names = ["w%d" % i for i in range(len(ps))]
    wd = {n: (0.01 if p.dim() > 1 else 0.0) for n, p in zip(names, ps)}
    ref_p = {n: p.detach().cpu().clone() for n, p in zip(names, ps)}
    okw = dict(lr=3e-4, warmup=0.2, t_total=10, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0)
    opt = BertAdam([{"params": [p for p in ps if p.dim() > 1], "weight_decay": 0.01},
                    {"params": [p for p in ps if p.dim() <= 1], "weight_decay": 0.0}], schedule="warmup_linear", **okw)
    state = {}
    for it in range(5):
        opt.zero_grad()
        grads = {}
        for i, (n, p) in enumerate(zip(names, ps)):
            g = rnd(*p.shape, seed=100 * it + i, scale=(3.0 if i % 2 else 0.01))    # clipped and unclipped tensors
            p.grad.copy_(g)
            grads[n] = g.cpu().clone()
        opt.step()
        O.bert_adam_step(ref_p, grads, state, it, wd, **okw)
        for n, p in zip(names, ps):
            check("step %d %s" % (it, n), p, ref_p[n], 2e-6)
            check("clipped grad %d %s" % (it, n), p.grad, grads[n], 2e-6)


def build_train_model(cfg, d, dtype=F32):
    from tvretrieval_amd.model_xml import XML
    m = XML(cfg, compute_dtype=dtype)
    sd = {k[len("sd_before/"):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("sd_before/")}
    m.load_state_dict(sd)
    return m.to(DEV).eval()     # the fixtures were captured from the reference in eval mode (dropout off)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight"]


@pytest.mark.parametrize("name", ["train_step_video_sub_h128", "train_step_nocross_lse_h128"])
def test_golden_train_steps_fp32(name):
    """Three iterations of the reference's loop on the fixture batch: step-1 loss terms and every parameter
    gradient, per-step losses, and the parameters after the third BertAdam step."""
    from tvretrieval_amd.train import BertAdam, xml_forward_train
    d, cfg, _ = load_golden(name)
    m = build_train_model(cfg, d)
    okw = json.loads(str(d["optim"]))
    named = list(m.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    opt = BertAdam(groups, **okw)
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    for it in range(3):
        loss, parts = xml_forward_train(m, neg_ctx_rank=d["neg_ctx_rank_steps"][it],
                                        neg_q_rank=d["neg_q_rank_steps"][it], **batch)
        assert abs(float(loss) - float(d["step_losses"][it])) < 5e-5, (it, float(loss), float(d["step_losses"][it]))
        opt.zero_grad()
        loss.backward()
        if it == 0:
            for k in ("loss_st_ed", "loss_neg_ctx", "loss_neg_q"):
                assert abs(parts[k] - float(d[k])) < 2e-5, (k, parts[k], float(d[k]))
            worst = []
            for n, p in named:
                if ("grad/" + n) in d:
                    want = torch.from_numpy(d["grad/" + n])
                    # key biases have an analytically ZERO gradient (softmax is shift invariant): both sides hold
                    # rounding noise there, hence the absolute floor
                    err = float((p.grad.cpu() - want).abs().max())
                    worst.append((err / max(float(want.abs().max()), 1e-4), n, err))
            worst.sort(reverse=True)
            print("worst gradient errors:", worst[:4])
            assert worst[0][0] < 5e-4, "gradient mismatch: %s" % worst[:5]
        opt.step()
    sd = m.state_dict()
    errs = sorted(((rel_err(sd[k[len("sd_after3/"):]], torch.from_numpy(v)), k) for k, v in d.items()
                   if k.startswith("sd_after3/")), reverse=True)
    print("worst parameter errors after 3 steps:", errs[:3])
    assert errs[0][0] < 2e-4, "parameters after 3 steps: %s" % errs[:5]
    # the moments moved the weights: make sure the comparison is not vacuous
    moved = max(float((sd[k[len("sd_after3/"):]].cpu() - torch.from_numpy(d["sd_before/" + k[len("sd_after3/"):]])).abs().max())
                for k in d if k.startswith("sd_after3/"))
    assert moved > 1e-5


@pytest.mark.parametrize("name", ["train_step_video_sub_h128", "train_step_nocross_lse_h128"])
def test_graphed_train_step_reproduces_the_reference_steps(name):
    """train.GraphedTrainStep (the whole iteration as ONE HIP graph) on the golden fixture: the replayed steps give the
    REFERENCE's per-step losses and its parameters after the third BertAdam step -- the schedule multiplier (0 at step 0,
    then 1.0, 0.889), the injected negatives and the per-tensor clip all reach the captured kernels -- and constructing
    the object (warm-up + capture) does not train."""
    from tvretrieval_amd.train import BertAdam, GraphedTrainStep
    d, cfg, _ = load_golden(name)
    m = build_train_model(cfg, d)
    okw = json.loads(str(d["optim"]))
    named = list(m.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    opt = BertAdam(groups, **okw)
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    before = opt.flat_p.clone()
    step = GraphedTrainStep(m, opt, batch)
    assert torch.equal(opt.flat_p, before) and opt.step_count == 0 and float(opt.flat_m.abs().max()) == 0.0
    for it in range(3):
        loss, parts = step(batch, neg_ctx_rank=d["neg_ctx_rank_steps"][it], neg_q_rank=d["neg_q_rank_steps"][it])
        assert abs(float(loss) - float(d["step_losses"][it])) < 5e-5, (it, float(loss), float(d["step_losses"][it]))
        assert abs(float(parts["loss_overall"]) - float(loss)) < 1e-6
    assert opt.step_count == 3
    worst = 0.0
    for n, p in named:
        want = torch.from_numpy(d["sd_after3/" + n])
        worst = max(worst, float((p.detach().cpu() - want).abs().max()))
    assert worst < 2e-5, worst


def test_graphed_train_step_bf16_dropout_draws_fresh_masks_and_descends():
    """bf16, model.train(): every replay advances the device-resident base seed (different masks => different losses on the
    SAME batch with the learning rate at zero), and with a learning rate the loss goes down like the eager loop's."""
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.train import BertAdam, GraphedTrainStep
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    ranks = dict(neg_ctx_rank=d["neg_ctx_rank_steps"][0], neg_q_rank=d["neg_q_rank_steps"][0])
    torch.manual_seed(3)
    m = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).train()
    opt = BertAdam(m.parameters(), lr=0.0, warmup=-1, t_total=-1, schedule="none")
    step = GraphedTrainStep(m, opt, batch)
    losses = [float(step(None, **ranks)[0]) for _ in range(4)]
    assert len({round(x, 6) for x in losses}) == 4, losses            # same weights, same batch: only the masks differ
    m.eval()                                                          # (the graph was captured in train mode: unchanged)
    torch.manual_seed(3)
    m2 = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).train()
    opt2 = BertAdam(m2.parameters(), lr=2e-3, warmup=-1, t_total=-1, schedule="none")
    step2 = GraphedTrainStep(m2, opt2, batch)
    ls = [float(step2(None, **ranks)[0]) for _ in range(12)]
    assert np.mean(ls[-3:]) < np.mean(ls[:3]) - 0.02, ls


def test_golden_staged_training_fp32():
    """train_span_start_epoch: two steps with lw_st_ed = 0, then three with the span loss.  The 42 tensors behind the span
    branch must stay untouched (no weight decay, no moments) until they first receive a gradient and then run their OWN
    warm-up (per-tensor step), xml/optimization.py:289-291,325-330 -- vs the reference's parameters after five steps."""
    from tvretrieval_amd.train import BertAdam, xml_forward_train
    d, cfg, _ = load_golden("train_step_staged_video_sub_h128")
    m = build_train_model(cfg, d)
    okw = json.loads(str(d["optim"]))
    named = list(m.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    opt = BertAdam(groups, **okw)
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    late = set(json.loads(str(d["no_grad_at_step0"])))
    before = {n: p.detach().clone() for n, p in named}
    assert opt.get_lr() == [0]
    for it, lw in enumerate(d["lw_st_ed_schedule"]):
        m.config.lw_st_ed = float(lw)
        loss, _ = xml_forward_train(m, neg_ctx_rank=d["neg_ctx_rank_steps"][it], neg_q_rank=d["neg_q_rank_steps"][it],
                                    **batch)
        assert abs(float(loss) - float(d["step_losses"][it])) < 5e-5, (it, float(loss), float(d["step_losses"][it]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it == 1:      # after two span-less steps the late tensors are bit-for-bit what they were
            for n, p in named:
                if n in late:
                    assert torch.equal(p.detach(), before[n]), n
    want_steps = json.loads(str(d["final_steps"]))
    got_steps = {n: opt.seg_steps[i] for i, (n, _) in enumerate((n, p) for g in groups for p in g["params"]
                                                                  for n, q in named if q is p)}
    assert got_steps == want_steps
    sd = m.state_dict()
    errs = sorted(((rel_err(sd[k[len("sd_after3/"):]], torch.from_numpy(v)), k) for k, v in d.items()
                   if k.startswith("sd_after3/")), reverse=True)
    print("worst parameter errors after the staged steps:", errs[:3])
    assert errs[0][0] < 2e-4, errs[:5]


def test_c5_shape_bf16_gradients_vs_fp32():
    """BASELINE configs[4] at its stated shape -- batch 128, video+sub, L = 100 clips, H = 768, Dv = 3072, Ds = Dq = 768
    (20.2 M parameters) -- one forward / backward in bf16 compute against the fp32 HIP path (which the golden fixtures pin
    to the reference's autograd): loss terms within 2 %, every parameter gradient within bf16 noise of the fp32 gradient
    (cosine >= 0.97, norm ratio within 10 % for tensors that carry a real gradient)."""
    from tvretrieval_amd.model_xml import XML, xml_base_config
    from tvretrieval_amd.train import xml_forward_train
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from rank_agreement import perturb_weights
    bsz, lc, lq, h = 128, 100, 30, 768
    cfg = dict(xml_base_config)
    cfg.update(visual_input_size=3072, sub_input_size=768, query_input_size=768, hidden_size=h, max_ctx_l=lc,
               max_desc_l=lq, lw_st_ed=0.01)
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(lc // 2, lc + 1, (bsz,), generator=g); lens[0] = lc
    qlens = torch.randint(5, lq + 1, (bsz,), generator=g); qlens[0] = lq
    mk = lambda ls, l: (torch.arange(l)[None] < ls[:, None]).float()                 # noqa: E731

    def feats(l, d, m):
        x = torch.nn.functional.normalize(torch.randn(bsz, l, d, generator=g), dim=-1) * m[:, :, None]
        return x.to(DEV)
    vm, qm = mk(lens, lc), mk(qlens, lq)
    st = torch.stack([torch.randint(0, int(x), (1,), generator=g)[0] for x in lens])
    ed = torch.stack([torch.randint(int(s), int(x), (1,), generator=g)[0] for s, x in zip(st, lens)])
    batch = dict(query_feat=feats(lq, 768, qm), query_mask=qm.to(DEV), video_feat=feats(lc, 3072, vm),
                 video_mask=vm.to(DEV), sub_feat=feats(lc, 768, vm), sub_mask=vm.to(DEV),
                 st_ed_indices=torch.stack([st, ed], 1).to(DEV),
                 neg_ctx_rank=torch.randint(1, bsz, (bsz,), generator=g), neg_q_rank=torch.randint(1, bsz, (bsz,), generator=g))
    grads, losses = {}, {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        m = perturb_weights(XML(cfg, compute_dtype=dt)).to(DEV).eval()       # eval: dropout off, like the goldens
        loss, parts = xml_forward_train(m, **batch)
        loss.backward()
        torch.cuda.synchronize()
        grads[name] = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        losses[name] = parts
        del m
    assert set(grads["f32"]) == set(grads["bf16"]) and len(grads["f32"]) > 80
    for k in ("loss_st_ed", "loss_neg_ctx", "loss_neg_q"):
        a, b = losses["f32"][k], losses["bf16"][k]
        assert abs(a - b) <= 2e-2 * max(abs(a), 1e-3), (k, a, b)
    gmax = max(float(v.norm()) for v in grads["f32"].values())
    worst = []
    for n, gf in grads["f32"].items():
        gb = grads["bf16"][n]
        nf, nb = float(gf.norm()), float(gb.norm())
        if nf < 1e-4 * gmax:            # analytically-zero gradients (key biases): rounding noise on both sides
            continue
        cos = float((gf * gb).sum()) / (nf * nb + 1e-30)
        worst.append((cos, nb / nf, n))
    worst.sort()
    print("C5 shape, bf16 vs fp32 gradients: worst cosine %.4f (%s), norm ratio range %.3f..%.3f over %d tensors"
          % (worst[0][0], worst[0][2], min(w[1] for w in worst), max(w[1] for w in worst), len(worst)))
    assert worst[0][0] >= 0.97, worst[:5]
    assert all(0.9 <= w[1] <= 1.1 for w in worst), sorted(worst, key=lambda w: abs(w[1] - 1))[-5:]


def test_train_step_bf16_runs_and_descends():
    """bf16 compute: the loss of the fixture batch must go down over a few steps (no golden for bf16)."""
    from tvretrieval_amd.train import BertAdam, train_step
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    m = build_train_model(cfg, d, torch.bfloat16)
    opt = BertAdam(m.parameters(), lr=5e-4, warmup=-1, t_total=-1, schedule="none")
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]), neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    losses = [float(train_step(m, opt, batch)[0]) for _ in range(12)]
    assert all(math.isfinite(x) for x in losses)
    assert losses[-1] < losses[0] - 0.02, losses


def test_model_forward_dispatch():
    """XML.forward: autograd on -> training graph; no_grad -> fused inference kernels; same loss values."""
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    m = build_train_model(cfg, d)
    args = (T(d["query_feat"]), T(d["query_mask"]), T(d["video_feat"]), T(d["video_mask"]), T(d["sub_feat"]),
            T(d["sub_mask"]), None, None, T(d["st_ed_indices"]))
    loss, parts = m(*args, neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    assert loss.requires_grad
    with torch.no_grad():
        loss2, parts2 = m(*args, neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    assert abs(float(loss) - float(d["loss"])) < 2e-5 and abs(float(loss2) - float(d["loss"])) < 2e-5
    for k in ("loss_st_ed", "loss_neg_ctx", "loss_neg_q"):
        assert abs(parts[k] - parts2[k]) < 2e-5


def test_dropout_kernel_statistics_and_backward():
    from tvretrieval_amd.autograd import DropoutFn
    from tvretrieval_amd import train_ops as TO
    x = torch.ones(1 << 20, device=DEV)
    for p in (0.1, 0.5):
        y = TO.dropout(x, p, seed=1234)
        keep = (y != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 4e-3, (p, keep)                     # ~4 sigma at n = 2^20
        assert torch.allclose(y[y != 0], torch.full((1,), 1 / (1 - p), device=DEV))
        assert torch.equal(y, TO.dropout(x, p, seed=1234))                # pure function of (seed, index)
        y2 = TO.dropout(x, p, seed=1235)
        both = ((y != 0) & (y2 != 0)).float().mean().item()               # independent masks across seeds
        assert abs(both - (1 - p) ** 2) < 6e-3, (p, both)
    # no visible structure along rows of a (rows, 768) activation: per-column keep rates stay binomial
    y = TO.dropout(torch.ones(4096, 768, device=DEV), 0.1, seed=77)
    col = (y != 0).float().mean(0)
    assert float((col - 0.9).abs().max()) < 0.03
    # backward = the same mask and scale
    xx = torch.randn(1000, 64, device=DEV, requires_grad=True)
    out = DropoutFn.apply(xx, 0.3, 99)
    out.backward(torch.ones_like(out))
    assert torch.equal(xx.grad != 0, out != 0)
    assert torch.allclose(xx.grad[xx.grad != 0], torch.full((1,), 1 / 0.7, device=DEV))
    # bf16
    yb = TO.dropout(torch.ones(1 << 16, device=DEV, dtype=torch.bfloat16), 0.1, seed=5)
    assert abs((yb != 0).float().mean().item() - 0.9) < 0.01


def test_attention_probs_dropout_matches_masked_reference():
    """AttentionCoreFn with dropout on the probabilities == torch attention with the SAME mask injected."""
    from tvretrieval_amd.autograd import AttentionCoreFn
    from tvretrieval_amd import train_ops as TO
    n, l, hsz, heads, p, seed = 2, 24, 128, 4, 0.2, 4242
    dh = hsz // heads
    q, k, v = rnd(n, l, hsz, seed=1), rnd(n, l, hsz, seed=2), rnd(n, l, hsz, seed=3)
    km = lens_mask(n, l, seed=4, lo=5)
    mask = TO.dropout(torch.ones(n * heads, l, l, device=DEV), p, seed).view(n, heads, l, l)    # l % 8 == 0: no padding

    def ref(q, k, v):
        add = (1 - km[:, None, None, :]) * -10000.0
        sp = lambda t: t.view(n, l, heads, dh).permute(0, 2, 1, 3)     # noqa: E731
        s = torch.matmul(sp(q), sp(k).transpose(-1, -2)) / math.sqrt(dh) + add
        o = torch.matmul(torch.softmax(s, -1) * mask, sp(v))
        return o.permute(0, 2, 1, 3).reshape(n, l, hsz)
    run_pair(lambda q, k, v: AttentionCoreFn.apply(q, k, v, None, km, heads, p, seed), ref, [q, k, v],
             [True, True, True], tol=5e-5)


@pytest.mark.parametrize("rows,n,k", [(12800, 768, 768), (3840, 2304, 768), (1000, 200, 72), (37, 8, 3072), (300, 768, 3080),
                                      # around the XCD-partitioned kernel (96 x 192 tiles; up to 32 tiles from 2048 rows, up to 96
                                      # from 8192): ragged row counts, 2 and 3 tiles per workgroup, fewer than 32 tiles per XCD
                                      # (its workgroups split the rows further), and the shapes just outside its window
                                      (12801, 768, 768), (2077, 96, 192), (5003, 1536, 768), (4099, 384, 384),
                                      (6400, 768, 3072), (2048, 192, 192), (8200, 1536, 768), (8193, 2304, 768)])
def test_weight_gradient_gemm_from_row_major_operands(rows, n, k):
    """xml_gemm_tn: dW = dY^T X read from the row-major bf16 operands (transpose reads, row ranges combined with atomics)
    == float64 matmul of the same bf16 values, and == the transpose + split-K path it replaces (up to summation order)."""
    from tvretrieval_amd import train_ops as TO
    dy = rnd(rows, n, seed=11).to(torch.bfloat16)
    x = rnd(rows, k, seed=12).to(torch.bfloat16)
    got, cs = TO.gemm_tn(dy, x, colsum=True)
    assert got is not None and got.shape == (n, k) and got.dtype == F32
    check("gemm_tn column sums", cs, dy.double().sum(0), 2e-5)
    want = dy.double().t() @ x.double()
    check("gemm_tn", got, want, 2e-5)
    r8 = (rows + 7) // 8 * 8
    old = TO.gemm_batched(TO.transpose(dy, r8), TO.transpose(x, r8), out_f32=True)
    check("gemm_tn vs transpose + split-K", got, old, 2e-5)
    assert TO.gemm_tn(dy.float(), x.float()) is None              # fp32 compute keeps the old path
    # accumulate mode (the optimizer's flat .grad views): out += dY^T X, colsum += column sums, no fill
    acc = torch.full((n, k), 0.25, device=DEV)
    acc_cs = torch.full((n,), -1.5, device=DEV)
    assert TO.gemm_tn(dy, x, out=acc, colsum_out=acc_cs)
    check("gemm_tn accumulate", acc, want + 0.25, 2e-5)
    check("gemm_tn accumulate column sums", acc_cs, dy.double().sum(0) - 1.5, 2e-5)


@pytest.mark.parametrize("shape", [(3, 100, 100, 768, 4), (2, 30, 30, 768, 4), (2, 40, 72, 256, 4), (2, 128, 128, 128, 4),
                                   (2, 17, 128, 768, 4)])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_fused_training_attention_vs_reference_and_unfused_chain(shape, p_drop):
    """attention_train.hip (bf16: one launch forward, one backward, P recomputed) against (1) float64 torch attention on
    the bf16-rounded inputs with the SAME dropout mask injected (bf16 tolerances) and (2) the unfused bf16 chain it
    replaces -- both the separate-q/k/v form with a query mask and the fused-QKV self-attention form."""
    from tvretrieval_amd.autograd import AttentionCoreFn, AttentionQkvFn
    from tvretrieval_amd import train_ops as TO
    n, lq, lk, hsz, heads = shape
    dh, seed = hsz // heads, 777
    bf = torch.bfloat16
    assert TO.attention_train_supported(lq, lk, hsz, heads, bf)
    q, k, v = (rnd(n, lq, hsz, seed=1).to(bf), rnd(n, lk, hsz, seed=2).to(bf), rnd(n, lk, hsz, seed=3).to(bf))
    km, qm = lens_mask(n, lk, seed=4, lo=5), lens_mask(n, lq, seed=5, lo=3)
    gout = rnd(n, lq, hsz, seed=6).to(bf)
    lq8, lk8 = (lq + 7) // 8 * 8, (lk + 7) // 8 * 8
    mask = torch.ones(n, heads, lq, lk, device=DEV, dtype=torch.float64)
    if p_drop > 0:
        mask = TO.dropout(torch.ones(n * heads, lq8, lk8, device=DEV), p_drop, seed).view(n, heads, lq8, lk8)[:, :, :lq, :lk].double()

    def run(fn, *ts):
        leaves = [t.clone().requires_grad_(True) for t in ts]
        y = fn(*leaves)
        y.backward(gout.to(y.dtype))
        return [y] + [t.grad for t in leaves]

    def ref(q, k, v):
        add = (1 - qm[:, None, :, None].double() * km[:, None, None, :].double()) * -10000.0
        sp = lambda t, l: t.view(n, l, heads, dh).permute(0, 2, 1, 3)     # noqa: E731
        s = torch.matmul(sp(q, lq), sp(k, lk).transpose(-1, -2)) / math.sqrt(dh) + add
        o = torch.matmul(torch.softmax(s, -1) * mask, sp(v, lk))
        return o.permute(0, 2, 1, 3).reshape(n, lq, hsz)
    want = run(ref, q.double(), k.double(), v.double())
    got = run(lambda a, b, c: AttentionCoreFn.apply(a, b, c, qm, km, heads, p_drop, seed), q, k, v)
    try:
        TO.DISABLE_FUSED_ATTENTION = True
        chain = run(lambda a, b, c: AttentionCoreFn.apply(a, b, c, qm, km, heads, p_drop, seed), q, k, v)
    finally:
        TO.DISABLE_FUSED_ATTENTION = False
    # rows of masked-out queries attend uniformly in both implementations; every row is compared
    for name, g, c, w in zip(("out", "dq", "dk", "dv"), got, chain, want):
        check("fused vs float64 " + name, g, w, 2e-2)
        check("unfused chain vs float64 " + name, c, w, 2e-2)
        check("fused vs unfused chain " + name, g, c, 2e-2)
    if lq == lk:          # self-attention on a fused (N, L, 3H) projection: same numbers through the other entry
        qkv = torch.cat([q, k, v], -1).contiguous()
        got2 = run(lambda t: AttentionQkvFn.apply(t, km, heads, p_drop, seed), qkv)
        want2 = run(lambda a, b, c: AttentionCoreFn.apply(a, b, c, None, km, heads, p_drop, seed), q, k, v)
        assert torch.equal(got2[0], want2[0])
        assert torch.equal(got2[1], torch.cat(want2[1:], -1))


def test_train_mode_step_with_dropout_runs():
    """model.train(): dropout active at every site; repeatable under torch.manual_seed, different from eval, finite."""
    from tvretrieval_amd.train import xml_forward_train
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    m = build_train_model(cfg, d).train()
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]), neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    torch.manual_seed(3)
    l1, _ = xml_forward_train(m, **batch)
    l1.backward()
    g1 = m.video_encoder1.self.query.weight.grad.clone()
    m.zero_grad()
    torch.manual_seed(3)
    l2, _ = xml_forward_train(m, **batch)
    l2.backward()
    # same masks -> same loss and gradients up to the summation order of the f32 atomics (loss reduction, split-K,
    # LayerNorm column sums); a different mask moves the loss by ~1e-2
    assert abs(float(l1) - float(l2)) < 1e-6
    assert torch.allclose(g1, m.video_encoder1.self.query.weight.grad, rtol=1e-4, atol=1e-7)
    assert math.isfinite(float(l1)) and abs(float(l1) - float(d["loss"])) > 1e-6
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


@pytest.mark.parametrize("rows,d,resid,p_in,p_out,dt", [
    (50, 128, True, 0.1, 0.0, F32), (33, 768, False, 0.0, 0.2, F32), (700, 384, True, 0.1, 0.1, F32),
    (9, 200, True, 0.3, 0.1, F32), (1300, 384, True, 0.1, 0.0, torch.bfloat16), (257, 768, False, 0.0, 0.1, torch.bfloat16),
    (5000, 384, True, 0.1, 0.1, torch.bfloat16)])
def test_layernorm_with_dropout_sites_vs_masked_reference(rows, d, resid, p_in, p_out, dt):
    """LayerNormFn with the dropout sites in front of `a` / behind the LayerNorm applied in its kernels
    (xml_add_layernorm_drop / xml_layernorm_bwd_drop) == float64 LayerNorm with xml_dropout's masks for the same seeds
    injected; f32: 5e-5 of the largest value, bf16 storage: 1.2e-2 (one bf16 rounding of y / of the gradients)."""
    from tvretrieval_amd import train_ops as TO
    from tvretrieval_amd.autograd import LayerNormFn
    si, so = 1234567, 7654321
    a = rnd(rows, d, seed=1).to(dt)
    b = rnd(rows, d, seed=2).to(dt) if resid else None
    g, beta = 1 + rnd(d, seed=3, scale=0.2), rnd(d, seed=4, scale=0.2)
    dy = rnd(rows, d, seed=5).to(dt)
    ones = torch.ones(rows, d, device=DEV)
    m_in = TO.dropout(ones, p_in, si).double() if p_in else ones.double()
    m_out = TO.dropout(ones, p_out, so).double() if p_out else ones.double()
    assert TO.layernorm_drop_supported(d, dt, True, resid, p_in)
    la = a.clone().requires_grad_(True)
    lb = b.clone().requires_grad_(True) if resid else None
    lg, lbeta = g.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = LayerNormFn.apply(la, lb, lg, lbeta, dt, (p_in, si) if p_in else None, (p_out, so) if p_out else None)
    y.backward(dy)
    ra = a.double().requires_grad_(True)
    rb = b.double().requires_grad_(True) if resid else None
    rg, rbeta = g.double().requires_grad_(True), beta.double().requires_grad_(True)
    x = ra * m_in + (rb if resid else 0)
    ry = F.layer_norm(x, (d,), rg, rbeta, 1e-5) * m_out
    ry.backward(dy.double())
    tol = 5e-5 if dt == F32 else 1.2e-2
    check("y", y, ry, tol)
    assert torch.equal(y == 0, (ry == 0) | (y == 0)) and int((y == 0).sum()) >= int((m_out == 0).sum())
    check("da", la.grad, ra.grad, tol)
    assert torch.equal(la.grad[m_in == 0], torch.zeros_like(la.grad[m_in == 0]))
    if resid:
        check("db", lb.grad, rb.grad, tol)
    check("dgamma", lg.grad, rg.grad, 5e-5 if dt == F32 else 6e-3)
    check("dbeta", lbeta.grad, rbeta.grad, 5e-5 if dt == F32 else 6e-3)


@pytest.mark.parametrize("rows,d,a_dt", [(5000, 3072, F32), (37, 3072, torch.bfloat16), (301, 2056, F32)])
def test_wide_layernorm_output_dropout_parameters_only(rows, d, a_dt):
    """3072-d input LayerNorm followed by the LinearLayer dropout (xml/model_components.py:103-114) in one launch each
    way: forward == dropout(LN) with the injected mask, dgamma / dbeta == the float64 sums over the masked dy."""
    from tvretrieval_amd import train_ops as TO
    p, seed = 0.1, 424242
    a = (rnd(rows, d, seed=1) * 2 + 0.3).to(a_dt)
    dy = rnd(rows, d, seed=2).to(torch.bfloat16)
    g, beta = 1 + rnd(d, seed=3, scale=0.2), rnd(d, seed=4, scale=0.2)
    assert TO.layernorm_drop_supported(d, torch.bfloat16, False, False, 0.0)
    assert not TO.layernorm_drop_supported(d, torch.bfloat16, True, False, 0.0)
    mask = TO.dropout(torch.ones(rows, d, device=DEV), p, seed).double()
    y = TO.add_layernorm_drop(a, None, g, beta, torch.bfloat16, 0.0, 0, p, seed)
    x = a.double()
    xh = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    check("y", y, (xh * g.double() + beta.double()) * mask, 1e-2)
    dx, dxa, dg, db = TO.layernorm_bwd_drop(a, None, g, dy, 0.0, 0, p, seed, need_dx=False)
    assert dx is None and dxa is None
    check("dgamma", dg, (dy.double() * mask * xh).sum(0), 2e-5)
    check("dbeta", db, (dy.double() * mask).sum(0), 2e-5)


def test_train_step_fused_dropout_sites_equal_separate_launches():
    """train.FUSE_DROPOUT on / off under the same torch seed: the same masks at every site (the seeds are drawn in the same
    order), so the fp32 loss and gradients agree to summation-order noise."""
    import tvretrieval_amd.train as TR
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    m = build_train_model(cfg, d).train()
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]), neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    res = []
    try:
        for fuse in (True, False):
            TR.FUSE_DROPOUT = fuse
            m.zero_grad()
            torch.manual_seed(11)
            loss, _ = TR.xml_forward_train(m, **batch)
            loss.backward()
            res.append((float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        TR.FUSE_DROPOUT = True
    assert abs(res[0][0] - res[1][0]) < 2e-6 * max(1.0, abs(res[1][0])), (res[0][0], res[1][0])
    assert abs(res[0][0] - float(d["loss"])) > 1e-6            # (dropout really on)
    assert res[0][1].keys() == res[1][1].keys()
    for n, g0 in res[0][1].items():
        if n.endswith(".key.bias"):        # softmax is invariant to the key bias: its gradient is rounding noise around 0
            continue
        check(n, g0, res[1][1][n], 2e-4)


def test_transpose_segments_one_launch():
    """xml_transpose_segments: bf16 W^T blocks of several f32 matrices of one flat buffer, incl. three matrices written side
    by side into one (K, 3H) block (fused QKV) and a row count that is not a multiple of 8 (zero-padded leading dimension)."""
    from tvretrieval_amd import train_ops as TO
    shapes = [(128, 384), (100, 64), (2, 128), (96, 96), (96, 96), (96, 96), (770, 130)]
    offs, total = [], 0
    for n, k in shapes:
        offs.append(total)
        total += (n * k + 3) // 4 * 4
    flat = rnd(total, seed=5)
    bufs, rows, tiles = [], [], 0
    groups = [(0,), (1,), (2,), (3, 4, 5), (6,)]
    for grp in groups:
        n_tot, k = sum(shapes[i][0] for i in grp), shapes[grp[0]][1]
        buf = torch.zeros(k, (n_tot + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
        col = 0
        for i in grp:
            n = shapes[i][0]
            rows.append([offs[i], n, k, buf.data_ptr(), buf.shape[1], col])
            tiles = max(tiles, ((n + 63) // 64) * ((k + 63) // 64))
            col += n
        bufs.append(buf)
    TO.transpose_segments(flat, torch.tensor(rows, dtype=torch.int64, device=DEV), tiles)
    for grp, buf in zip(groups, bufs):
        w = torch.cat([flat[offs[i]:offs[i] + shapes[i][0] * shapes[i][1]].view(shapes[i]) for i in grp], 0)
        want = torch.zeros_like(buf)
        want[:, :w.shape[0]] = w.to(torch.bfloat16).t()
        assert torch.equal(buf, want), grp


def test_weight_shadows_equal_per_use_conversion_and_never_go_stale():
    """bf16 steps with the optimizer's per-step weight copies (BertAdam.refresh_shadows: one bf16 copy of the flat buffer, one
    launch of transposes) == steps that convert / transpose every weight at its use: the GEMM operands are the same bits,
    so losses agree to the summation-order noise of the f32 atomics.  A weight written through torch after the refresh is
    NOT served from the shadow."""
    import tvretrieval_amd.train as TR
    from tvretrieval_amd.autograd import LinearFn
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]), neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])
    runs = []
    try:
        for shadows in (True, False):
            TR.SHADOW_WEIGHTS = shadows
            m = build_train_model(cfg, d, torch.bfloat16)
            opt = TR.BertAdam(m.parameters(), lr=5e-4, warmup=-1, t_total=-1, schedule="none")
            runs.append([float(TR.train_step(m, opt, batch)[0]) for _ in range(6)])
            if shadows:
                sh = opt._shadow
                assert sh is not None and sh["table"] is not None and len(sh["t"]) >= 8 and not sh["fresh"]
                # every kept transpose is the transpose of the CURRENT masters after a refresh
                opt.refresh_shadows(torch.bfloat16)
                for idx, ent in sh["t"].items():
                    assert ent["ready"]
                    w = torch.cat([opt.params[i].detach() for i in idx], 0).to(torch.bfloat16)
                    assert torch.equal(ent["buf"][:, :w.shape[0]], w.t()), idx
                # a torch-side write after the refresh: the shadow of that tensor is refused, the fresh value is used
                lin = m.video_query_linear
                x = rnd(16, lin.weight.shape[1], seed=3).to(torch.bfloat16)
                y0 = LinearFn.apply(x, lin.weight, lin.bias, False)
                with torch.no_grad():
                    lin.weight.mul_(2.0)
                y1 = LinearFn.apply(x, lin.weight, lin.bias, False)
                want = F.linear(x.float(), lin.weight.detach().to(torch.bfloat16).float(), lin.bias.detach())
                check("after the write", y1, want, 1e-2)
                assert rel_err(y0, want) > 0.2
    finally:
        TR.SHADOW_WEIGHTS = True
    for a, b in zip(*runs):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), runs
    assert runs[0][-1] < runs[0][0] - 0.02


def test_two_stream_step_survives_a_caller_that_drops_its_batch():
    """The subtitle branch runs on a side stream (train.PARALLEL_BRANCHES) and keeps the caller's feature / mask tensors for
    its backward kernels.  A caller that passes temporaries lets the allocator take those blocks back as soon as autograd
    releases them -- before the side stream's kernels have run, unless encode_context_train recorded the side stream on them.
    Made deterministic here: the side stream is stalled in front of the backward pass, and the main stream refills every freed
    block with garbage right after backward() returns.  Gradients must equal those of the one-stream run."""
    import tvretrieval_amd.train as TR
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    if not cfg.get("cross_att", False):
        pytest.skip("fixture without cross attention: no two-stream path")
    m = build_train_model(cfg, d)
    names = ("query_feat", "query_mask", "video_feat", "video_mask", "sub_feat", "sub_mask", "st_ed_indices")
    kw = dict(neg_ctx_rank=d["neg_ctx_rank"], neg_q_rank=d["neg_q_rank"])

    def grads():
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    try:
        TR.PARALLEL_BRANCHES = False
        m.zero_grad()
        TR.xml_forward_train(m, **{k: T(d[k]) for k in names}, **kw)[0].backward()
        want = grads()
        TR.PARALLEL_BRANCHES = True
        side = TR._side_stream(DEV)
        for _ in range(3):
            m.zero_grad()
            loss, _ = TR.xml_forward_train(m, **{k: T(d[k]) for k in names}, **kw)      # temporaries: autograd holds the last reference
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                torch.cuda._sleep(200_000_000)                                           # ~0.1 s: the side stream falls behind
            loss.backward()
            junk = [torch.full((int(d[k].size),), 7e3, device=DEV) for k in names for _ in range(4)]      # reuse what was freed
            torch.cuda.synchronize()
            del junk
            got = grads()
            for n, g in want.items():
                if not n.endswith(".key.bias"):
                    check(n, got[n], g, 5e-5)
    finally:
        TR.PARALLEL_BRANCHES = True


def test_global_grad_clip():
    from tvretrieval_amd import train_ops as TO
    for scale, n in ((5.0, 100003), (1e-4, 4096)):          # clipped / left alone
        g = rnd(n, seed=3, scale=scale)
        want = g.clone()
        torch.nn.utils.clip_grad_norm_([torch.nn.Parameter(want)], 1.0)      # reference semantics on a plain tensor
        p = torch.nn.Parameter(torch.zeros_like(g)); p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        got = TO.clip_grad_norm(g.clone(), 1.0)
        check("clip", got, p.grad, 2e-6)


def test_fused_qkv_self_attention():
    """QkvFn + AttentionQkvFn (one stacked projection GEMM, fused column blocks) == three Linear + attention in torch."""
    from tvretrieval_amd.autograd import AttentionQkvFn, QkvFn
    n, l, hsz, heads = 3, 21, 128, 4
    x = rnd(n, l, hsz, seed=1)
    ws = [rnd(hsz, hsz, seed=10 + i, scale=0.1) for i in range(3)]
    bs = [rnd(hsz, seed=20 + i, scale=0.1) for i in range(3)]
    km = lens_mask(n, l, seed=4, lo=3)

    def hip(x, wq, bq, wk, bk, wv, bv):
        return AttentionQkvFn.apply(QkvFn.apply(x, wq, bq, wk, bk, wv, bv), km, heads)

    def ref(x, wq, bq, wk, bk, wv, bv):
        return ref_attention(F.linear(x, wq, bq), F.linear(x, wk, bk), F.linear(x, wv, bv), None, km, heads)
    args = [x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2]]
    # the key bias has an analytically zero gradient (softmax is shift invariant): only rounding noise on both sides
    run_pair(hip, ref, args, [True, True, True, True, False, True, True], tol=5e-5)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_cross_attention_stacked_key_value_projection(p_drop):
    """bf16 cross attention with key + value of the other stream as ONE stacked projection (QkvFn with two weights +
    AttentionKvFn) == two LinearFn + AttentionCoreFn: same GEMM operands column for column, same dropout mask (same seed),
    so outputs are bitwise equal and gradients agree to the summation order of the f32 atomics."""
    from tvretrieval_amd.autograd import AttentionCoreFn, AttentionKvFn, LinearFn, QkvFn
    from tvretrieval_amd import train_ops as TO
    bf = torch.bfloat16
    n, lq, lk, hsz, heads, seed = 3, 100, 100, 768, 4, 4242
    assert TO.attention_train_supported(lq, lk, hsz, heads, bf)
    q0, side0 = rnd(n, lq, hsz, seed=1).to(bf), rnd(n, lk, hsz, seed=2).to(bf)
    ws = [rnd(hsz, hsz, seed=10 + i, scale=0.05) for i in range(2)]
    bs = [rnd(hsz, seed=20 + i, scale=0.1) for i in range(2)]
    qm, km = lens_mask(n, lq, seed=4, lo=3), lens_mask(n, lk, seed=5, lo=5)
    gout = rnd(n, lq, hsz, seed=6).to(bf)
    res = []
    for fused in (True, False):
        q, side = q0.clone().requires_grad_(True), side0.clone().requires_grad_(True)
        wk, wv = (w.clone().requires_grad_(True) for w in ws)
        bk, bv = (b.clone().requires_grad_(True) for b in bs)
        if fused:
            out = AttentionKvFn.apply(q, QkvFn.apply(side, wk, bk, wv, bv), qm, km, heads, p_drop, seed)
        else:
            out = AttentionCoreFn.apply(q, LinearFn.apply(side, wk, bk, False), LinearFn.apply(side, wv, bv, False), qm, km,
                                        heads, p_drop, seed)
        out.backward(gout)
        res.append((out.detach(), q.grad, side.grad, wk.grad, wv.grad, bv.grad))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])                      # dq: same kernel, same operands
    check("dside", res[0][2], res[1][2], 1.2e-2)                  # one GEMM over 2H vs two over H, added in bf16
    for name, i in (("dWk", 3), ("dWv", 4), ("dbv", 5)):
        check(name, res[0][i], res[1][i], 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 768, 768), (3, 104, 200), (1, 12800, 768), (2, 64, 64), (1, 50, 70),
                                   (2, 72, 3072), (1, 8, 8)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_batched_layouts(shape, dtype):
    """both transpose kernels (2-byte 32 x 32 tiles; 16-byte 64 x 64 tiles for bf16 shapes that are multiples of 8),
    with and without a padded output row"""
    from tvretrieval_amd import train_ops as T
    b, r, c = shape
    x = torch.randn(b, r, c, device="cuda").to(dtype)
    y = T.transpose(x)
    assert torch.equal(y, x.transpose(1, 2).contiguous())
    ld = (r + 15) // 8 * 8 + 8
    z = T.transpose(x, ld_out=ld)
    assert z.shape == (b, c, ld)
    assert torch.equal(z[:, :, :r], x.transpose(1, 2)) and float(z[:, :, r:].float().abs().max()) == 0.0


def test_graphed_step_construction_leaves_no_stale_weight_shadows():
    """GraphedTrainStep.__init__ runs warm-up steps on the real optimizer and restores the state afterwards; the per-step
    bf16 weight shadows it refreshed on the way belong to the LAST WARM-UP STEP, one update away from the restored masters.
    An eager no_grad forward between construction and the first replay (a validation loss) must not read them: its loss is
    the loss of a model that never saw an optimizer."""
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.train import BertAdam, GraphedTrainStep, xml_forward_train
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    ranks = dict(neg_ctx_rank=d["neg_ctx_rank_steps"][0], neg_q_rank=d["neg_q_rank_steps"][0])
    torch.manual_seed(5)
    ref = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        want = float(xml_forward_train(ref, **batch, **ranks)[0])
    torch.manual_seed(5)
    m = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).eval()
    opt = BertAdam(m.parameters(), lr=5e-2, warmup=-1, t_total=-1, schedule="none")     # a step moves the weights visibly
    GraphedTrainStep(m, opt, batch, warmup_steps=2)
    assert opt._shadow is None or not opt._shadow["fresh"]
    with torch.no_grad():
        got = float(xml_forward_train(m, **batch, **ranks)[0])
    assert abs(got - want) < 1e-6, (got, want)


def test_backward_after_an_intervening_step_is_refused_and_f16s_training_says_so():
    """forward A, full step B (which overwrites the optimizer's per-step weight shadows in place), backward A: A's saved
    weights are a view of that buffer and would silently be B's -- the backward must refuse.  And ops.F16S (the exact-rank
    mode's inference model) is rejected by the training forward with a message, not deep inside the graph."""
    from tvretrieval_amd import ops
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.train import BertAdam, xml_forward_train
    d, cfg, _ = load_golden("train_step_video_sub_h128")
    batch = dict(query_feat=T(d["query_feat"]), query_mask=T(d["query_mask"]), video_feat=T(d["video_feat"]),
                 video_mask=T(d["video_mask"]), sub_feat=T(d["sub_feat"]), sub_mask=T(d["sub_mask"]),
                 st_ed_indices=T(d["st_ed_indices"]))
    ranks = dict(neg_ctx_rank=d["neg_ctx_rank_steps"][0], neg_q_rank=d["neg_q_rank_steps"][0])
    torch.manual_seed(6)
    m = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).eval()
    opt = BertAdam(m.parameters(), lr=1e-3, warmup=-1, t_total=-1, schedule="none")
    loss_a = xml_forward_train(m, **batch, **ranks)[0]
    loss_b = xml_forward_train(m, **batch, **ranks)[0]
    opt.zero_grad()
    loss_b.backward()
    opt.step()
    xml_forward_train(m, **batch, **ranks)            # refreshes the shadows for the new weights
    with pytest.raises(RuntimeError, match="before the last optimizer step"):
        loss_a.backward()
    m16 = XML(cfg, compute_dtype=ops.F16S).to(DEV).eval()
    with pytest.raises(ValueError, match="inference-only"):
        xml_forward_train(m16, **batch, **ranks)
