"""Split-f16 forms (XML_F16 / XML_F16S, include/xmlhip.h "Exact-rank mode on the 16-bit pipe"): f32-grade results on the
16-bit MFMA pipe.  Every kernel against a float64 restatement of what it must compute, the ops.F16S model against the plain
f32 model, and the exact-rank mode built on them against the plain f32 path and the oracle."""
import numpy as np
import pytest
import torch

from oracle import xml_oracle as O
from oracle.listcmp import moment_keys, tie_aware_equal
from test_gpu_kernels import DEV, close
from test_gpu_model import _feats, _synthetic_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tvretrieval_amd import ops as o
    o._lib.load()
    return o


def _unit_rows(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(*shape, generator=g), dim=-1)


def _split_ref(x, log2s):
    """hi / lo halves of x * 2^log2s per row as float64 arrays, subnormal halves flushed -- the definition in split16.hip"""
    xs = x.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), log2s.double())[..., None]
    hi = xs.float().half()
    hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
    r = (xs - hi.double()).float()
    lo = r.half()
    lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi, lo, r


def test_split_rows_fixed_scale(ops):
    """unit-norm rows at the fixed scale 2^14: hi plane == rn_f16(x 2^14) bit for bit, err == || x - hi / S ||, and the
    halves give x back to 2^-22 relative (+ the flushed-subnormal floor 2^-14 / S)."""
    x = _unit_rows(700, 768, seed=1)
    x[13] = 0
    x[14, :700] = 0                    # a spiky row: few large elements
    x[14] = torch.nn.functional.normalize(x[14], dim=-1)
    sr, hi, err = ops.split_f16_rows(x.to(DEV), ops.F16_UNIT_LOG2, want_hi=True, want_err=True)
    want_hi, want_lo, r = _split_ref(x, torch.full((700,), 14.0))
    assert torch.equal(hi.cpu(), want_hi)
    assert torch.equal(sr.inv.cpu(), torch.full((700,), 2.0 ** -14))
    close("rounding-error norms", err, (r.double().norm(dim=-1) * 2.0 ** -14).float(), 1e-10, 1e-5)
    assert float(err[13]) == 0.0 and 5e-5 < float(err.mean()) < 3e-4          # ~1/8 of the bf16 filter's 0.8e-3
    back = ops.unsplit_f16_rows(sr).cpu()
    assert float((back.double() - (want_hi.double() + want_lo.double()) * 2.0 ** -14).abs().max()) == 0.0
    lim = 2.0 ** -21 * x.abs() + 2.0 ** -28
    assert bool(((back - x).abs() <= lim).all())
    # interleaved layout: per 32 elements [32 x hi | 32 x lo]
    raw = sr.data.cpu().view(torch.float16).view(700, 768 // 32, 2, 32)
    assert torch.equal(raw[:, :, 0].reshape(700, 768), want_hi) and torch.equal(raw[:, :, 1].reshape(700, 768), want_lo)


def test_split_rows_dynamic_scale(ops):
    """per-row power-of-two scales: rows of very different magnitude all keep 2^-22 relative precision."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 256, generator=g) * torch.logspace(-6, 4, 300)[:, None]
    x[7] = 0
    sr = ops.split_f16_rows(x.to(DEV))
    inv = sr.inv.cpu()
    mx = x.abs().amax(1)
    ok = mx > 0
    assert bool((inv[~ok] == 1).all())
    assert bool(((mx[ok] / inv[ok] >= 2.0 ** 13) & (mx[ok] / inv[ok] < 2.0 ** 14)).all())           # row maximum -> [2^13, 2^14)
    assert bool((torch.log2(inv) == torch.log2(inv).round()).all())
    back = ops.unsplit_f16_rows(sr).cpu()
    lim = 2.0 ** -21 * x.abs() + (2.0 ** -14 * inv)[:, None] * 1.01
    assert bool(((back - x).abs() <= lim).all())
    rel = ((back - x).abs().amax(1)[ok] / mx[ok])
    assert float(rel.max()) < 2.0 ** -21


@pytest.mark.parametrize("rows,n,k,relu", [(1000, 768, 768, False), (300, 256, 3072, True), (37, 128, 64, False),
                                           (5000, 2304, 768, False)])
def test_linear_f16s_vs_float64(ops, rows, n, k, relu):
    """y = x W^T + b through the split-f16 projection (both GEMM kernels: 256 x 256 LDS-DMA and 128 x 128) against float64:
    f32-grade (a few 1e-7 of |x||w|), far from bf16's 1e-2."""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, k, generator=g) * torch.logspace(-2, 2, rows)[:, None]
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    b = torch.randn(n, generator=g)
    want = x.double() @ w.double().t() + b.double()
    if relu:
        want = want.clamp_min(0)
    sw = ops.pack_weights_f16s(w.to(DEV))
    got = ops.linear(x.to(DEV), sw, b.to(DEV), relu=relu).cpu()
    scale = (x.norm(dim=1, keepdim=True) * w.norm(dim=1)[None]).double()              # |x||w| per output
    err = ((got.double() - want).abs() / scale).max()
    f32 = ((x.to(DEV) @ w.to(DEV).t() + b.to(DEV)).cpu().double() - (x.double() @ w.double().t() + b.double())).abs() / scale
    print("linear_f16s %dx%dx%d: max err / (|x||w|) = %.2e   (torch f32 GEMM: %.2e)" % (rows, n, k, float(err), float(f32.max())))
    assert float(err) < max(8e-7, 2.0 * float(f32.max()))            # f32-grade: the accuracy of an f32 GEMM


@pytest.mark.parametrize("ctx_mode", ["video_sub", "video"])
def test_f16s_model_matches_f32_model(ops, ctx_mode):
    """XML(compute_dtype=ops.F16S) -- every projection a split-f16 product -- returns the f32 model's encoder outputs and
    query vectors (same weights) to f32 rounding, for ragged batches and both query paths (padded / packed tokens)."""
    from tvretrieval_amd import model_xml
    l, hidden = 64, 128
    m32, cfg = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, torch.float32, seed=5)
    m16, _ = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, ops.F16S, seed=5)
    m16.load_state_dict(m32.state_dict())
    rng = np.random.default_rng(0)
    nv = 24
    lens = rng.integers(5, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    a = [t.to(DEV) for t in (vf, vm, sf, sm)]
    if ctx_mode == "video":
        a[2] = a[3] = None
    with torch.no_grad():
        want = m32.encode_context(*a)
        got = m16.encode_context(*a)
    for nm, g_, w_ in zip(("vf1", "vf2", "sf1", "sf2"), got, want):
        if w_ is None:
            assert g_ is None
            continue
        assert g_.dtype == torch.float32
        valid = vm.bool()
        close("F16S %s (valid clips)" % nm, g_.cpu()[valid], w_.cpu()[valid], 6e-6, 1e-5)
        # padded positions: a row whose keys are all masked sees scores s - 10000, where f32 keeps 1e-3 of s -- the
        # reference's own values there move by 1e-4 under any 1e-7 change of the projections
        close("F16S %s (padded clips)" % nm, g_.cpu()[~valid], w_.cpu()[~valid], 1e-3, 0)
    for nq in (40, model_xml.PACK_MIN_ROWS // 30 + 20):          # padded path, packed-token path
        qf, qm = _feats(nq, rng.integers(3, 31, nq), 128, 7)
        with torch.no_grad():
            wq = m32.encode_query(qf.to(DEV), qm.to(DEV))
            gq = m16.encode_query(qf.to(DEV), qm.to(DEV))
        for g_, w_ in zip(gq, wq):
            close("F16S query vectors (nq=%d)" % nq, g_, w_, 6e-6, 1e-5)
        # query linears (ConvSE side)
        for name in (["video", "sub"] if ctx_mode == "video_sub" else ["video"]):
            with torch.no_grad():
                close("F16S %s_query_linear" % name, getattr(m16, name + "_query_linear")(wq[0].contiguous()),
                      getattr(m32, name + "_query_linear")(wq[0].contiguous()), 4e-6, 1e-5)


@pytest.mark.parametrize("n_mod,ragged", [(2, False), (2, True), (1, True)])
def test_k6_f16_filter_and_rescore_f16s(ops, n_mod, ragged):
    """K6 on the f16 hi planes == the float64 dot products of those planes / 2^28 (f16 x f16 products are exact in f32),
    within the Cauchy-Schwarz bound e_q + e_c of the true scores; xml_q2c_rescore on the split rows == the true scores to
    f32 rounding."""
    nq, nv, l, hidden = 300, 70, 128, 256
    g = torch.Generator().manual_seed(11)
    lens = torch.randint(10, l + 1, (nv,), generator=g) if ragged else torch.full((nv,), l)
    lens[0] = l
    mask = (torch.arange(l)[None] < lens[:, None]).float()
    qn = [_unit_rows(nq, hidden, seed=30 + m) for m in range(n_mod)]
    cn = [_unit_rows(nv, l, hidden, seed=40 + m) * mask[..., None] for m in range(n_mod)]
    q_sr, q_hi, eq, c_sr, c_tiles, ec = [], [], [], [], [], []
    for m in range(n_mod):
        s_, h_, e_ = ops.split_f16_rows(qn[m].to(DEV), ops.F16_UNIT_LOG2, want_hi=True, want_err=True)
        q_sr.append(s_), q_hi.append(h_), eq.append(e_)
        s_, h_, e_ = ops.split_f16_rows(cn[m].to(DEV), ops.F16_UNIT_LOG2, want_hi=True, want_err=True)
        c_sr.append(s_), ec.append(float(e_.max()))
        c_tiles.append(ops.pack_q2c_corpus(h_, mask.to(DEV), None, normalize=False))
    assert isinstance(c_tiles[0], ops.TiledRows) and c_tiles[0].dtype == torch.float16
    masks = [mask.to(DEV)] * n_mod
    filt = ops.q2c_scores_fused(q_hi, c_tiles, masks).cpu()

    def scores(qs, cs):
        tot = 0
        for q, c in zip(qs, cs):
            s = torch.einsum("md,nld->mln", q.double(), c.double())
            s = s * mask.double().t()[None] + (1 - mask.double().t()[None]) * -1e10
            tot = tot + s.max(1)[0]
        return tot / len(qs)
    true = scores(qn, cn)
    hi_only = scores([h.cpu().double() * 2.0 ** -14 for h in q_hi], [t.to_rows().cpu().double() * 2.0 ** -14 for t in c_tiles])
    close("f16 filter == dot products of the hi planes", filt, hi_only.float(), 2e-7, 0)
    bound = sum(float(e.max()) + c for e, c in zip(eq, ec)) / n_mod + 1e-6
    assert float((filt.double() - true).abs().max()) < bound, "filter error beyond the certificate's bound"
    assert float((filt.double() - true).abs().max()) > 1e-6          # (it IS a 16-bit pass)
    pair = torch.randint(0, nv, (nq, 9), generator=g).int()
    pair[:, 0] = 3
    pair[5, 1] = -1
    got = ops.q2c_rescore(q_sr, c_sr, masks, pair.to(DEV)).cpu()
    ok = pair >= 0
    want = torch.gather(true, 1, pair.clamp_min(0).long())
    assert torch.isinf(got[~ok]).all()
    err = float((got[ok].double() - want[ok]).abs().max())
    print("split-f16 re-score: max abs err %.2e (scores ~ %.2f)" % (err, float(want[ok].abs().mean())))
    assert err < 4e-7


def test_rescore_f16s_128_row_chunks(ops):
    """more than 64 pairs per video on average: the re-score takes its 128-row chunks (one tile fetch per video instead of
    two) -- same values as the 64-row kernel computes for the same pairs, and the true scores to f32 rounding."""
    nq, nv, l, hidden, kp = 500, 40, 128, 128, 9          # 4500 pairs over 40 videos: ~112 per video
    g = torch.Generator().manual_seed(21)
    lens = torch.randint(10, l + 1, (nv,), generator=g); lens[0] = l
    mask = (torch.arange(l)[None] < lens[:, None]).float()
    qn = [_unit_rows(nq, hidden, seed=50 + m) for m in range(2)]
    cn = [_unit_rows(nv, l, hidden, seed=60 + m) * mask[..., None] for m in range(2)]
    q_sr = [ops.split_f16_rows(q.to(DEV), ops.F16_UNIT_LOG2) for q in qn]
    c_sr = [ops.split_f16_rows(c.to(DEV), ops.F16_UNIT_LOG2) for c in cn]
    pair = torch.randint(0, nv, (nq, kp), generator=g).int()
    pair[:, 0] = 3                                          # one video listed by every query: 500 pairs = 4 chunks of 128
    pair[:, 1] = torch.where(torch.arange(nq) % 7 == 0, torch.full((nq,), 5), pair[:, 1]).int()     # ... and a 72-pair one
    pair[9, 4] = -1
    masks = [mask.to(DEV)] * 2
    big = ops.q2c_rescore(q_sr, c_sr, masks, pair.to(DEV)).cpu()
    true = 0
    for q, c in zip(qn, cn):
        s_ = torch.einsum("md,nld->mln", q.double(), c.double())
        s_ = s_ * mask.double().t()[None] + (1 - mask.double().t()[None]) * -1e10
        true = true + s_.max(1)[0]
    want = torch.gather(true / 2, 1, pair.clamp_min(0).long())
    ok = pair >= 0
    assert torch.isinf(big[~ok]).all()
    assert float((big[ok].double() - want[ok]).abs().max()) < 4e-7
    # the same pairs through 64-row chunks: few pairs per call (P <= 64 nv)
    small = torch.cat([ops.q2c_rescore([ops.SplitRows(s.data[b:b + 250].contiguous(), s.inv[b:b + 250].contiguous())
                                        for s in q_sr], c_sr, masks, pair[b:b + 250].to(DEV).contiguous()).cpu()
                       for b in (0, 250)])
    assert torch.equal(small[ok], big[ok])                  # same products, same accumulation order per (pair, clip)


@pytest.mark.parametrize("merged,n_mod", [(True, 2), (False, 2), (False, 1)])
def test_convse_f16s_vs_f32_kernel(ops, merged, n_mod):
    """xml_convse_rerank_f16s (split rows, per-row scales, modality accumulation across DIFFERENT scales) == the f32 kernel
    on the same values: masked logits and probabilities."""
    nq, nv, l, hidden, kp = 90, 23, 128, 128, 7
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(8, l + 1, (nv,), generator=g); lens[0] = l
    mask = (torch.arange(l)[None] < lens[:, None]).float()
    # rows of very different magnitude per modality: the scales of the two streams differ by many powers of two
    q_lin = [torch.randn(nq, hidden, generator=g) * (0.05 if m == 0 else 3.0) for m in range(n_mod)]
    feat2 = [torch.randn(nv, l, hidden, generator=g) * mask[..., None] * (2.0 if m == 0 else 0.01) *
             torch.logspace(-1, 1, l)[None, :, None] for m in range(n_mod)]
    conv_w = torch.randn(2 * (1 if merged else n_mod) * 5, generator=g) * 0.5
    pair = torch.randint(0, nv, (nq, kp), generator=g).int()
    pair[3, 2] = -1
    masks = [mask.to(DEV)] * n_mod
    for softmax in (False, True):
        w_st, w_ed = ops.convse_rerank([q.to(DEV) for q in q_lin], [f.to(DEV) for f in feat2], masks, pair.to(DEV),
                                       conv_w.to(DEV), l, merged, 5, softmax=softmax)
        g_st, g_ed = ops.convse_rerank([ops.split_f16_rows(q.to(DEV)) for q in q_lin],
                                       [ops.split_f16_rows(f.to(DEV)) for f in feat2], masks, pair.to(DEV),
                                       conv_w.to(DEV), l, merged, 5, softmax=softmax)
        for nm, a, b in (("st", g_st, w_st), ("ed", g_ed, w_ed)):
            a, b = a.cpu(), b.cpu()
            live = b > -1e9
            assert torch.equal(a > -1e9, live)
            scale = float(b[live].abs().max())
            close("convse f16s %s (softmax=%s)" % (nm, softmax), a[live], b[live], 3e-6 * max(scale, 1.0), 2e-6)


def _lists_equal(out, ref, l, kv, kn, what):
    gi = out["top_indices"].cpu().numpy()
    ww, wi = torch.topk(torch.exp(20.0 * ref["q2c"]), min(kv + 8, ref["q2c"].shape[1]), dim=1)
    n_v = tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi.cpu().numpy(), ww.cpu().numpy(), kv, 2e-5, what + " videos")
    ri = ref["top_indices"].cpu().numpy()
    same = np.nonzero((gi == ri).all(1))[0]
    gk = moment_keys(out["flat_indices"].cpu().numpy(), gi, l)
    wk = moment_keys(ref["flat_indices"].cpu().numpy(), ri, l)
    n_m = tie_aware_equal(gk[same], out["flat_scores"].cpu().numpy()[same], wk[same], ref["flat_scores"].cpu().numpy()[same],
                          kn - 8, 5e-5, what + " moments")
    return n_v, n_m, len(same)


@pytest.mark.parametrize("ctx_mode,ragged,filt", [("video_sub", False, "bf16"), ("video_sub", True, "bf16"),
                                                  ("video", False, "bf16"), ("video_sub", True, "f16"), ("video", False, "f16")])
def test_exact_mode_f16s_equals_f32_path(ctx_mode, ragged, filt, monkeypatch):
    """The split-f16 exact-rank mode (F16S model, f16 filter, split re-score, on-device second tier) returns the plain f32
    path's lists -- whatever the filter did: normal run, every certificate forced to fail (second tier), second tier
    overflowing (third tier), and with the check deferred (what a captured graph runs)."""
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops
    monkeypatch.setattr(inf, "EXACT_F16S_FILTER", filt)
    nq, nv, l, hidden = 64, 700, 128, 128
    m32, cfg = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, torch.float32, seed=3)
    m16, _ = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, ops.F16S, seed=3)
    m16.load_state_dict(m32.state_dict())
    rng = np.random.default_rng(1)
    lens = rng.integers(10, l + 1, nv) if ragged else np.full(nv, l)
    lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    qf, qm = _feats(nq, rng.integers(5, 31, nq), 128, 3)
    bs = 100

    def batches():
        for b in range(0, nv, bs):
            yield (vf[b:b + bs].to(DEV), vm[b:b + bs].to(DEV), sf[b:b + bs].to(DEV), sm[b:b + bs].to(DEV))
    with torch.no_grad():
        plain = inf.build_corpus_index(m32, batches(), l_ref=l)
        ref = inf.vcmr_search(m32, plain, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
        exact = inf.build_corpus_index(m16, batches(), l_ref=l, exact_filter=True)
        assert exact.exact.mode == "f16s"
        assert exact.feat1n[exact.modalities[0]].dtype == (torch.float16 if filt == "f16" else torch.bfloat16)
        assert exact.exact.n_candidates == (128 if filt == "f16" else 256)
        assert exact.feat2[exact.modalities[0]].dtype is ops.F16S
        exact.exact.n_candidates = 20
        out = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    info = out["exact"]
    n_v, n_m, n_same = _lists_equal(out, ref, l, 10, 200, "f16s exact vs f32")
    assert n_same >= nq - 2
    f32_q2c = ref["q2c"]
    d = float((info["q2c_filter"] - f32_q2c).abs().max())
    assert 1e-6 < d < float(info["eps"].min())                          # a real 16-bit pass, inside the bound
    close("re-scored candidates", info["cand_scores"], torch.gather(f32_q2c, 1, info["cand_indices"].long()), 3e-6)
    passed = (info["fail"] == 0).cpu().numpy()
    ci, wi = info["cand_indices"].cpu().numpy(), ref["top_indices"].cpu().numpy()
    for q in np.nonzero(passed)[0]:
        assert set(wi[q].tolist()) <= set(ci[q].tolist()), q
    print("f16s exact mode (%s, ragged=%s, %s filter): %d / %d queries failed the certificate; %d video / %d moment positions "
          "swapped in f32 ties; filter err %.1e, eps %.1e" % (ctx_mode, ragged, filt, info["n_fail"], nq, n_v, n_m, d,
                                                              float(info["eps"].mean())))

    # every certificate forced to fail: the on-device second tier re-scores every video above T_k - eps (eps = 10: all of them)
    exact.exact.e_c = {k: 10.0 for k in exact.exact.e_c}
    exact.exact.tier2_rows, exact.exact.tier2_cap = nq, nv
    with torch.no_grad():
        forced = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
        deferred = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200,
                                   defer_exact_check=True)
    assert forced["exact"]["n_fail"] == nq and forced["exact"]["n_full_rows"] == 0
    assert not bool(forced["exact"]["overflow_dev"].item())
    _lists_equal(forced, ref, l, 10, 200, "second tier vs f32")
    assert "n_fail" not in deferred["exact"] and int(deferred["exact"]["n_fail_dev"].item()) == nq
    for key in ("top_indices", "top_scores", "flat_indices", "flat_scores"):
        assert torch.equal(deferred[key], forced[key]), key
    # second tier too small on both axes: the overflow flag is raised and the eager pass re-scores against the whole corpus
    exact.exact.tier2_rows, exact.exact.tier2_cap = 5, 16
    with torch.no_grad():
        over = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
        deferred = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200,
                                   defer_exact_check=True)
    assert bool(over["exact"]["overflow_dev"].item()) and over["exact"]["n_full_rows"] == nq
    assert bool(deferred["exact"]["overflow_dev"].item())               # (the deferred pass only reports it)
    _lists_equal(over, ref, l, 10, 200, "third tier vs f32")
    exact.exact.tier2_rows, exact.exact.tier2_cap = nq, 16              # rows fit, candidate lists overflow
    with torch.no_grad():
        over = inf.vcmr_search(m16, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    assert over["exact"]["n_full_rows"] == nq
    _lists_equal(over, ref, l, 10, 200, "third tier (list overflow) vs f32")

    # and the oracle (reference formulation) on the same inputs
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m32.state_dict().items()})
    with torch.no_grad():
        f1v, f2v, f1s, f2s = [], [], [], []
        for b in range(0, nv, bs):
            o = om.encode_context(vf[b:b + bs], vm[b:b + bs], sf[b:b + bs] if om.use_sub else None,
                                  sm[b:b + bs] if om.use_sub else None)
            f1v.append(o[0]), f2v.append(o[1]), f1s.append(o[2]), f2s.append(o[3])
        cat = lambda xs: torch.cat(xs) if xs[0] is not None else None      # noqa: E731
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, cat(f1v), cat(f2v), vm, cat(f1s), cat(f2s),
                                                 sm if om.use_sub else None, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, 10, 2, 16, 216)
        ww, wi2 = torch.topk(torch.exp(20.0 * q2c), 18, dim=1)
    gi = out["top_indices"].cpu().numpy()
    tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi2.numpy(), ww.numpy(), 10, 2e-3, "f16s exact vs oracle videos")
    same = np.nonzero((gi == want["top_indices"].numpy()).all(1))[0]
    assert len(same) >= 0.9 * nq
    gk = moment_keys(out["flat_indices"].cpu().numpy(), gi, l)
    wk = moment_keys(want["flat_indices"].numpy(), want["top_indices"].numpy(), l)
    tie_aware_equal(gk[same], out["flat_scores"].cpu().numpy()[same], wk[same], want["flat_scores"].numpy()[same], 200,
                    5e-4, "f16s exact vs oracle moments")


def test_exact_mode_f16s_is_capturable():
    """GraphedVcmrSearch accepts a split-f16 exact index (the second tier runs on the device with fixed shapes): replays ==
    the eager exact pass, for batches that fail no / some certificates; an overflowing batch falls back to the eager pass."""
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops
    nq, nv, l = 50, 300, 128
    m16, cfg = _synthetic_model("video_sub", 128, 256, 128, 128, l, ops.F16S, seed=9)
    rng = np.random.default_rng(3)
    lens = rng.integers(20, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    with torch.no_grad():
        index = inf.build_corpus_index(m16, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))], exact_filter=True)
        index.exact.n_candidates = 16
        index.exact.e_c = {k: 4.0 * v for k, v in index.exact.e_c.items()}      # a looser bound: some certificates fail
        g = inf.GraphedVcmrSearch(m16, index, nq, 30, 128, max_vcmr_video=10, max_before_nms=100)
        n_failed = []
        for seed in (3, 4, 5):
            qf, qm = _feats(nq, np.concatenate([[30], rng.integers(5, 31, nq - 1)]), 128, seed)
            want = inf.vcmr_search(m16, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=100)
            n_failed.append(want["exact"]["n_fail"])
            want = {k: v.clone() for k, v in want.items() if torch.is_tensor(v)}
            got = g(qf.to(DEV), qm.to(DEV))
            for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
                assert torch.equal(got[k], want[k]), (seed, k)
        print("graphed f16s exact mode: certificates failed per batch:", n_failed)
        # capacity exceeded inside the graph -> the call answers through the eager pass
        index.exact.e_c = {k: 10.0 for k in index.exact.e_c}
        index.exact.tier2_rows = 4
        g2 = inf.GraphedVcmrSearch(m16, index, nq, 30, 128, max_vcmr_video=10, max_before_nms=100)
        want = inf.vcmr_search(m16, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=100)
        got = g2(qf.to(DEV), qm.to(DEV))
        assert want["exact"]["n_full_rows"] > 0
        for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
            assert torch.equal(got[k], want[k]), k
        # the driver's replayed batches (opt.graph_search) do not read the flag per batch: flagged batches are searched again
        # eagerly before the result sinks are fetched -- the lists of the eager driver, bit for bit
        import argparse

        class _Queries(object):
            video2idx = {"vid_%03d" % i: 1000 + i for i in range(nv)}

            def __init__(self, n):
                self.q = [_feats(1, [int(x)], 128, 900 + i)[0][0, :int(x)].numpy() for i, x in enumerate(rng.integers(5, 31, n))]

            def set_data_mode(self, mode):
                pass

            def load_gt_vid_name_for_query(self, flag):
                pass

            def __len__(self):
                return len(self.q)

            def __getitem__(self, i):
                return dict(meta=dict(desc_id=i, desc="q%d" % i, vid_name="vid_%03d" % (i % nv)),
                            model_inputs=dict(query_feat=self.q[i]))
        ds = _Queries(2 * nq + 7)
        ctx = dict(index=index, video_metas=[dict(vid_name="vid_%03d" % i) for i in range(nv)])
        opt = argparse.Namespace(eval_query_bsz=nq, device=torch.device(DEV), q2c_alpha=20.0, min_pred_l=2, max_pred_l=16,
                                 clip_length=1.5, debug=False, external_inference_vr_res_path=None, max_desc_l=30)
        eager = inf.compute_query2ctx_info(m16, ds, opt, ctx, max_before_nms=100, max_n_videos=10, tasks=("VCMR", "SVMR"),
                                           as_arrays=True)
        opt.graph_search = True
        graphed = inf.compute_query2ctx_info(m16, ds, opt, ctx, max_before_nms=100, max_n_videos=10, tasks=("VCMR", "SVMR"),
                                             as_arrays=True)
        for task in ("VCMR", "SVMR"):
            np.testing.assert_array_equal(eager[task].count, graphed[task].count)
            for col in ("vid", "st", "ed", "score"):
                np.testing.assert_array_equal(getattr(eager[task], col), getattr(graphed[task], col),
                                              err_msg="%s.%s: graphed batches with an overflowing second tier" % (task, col))


# ---- seeded sweeps over the supported shape space (one random case per test id) ----------------------------------------
@pytest.mark.parametrize("seed", range(12))
def test_fuzz_linear_f16s(ops, seed):
    """split-f16 projections over random (rows, n, k): both GEMM kernels, ragged row / column tiles, bias and ReLU, row
    magnitudes spread over six decades -- float64 is the judge, an f32 GEMM the yardstick."""
    rng = np.random.default_rng(21000 + seed)
    rows = int(rng.choice([1, 7, 130, 255, 256, 700, 3000]))
    n = int(rng.integers(1, 160)) * 8
    k = int(rng.integers(1, 200)) * 8
    relu = bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, k, generator=g) * torch.logspace(-3, 3, rows)[:, None]
    w = torch.randn(n, k, generator=g) * float(rng.choice([1e-3, 0.05, 3.0]))
    b = torch.randn(n, generator=g) if rng.integers(0, 2) else None
    want = x.double() @ w.double().t() + (b.double() if b is not None else 0)
    if relu:
        want = want.clamp_min(0)
    got = ops.linear(x.to(DEV), ops.pack_weights_f16s(w.to(DEV)), None if b is None else b.to(DEV), relu=relu).cpu()
    scale = (x.norm(dim=1, keepdim=True) * w.norm(dim=1)[None]).double().clamp_min(1e-30)
    err = float(((got.double() - want).abs() / scale).max())
    f32 = x.to(DEV) @ w.to(DEV).t() + (b.to(DEV) if b is not None else 0)
    f32 = (f32.clamp_min(0) if relu else f32).cpu()
    f32err = float(((f32.double() - want).abs() / scale).max())
    assert err <= max(8e-7, 2.5 * f32err), (rows, n, k, err, f32err)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_rescore_f16s(ops, seed):
    """xml_q2c_rescore on split rows over random clip paddings / hidden sizes / modalities / pair lists (skipped pairs,
    videos listed by many queries -> 64- and 128-row chunks) against float64."""
    rng = np.random.default_rng(22000 + seed)
    n_mod = int(rng.integers(1, 3))
    lpad = int(rng.choice([16, 48, 64, 112, 128]))
    hidden = int(rng.choice([32, 64, 128, 256, 384, 768]))
    nq, nv, kp = int(rng.integers(1, 400)), int(rng.integers(1, 60)), int(rng.integers(1, 12))
    g = torch.Generator().manual_seed(100 + seed)
    lens = torch.randint(1, lpad + 1, (nv,), generator=g)
    mask = (torch.arange(lpad)[None] < lens[:, None]).float()
    if nv > 2:
        mask[1] = 0
    qn = [_unit_rows(nq, hidden, seed=seed * 7 + m) for m in range(n_mod)]
    cn = [_unit_rows(nv, lpad, hidden, seed=seed * 11 + 3 + m) * mask[..., None] for m in range(n_mod)]
    pair = torch.randint(-1, nv + 1, (nq, kp), generator=g).int()
    pair[:, 0] = int(rng.integers(0, nv))
    got = ops.q2c_rescore([ops.split_f16_rows(q.to(DEV), ops.F16_UNIT_LOG2) for q in qn],
                          [ops.split_f16_rows(c.to(DEV), ops.F16_UNIT_LOG2) for c in cn], [mask.to(DEV)] * n_mod,
                          pair.to(DEV)).cpu()
    tot = 0
    for q, c in zip(qn, cn):
        s = torch.einsum("md,nld->mln", q.double(), c.double())
        s = s * mask.double().t()[None] + (1 - mask.double().t()[None]) * -1e10
        tot = tot + s.max(1)[0]
    ok = (pair >= 0) & (pair < nv)
    want = torch.gather(tot / n_mod, 1, pair.clamp(0, nv - 1).long())
    assert torch.isinf(got[~ok]).all() and bool((got[~ok] < 0).all())
    live = ok & (want > -1e9)
    assert float((got[live].double() - want[live]).abs().max()) < 5e-7 if live.any() else True
    dead = ok & ~live                                   # fully masked videos: -1e10 on both sides
    assert bool((got[dead] < -1e9).all())


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_convse_f16s(ops, seed):
    """xml_convse_rerank_f16s against the f32 kernel over clip paddings, reference lengths, kernel sizes, merged or
    per-stream predictors, skipped pairs and operand magnitudes that differ by many powers of two between the streams."""
    rng = np.random.default_rng(23000 + seed)
    n_mod = int(rng.integers(1, 3))
    merged = bool(n_mod == 2 and rng.integers(0, 2))
    lpad = int(rng.choice([16, 32, 64, 112, 128]))
    l_ref = int(rng.integers(max(1, lpad - 15), lpad + 1))
    hidden = int(rng.choice([32, 96, 128, 256, 768]))
    ksize = int(rng.choice([1, 3, 5, 7]))
    nq, nv, kp = int(rng.integers(1, 120)), int(rng.integers(1, 30)), int(rng.integers(1, 9))
    g = torch.Generator().manual_seed(300 + seed)
    lens = torch.randint(1, l_ref + 1, (nv,), generator=g)
    mask = (torch.arange(lpad)[None] < lens[:, None]).float()
    q_lin = [torch.randn(nq, hidden, generator=g) * float(rng.choice([0.01, 1.0, 40.0])) for _ in range(n_mod)]
    feat2 = [torch.randn(nv, lpad, hidden, generator=g) * mask[..., None] * float(rng.choice([0.02, 1.0, 8.0]))
             for _ in range(n_mod)]
    conv_w = torch.randn(2 * (1 if merged else n_mod) * ksize, generator=g) * 0.5
    pair = torch.randint(-1, nv, (nq, kp), generator=g).int()
    masks = [mask.to(DEV)] * n_mod
    lscale = 1.0
    for softmax in (False, True):
        w_st, w_ed = ops.convse_rerank([q.to(DEV) for q in q_lin], [f.to(DEV) for f in feat2], masks, pair.to(DEV),
                                       conv_w.to(DEV), l_ref, merged, ksize, softmax=softmax)
        g_st, g_ed = ops.convse_rerank([ops.split_f16_rows(q.to(DEV)) for q in q_lin],
                                       [ops.split_f16_rows(f.to(DEV)) for f in feat2], masks, pair.to(DEV),
                                       conv_w.to(DEV), l_ref, merged, ksize, softmax=softmax)
        for nm, a, b in (("st", g_st, w_st), ("ed", g_ed, w_ed)):
            a, b = a.cpu(), b.cpu()
            live = b > -1e9
            assert torch.equal(a > -1e9, live), (seed, nm)
            if not live.any():
                continue
            if not softmax:      # logits: a few f32 ulps of their own magnitude
                lscale = max(lscale, float(b[live].abs().max()))
                close("fuzz %d convse f16s %s logits" % (seed, nm), a[live], b[live], 4e-6 * lscale, 4e-6)
            else:                # probabilities: exp() turns a logit error d into a RELATIVE error d
                close("fuzz %d convse f16s %s probabilities" % (seed, nm), a[live], b[live], 1e-9, 8e-6 * lscale + 4e-6)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_f16s_search_vs_oracle(seed):
    """WHOLE searches on an ops.F16S model -- exact-rank mode when the corpus is large enough to filter, the plain path
    otherwise -- against the reference formulation on the CPU: context mode, cross attention, merged / per-stream predictors,
    hidden size, clip count, ragged lengths."""
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops
    rng = np.random.default_rng(24000 + seed)
    ctx_mode = str(rng.choice(["video_sub", "video_sub", "video", "sub"]))
    cross, merge = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    hidden = int(rng.choice([128, 256]))
    l = int(rng.choice([128, 100, 64, 40]))
    dv, ds_, dq = int(rng.choice([256, 512])), int(rng.choice([128, 256])), int(rng.choice([128, 256]))
    nv, nq = int(rng.integers(12, 70)), int(rng.integers(1, 40))
    kv = int(rng.integers(1, 6))
    n_mom = int(rng.integers(5, 120))
    m16, cfg = _synthetic_model(ctx_mode, hidden, dv, ds_, dq, l, ops.F16S, seed=400 + seed, cross=cross, merge=merge)
    lens = rng.integers(max(2, l // 6), l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, dv, 1 + seed)
    sf, sm = _feats(nv, lens, ds_, 2 + seed)
    qf, qm = _feats(nq, rng.integers(1, 31, nq), dq, 3 + seed)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m16.state_dict().items()})
    b = [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))]
    with torch.no_grad():
        index = inf.build_corpus_index(m16, b, exact_filter=True)
        index.exact.n_candidates = max(kv, min(nv - 1, 2 * kv + 3))
        out = inf.vcmr_search(m16, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=kv, max_before_nms=n_mom)
        o = om.encode_context(vf if om.use_video else None, vm if om.use_video else None, sf if om.use_sub else None,
                              sm if om.use_sub else None)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, o[0], o[1], vm if om.use_video else None, o[2], o[3],
                                                 sm if om.use_sub else None, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, kv, 2, 16, n_mom + 8)
        ww, wi = torch.topk(torch.exp(20.0 * q2c), min(kv + 6, nv), dim=1)
    gi = out["top_indices"].cpu().numpy()
    tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi.numpy(), ww.numpy(), kv, 2e-3, "fuzz %d f16s videos" % seed)
    same = np.nonzero((gi == want["top_indices"].numpy()).all(1))[0]
    assert len(same) >= nq - max(1, nq // 10)
    gk = moment_keys(out["flat_indices"].cpu().numpy(), gi, index.l_ref)
    wk = moment_keys(want["flat_indices"].numpy(), want["top_indices"].numpy(), index.l_ref)
    n_pos = (out["flat_indices"].cpu().numpy()[same] >= 0).sum(1).min() if len(same) else 0
    tie_aware_equal(gk[same], out["flat_scores"].cpu().numpy()[same], wk[same], want["flat_scores"].numpy()[same],
                    int(min(n_mom, n_pos)), 1e-3, "fuzz %d f16s moments" % seed)
