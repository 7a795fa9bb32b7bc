"""Exact-rank mode (bf16 K6 as a filter, f32 re-score, per-query certificate; include/xmlhip.h "Exact-rank mode"):
kernels against the oracle formulation, and the whole mode against the plain f32 HIP path and the oracle.

The claim under test: on an index built with exact_filter=True, vcmr_search returns the f32 path's lists -- top-k videos
and top-n (video, st, ed) moments identical up to groups of scores tied to f32 rounding -- whatever the bf16 filter did."""
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


def test_round_bf16_rows_err(ops):
    y = _unit_rows(1000, 768, seed=1)
    y[17] = 0
    yb, err = ops.round_bf16_rows_err(y.to(DEV))
    want_b = y.to(torch.bfloat16)
    assert torch.equal(yb.cpu(), want_b)                                   # round-to-nearest-even, bit for bit
    want_e = (y.double() - want_b.double()).norm(dim=-1)
    close("rounding-error norms", err, want_e.float(), 1e-9, 1e-5)
    assert float(err[17]) == 0.0
    # the scale the certificate's bound lives on: ~0.8e-3 for a unit vector of 768 random components
    assert 3e-4 < float(err.mean()) < 2e-3


@pytest.mark.parametrize("n_mod,lpad,hidden,dtype", [(2, 128, 768, torch.float32), (1, 128, 256, torch.float32),
                                                     (2, 64, 128, torch.float32), (2, 128, 256, torch.bfloat16)])
def test_q2c_rescore_vs_reference_formulation(ops, n_mod, lpad, hidden, dtype):
    """xml_q2c_rescore == get_video_level_scores (+ (video + sub) / 2) on the listed pairs, incl. ragged masks, a fully
    masked video, repeated videos (several 64-pair chunks per video) and skipped ids."""
    nq, nv, kp = 300, 37, 9
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(3, lpad + 1, (nv,), generator=g)
    lens[0] = lpad
    mask = (torch.arange(lpad)[None] < lens[:, None]).float()
    mask[5] = 0                                                            # fully masked video: -1e10
    qn = [_unit_rows(nq, hidden, seed=10 + m).to(dtype) for m in range(n_mod)]
    cn = [(_unit_rows(nv, lpad, hidden, seed=20 + m) * mask[..., None]).to(dtype) for m in range(n_mod)]
    pair = torch.randint(0, nv, (nq, kp), generator=g).int()
    pair[:, 0] = 3                                                         # video 3 is in every query's list: 300 pairs = 5 chunks
    pair[7, 2] = -1
    pair[9, 4] = nv + 3
    got = ops.q2c_rescore([q.to(DEV) for q in qn], [c.to(DEV) for c in cn], [mask.to(DEV)] * n_mod, pair.to(DEV)).cpu()
    want = 0
    for m in range(n_mod):      # xml/model_xml.py:448-452
        s = torch.einsum("md,nld->mln", qn[m].float(), cn[m].float())
        s = s * mask.t()[None] + (1 - mask.t()[None]) * -1e10
        want = want + s.max(1)[0]
    want = want / n_mod
    ok = (pair >= 0) & (pair < nv)
    w = torch.gather(want, 1, pair.clamp(0, nv - 1).long())
    tol = 2e-6 if dtype == torch.float32 else 1e-5      # bf16 operands: exact products, f32 accumulation order only
    assert torch.isinf(got[~ok]).all() and (got[~ok] < 0).all()
    close("rescored pairs", got[ok], w[ok], tol, 1e-6)
    # and the same values as the all-pairs K6 kernel on the same operands (what the fallback rows come from)
    full = ops.q2c_scores_fused([q.to(DEV) for q in qn], [c.to(DEV) for c in cn], [mask.to(DEV)] * n_mod).cpu()
    close("rescore vs K6", got[ok], torch.gather(full, 1, pair.clamp(0, nv - 1).long())[ok], tol, 1e-6)


def test_exact_certificate_kernel(ops):
    nq, m, k = 500, 64, 10
    g = torch.Generator().manual_seed(2)
    filt = torch.sort(torch.rand(nq, m, generator=g) * 0.02 + 0.3, dim=1, descending=True)[0].contiguous()
    top = torch.sort(torch.rand(nq, k, generator=g) * 0.02 + 0.3, dim=1, descending=True)[0].contiguous()
    eq = [torch.rand(nq, generator=g) * 1e-3, torch.rand(nq, generator=g) * 1e-3]
    ec, slack, alpha = [0.9e-3, 1.1e-3], 1e-4, 20.0
    c = 1 + 1e-6
    eps = sum(e * c + (c + e) * x for e, x in zip(eq, ec)) / 2 + slack
    want_fail = ~(filt[:, -1] + eps < top[:, -1])
    tv = top.clone().to(DEV)
    fail, eps_g, thr_g, n_fail = ops.exact_certificate(filt.to(DEV), tv, [e.to(DEV) for e in eq], ec, slack, alpha, True)
    margin = (filt[:, -1] + eps - top[:, -1]).abs() > 1e-6            # (f32 vs f64 evaluation of the same inequality)
    assert torch.equal(fail.cpu().bool()[margin], want_fail[margin])
    assert int(n_fail.item()) == int(fail.sum().item()) and 0 < int(n_fail.item()) < nq
    close("eps", eps_g, eps, 1e-8, 1e-5)
    close("second-tier line", thr_g, top[:, -1] - eps, 1e-7, 1e-6)
    close("exp(alpha s)", tv, torch.exp(alpha * top), 0, 1e-6)
    # no videos outside the candidate set: nothing can fail; one modality
    tv = top.clone().to(DEV)
    fail, _, _, n_fail = ops.exact_certificate(filt.to(DEV), tv, [eq[0].to(DEV)], ec[:1], slack, 0.0, False)
    assert int(n_fail.item()) == 0 and not bool(fail.any())
    assert torch.equal(tv.cpu(), top)                                   # alpha == 0: values untouched


def test_select_ge_rows(ops):
    g = torch.Generator().manual_seed(7)
    s_ = torch.rand(37, 5000, generator=g)
    thr = torch.rand(37, generator=g) * 0.1 + 0.9
    thr[3] = 2.0                                                          # nothing reaches it
    cnt = ops.select_ge_rows(s_.to(DEV), thr.to(DEV)).cpu()
    want = (s_ >= thr[:, None]).sum(1)
    assert torch.equal(cnt.long(), want) and int(cnt[3]) == 0
    idx, cnt2 = ops.select_ge_rows(s_.to(DEV), thr.to(DEV), int(want.max()))
    assert torch.equal(cnt2.cpu().long(), want)
    for r in range(37):
        got = sorted(idx[r, :int(want[r])].cpu().tolist())
        assert got == torch.nonzero(s_[r] >= thr[r]).reshape(-1).tolist()
        assert (idx[r, int(want[r]):] == -1).all()


def _lists_equal(out, ref, l, kv, kn, what):
    """out's lists == ref's lists (ref: vcmr_search output of the plain f32 path, with its (Nq, Nv) q2c), tie-aware at f32
    rounding (2e-5 relative on exp(20 s) = 1e-6 on s)."""
    gi = out["top_indices"].cpu().numpy()
    ww, wi = torch.topk(torch.exp(20.0 * ref["q2c"]), min(kv + 8, ref["q2c"].shape[1]), dim=1)   # a few past the boundary
    n_v = tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi.cpu().numpy(), ww.cpu().numpy(), kv, 2e-5,
                          what + " videos")
    ri = ref["top_indices"].cpu().numpy()
    same = np.nonzero((gi == ri).all(1))[0]
    gk = moment_keys(out["flat_indices"].cpu().numpy(), gi, l)
    wk = moment_keys(ref["flat_indices"].cpu().numpy(), ri, l)
    n_m = tie_aware_equal(gk[same], out["flat_scores"].cpu().numpy()[same], wk[same],
                          ref["flat_scores"].cpu().numpy()[same], kn - 8, 5e-5, what + " moments")
    return n_v, n_m, len(same)


@pytest.mark.parametrize("ctx_mode,ragged", [("video_sub", False), ("video_sub", True), ("video", False)])
def test_exact_mode_equals_f32_path(ctx_mode, ragged, monkeypatch):
    """nv > M so the filter really filters: 700 videos, 20 candidates per query for the top-10 videos.  The exact-mode lists
    must be the plain f32 path's lists; with the certificate forced to fail for every query they must be BITWISE the f32
    path's (the fallback IS the f32 path); and the oracle agrees."""
    from tvretrieval_amd import inference as inf
    nq, nv, l, hidden = 64, 700, 128, 128
    m, cfg = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, torch.float32, seed=3)
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
        plain = inf.build_corpus_index(m, batches(), l_ref=l)
        ref = inf.vcmr_search(m, plain, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
        exact = inf.build_corpus_index(m, batches(), l_ref=l, exact_filter=True)
        assert exact.exact is not None and exact.feat1n[exact.modalities[0]].dtype == torch.bfloat16
        exact.exact.n_candidates = 20
        out = inf.vcmr_search(m, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    info = out["exact"]
    n_v, n_m, n_same = _lists_equal(out, ref, l, 10, 200, "exact vs f32")
    assert n_same >= nq - 2
    # the filter was a real bf16 pass over candidates it did not pick trivially
    f32_q2c = ref["q2c"]
    assert info["q2c_filter"].shape == f32_q2c.shape and float((info["q2c_filter"] - f32_q2c).abs().max()) > 1e-6
    assert float((info["q2c_filter"] - f32_q2c).abs().max()) < float(info["eps"].min())      # the bound really bounds
    # re-scored candidates carry the f32 path's scores
    close("re-scored candidates", info["cand_scores"], torch.gather(f32_q2c, 1, info["cand_indices"].long()), 2e-6)
    # certificate honesty: wherever it passed, the f32 top-10 SET is inside the candidate set
    passed = (info["fail"] == 0).cpu().numpy()
    ci, wi = info["cand_indices"].cpu().numpy(), ref["top_indices"].cpu().numpy()
    for q in np.nonzero(passed)[0]:
        assert set(wi[q].tolist()) <= set(ci[q].tolist()), q
    print("exact mode (%s, ragged=%s): %d / %d queries fell back; %d video / %d moment positions swapped in f32 ties"
          % (ctx_mode, ragged, info["n_fail"], nq, n_v, n_m))

    # every query's certificate forced to fail.  Second tier (every video whose filter score reaches T_k - eps is re-scored
    # too; with eps = 10 that is the whole corpus): the f32 path's lists again ...
    exact.exact.e_c = {k: 10.0 for k in exact.exact.e_c}
    with torch.no_grad():
        forced = inf.vcmr_search(m, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    assert forced["exact"]["n_fail"] == nq and forced["exact"]["n_full_rows"] == 0
    _lists_equal(forced, ref, l, 10, 200, "second tier vs f32")
    # ... and with the second tier capped away the fallback is the f32 K6 row itself: BITWISE the f32 path
    monkeypatch.setattr(inf, "EXACT_TIER2_CAP", 0)
    with torch.no_grad():
        forced = inf.vcmr_search(m, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    assert forced["exact"]["n_fail"] == nq and forced["exact"]["n_full_rows"] == nq
    for key in ("top_indices", "top_scores", "flat_indices", "flat_scores"):
        assert torch.equal(forced[key], ref[key]), key

    # and the oracle (reference formulation) on the same inputs
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
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
    tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi2.numpy(), ww.numpy(), 10, 2e-3, "exact vs oracle videos")
    same = np.nonzero((gi == want["top_indices"].numpy()).all(1))[0]
    assert len(same) >= 0.9 * nq
    gk = moment_keys(out["flat_indices"].cpu().numpy(), gi, l)
    wk = moment_keys(want["flat_indices"].numpy(), want["top_indices"].numpy(), l)
    tie_aware_equal(gk[same], out["flat_scores"].cpu().numpy()[same], wk[same], want["flat_scores"].numpy()[same], 200,
                    5e-4, "exact vs oracle moments")


def test_exact_mode_svmr_and_small_corpus():
    """SVMR lists come out of the f32 K7 / K9 in exact-rank mode unchanged, and a corpus smaller than the candidate count
    needs no filter at all (every video is a candidate: no certificate can fail)."""
    from tvretrieval_amd import inference as inf
    nq, nv, l = 9, 40, 64
    m, cfg = _synthetic_model("video_sub", 128, 256, 128, 128, l, torch.float32, seed=6)
    rng = np.random.default_rng(2)
    lens = rng.integers(10, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    qf, qm = _feats(nq, rng.integers(3, 21, nq), 128, 3)
    gt = torch.tensor(rng.integers(0, nv, nq), dtype=torch.int32, device=DEV)
    b = [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))]
    with torch.no_grad():
        ref = inf.vcmr_search(m, inf.build_corpus_index(m, b), qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=50,
                              svmr_video=gt)
        out = inf.vcmr_search(m, inf.build_corpus_index(m, b, exact_filter=True), qf.to(DEV), qm.to(DEV), max_vcmr_video=10,
                              max_before_nms=50, svmr_video=gt)
    assert out["exact"]["n_fail"] == 0 and out["exact"]["n_candidates"] == nv
    for k in ("svmr_scores", "svmr_flat"):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(out["top_indices"], ref["top_indices"]) and torch.equal(out["flat_indices"], ref["flat_indices"])
    close("video weights", out["top_scores"], ref["top_scores"], 0, 2e-5)


def test_exact_mode_rejects_bf16_model():
    from tvretrieval_amd import inference as inf
    m, _ = _synthetic_model("video", 128, 256, 128, 128, 64, torch.bfloat16, seed=3)
    vf, vm = _feats(8, np.full(8, 64), 256, 1)
    with pytest.raises(ValueError), torch.no_grad():
        inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), None, None)], exact_filter=True)
