"""Seeded random-shape parity sweeps of the HIP kernels against the oracle: the fixed cases of test_gpu_kernels.py pin the
documented corner cases, these walk the supported shape space (ragged lengths, odd row counts, tie-heavy scores, skipped
pairs, every clip padding) with a different shape per seed.  Deterministic: the seed is the test id."""
import numpy as np
import pytest
import torch

from oracle import xml_oracle as O
from test_gpu_kernels import (DEV, _att_weights, _check_moment_lists, _conv_case, _conv_oracle, _normed, _ragged_mask,  # noqa: F401
                              _tol, close, dev, ops, rnd)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_topk_rows(ops, seed):
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.integers(1, 13))
    n = int(np.exp(rng.uniform(np.log(2), np.log(6000))))
    k = int(rng.integers(1, min(256, n) + 1))
    s = rnd(rows, n, seed=2000 + seed) * 0.3
    q = float(rng.choice([0.0, 1 / 8, 1 / 64, 1 / 1024]))            # tie density
    if q:
        s = torch.round(s / q) * q
    alpha = float(rng.choice([0.0, 20.0]))
    vals, idx = ops.topk_rows(dev(s), k, alpha=alpha)
    vals, idx = vals.cpu(), idx.cpu().long()
    wv = torch.topk(s, k, dim=1)[0]
    close("topk values", vals, torch.exp(alpha * wv) if alpha else wv, 0, 1e-5 if alpha else 0)
    assert torch.equal(torch.gather(s, 1, idx), wv)
    for r in range(rows):
        key = [(-float(s[r, i]), int(i)) for i in idx[r]]
        assert key == sorted(key) and len(set(idx[r].tolist())) == k
        thr = float(s[r, idx[r, -1]])
        tied = (s[r] == thr).nonzero().flatten().tolist()
        took = sorted(i for i in idx[r].tolist() if float(s[r, i]) == thr)
        assert took == tied[:len(took)], "ties at the threshold go to the lowest columns"


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_moment_topk(ops, seed):
    rng = np.random.default_rng(3000 + seed)
    nq, k = int(rng.integers(1, 5)), int(rng.integers(1, 101))
    l = int(rng.integers(2, 129))
    lpad = (l + 15) // 16 * 16
    min_l = int(rng.integers(0, 4))
    max_l = min_l + int(rng.integers(1, 21))
    n_out = int(rng.integers(1, 301))
    temp = float(rng.choice([0.05, 1.0, 3.0, 8.0]))                  # flat ... peaky span distributions
    g = torch.Generator().manual_seed(4000 + seed)
    mask = (torch.arange(l)[None, None] < torch.randint(1, l + 1, (nq, k, 1), generator=g)).float()
    st = torch.softmax(O.mask_logits(torch.randn(nq, k, l, generator=g) * temp, mask), -1)
    ed = torch.softmax(O.mask_logits(torch.randn(nq, k, l, generator=g) * temp, mask), -1)
    w, _ = torch.sort(torch.exp(20 * (torch.rand(nq, k, generator=g) * 0.3)), dim=1, descending=True)
    if seed % 3 == 0:
        w = torch.where(torch.rand(nq, k, generator=g) < 0.4, torch.zeros_like(w), w)     # pairs owned elsewhere
    prod = torch.einsum("qvm,qv,qvn->qvmn", st, w, ed) * torch.from_numpy(O.min_max_length_mask(l, min_l, max_l))
    ws, wi = torch.sort(prod.reshape(nq, -1), dim=1, descending=True, stable=True)
    n_have = ws.shape[1]
    stp, edp = torch.zeros(nq, k, lpad), torch.zeros(nq, k, lpad)
    stp[..., :l], edp[..., :l] = st, ed
    sc, fl = ops.moment_topk(dev(stp), dev(edp), dev(w.contiguous()), l, min_l, max_l, n_out)
    want_s, want_i = torch.zeros(nq, n_out), torch.full((nq, n_out), -1, dtype=torch.long)
    m = min(n_out, n_have)
    want_s[:, :m], want_i[:, :m] = ws[:, :m], wi[:, :m]
    _check_moment_lists(sc, fl, want_s, want_i, l)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_convse_rerank(ops, seed):
    rng = np.random.default_rng(5000 + seed)
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    nq, nv = int(rng.integers(1, 71)), int(rng.integers(1, 41))
    l = int(rng.integers(5, 129))
    h = int(rng.choice([64, 128, 256, 768]))
    n_mod = int(rng.integers(1, 3))
    merged = bool(n_mod == 2 and rng.integers(0, 2))
    softmax = bool(rng.integers(0, 2))
    lpad = (l + 15) // 16 * 16
    q, f, mask, cw = _conv_case(nq, nv, l, h, merged, n_mod, 6000 + seed)
    k = int(rng.integers(1, min(8, nv) + 1))
    g = torch.Generator().manual_seed(7000 + seed)
    pair = torch.stack([torch.randperm(nv, generator=g)[:k] for _ in range(nq)]).int()
    skip = torch.rand(nq, k, generator=g) < 0.15
    want_st, want_ed = _conv_oracle(q, f, mask, cw, merged, pair, softmax)
    want_st[skip] = 0; want_ed[skip] = 0
    fp = [torch.zeros(nv, lpad, h) for _ in f]
    for a, b in zip(fp, f):
        a[:, :l] = b
    mp = torch.zeros(nv, lpad); mp[:, :l] = mask
    pair_skip = torch.where(skip, torch.full_like(pair, -1), pair).contiguous()
    st, ed = ops.convse_rerank([dev(x, dtype) for x in q], [dev(x, dtype) for x in fp], [dev(mp)] * n_mod, dev(pair_skip),
                               dev(cw), l, merged, 5, softmax=softmax)
    st, ed = st[..., :l].cpu(), ed[..., :l].cpu()
    if softmax:
        close("convse st prob", st, want_st, 1e-5, 1e-4)
        close("convse ed prob", ed, want_ed, 1e-5, 1e-4)
    else:
        close("convse st logits", st, want_st, 1e-4, 1e-5)
        close("convse ed logits", ed, want_ed, 1e-4, 1e-5)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_q2c_fused(ops, seed):
    rng = np.random.default_rng(8000 + seed)
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    nq, nv = int(rng.integers(1, 400)), int(rng.integers(1, 80))
    l = int(rng.integers(1, 129)) if seed % 3 else 128
    h = int(rng.choice([64, 128, 192, 256, 512, 768]))
    n_mod = int(rng.integers(1, 3))
    lpad = (l + 15) // 16 * 16
    qs = [_normed(nq, h, seed=8100 + seed + m) for m in range(n_mod)]
    cs = [_normed(nv, l, h, seed=8200 + seed + m) for m in range(n_mod)]
    masks = [_ragged_mask(nv, l, 8300 + seed + m) for m in range(n_mod)]
    want = None
    for m in range(n_mod):
        s = torch.einsum("md,nld->mln", qs[m], cs[m])
        s = torch.max(O.mask_logits(s, masks[m].t().unsqueeze(0)), dim=1)[0]
        want = s if want is None else (want + s) / 2
    cps, mps = [], []
    for m in range(n_mod):
        cp = torch.zeros(nv, lpad, h); cp[:, :l] = cs[m]
        mp = torch.zeros(nv, lpad); mp[:, :l] = masks[m]
        cps.append(dev(cp, dtype)); mps.append(dev(mp))
    out = torch.full((nq, nv), float("nan"), device=DEV)
    got = ops.q2c_scores_fused([dev(q, dtype) for q in qs], cps, mps, out=out)
    close("q2c fused", got, want, 1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_select_ge_and_certificate(ops, seed):
    rng = np.random.default_rng(9000 + seed)
    rows, n = int(rng.integers(1, 60)), int(rng.integers(1, 3000))
    s = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32))
    thr = torch.from_numpy(rng.standard_normal(rows).astype(np.float32))
    cnt = ops.select_ge_rows(dev(s), dev(thr)).cpu().numpy()
    want_cnt = (s >= thr[:, None]).sum(1).numpy()
    assert np.array_equal(cnt, want_cnt)
    cap = int(max(1, want_cnt.max()))
    idx, cnt2 = ops.select_ge_rows(dev(s), dev(thr), cap)
    idx = idx.cpu().numpy()
    for r in range(rows):
        assert sorted(idx[r][:want_cnt[r]].tolist()) == np.nonzero((s[r] >= thr[r]).numpy())[0].tolist()
        assert (idx[r][want_cnt[r]:] == -1).all()
    # certificate: b_M + eps < T_k  with eps from the two error norms
    nq, m, k = rows, int(rng.integers(2, 40)), 1
    k = int(rng.integers(1, m + 1))
    filt = torch.sort(torch.from_numpy(rng.uniform(0.2, 0.3, (nq, m)).astype(np.float32)), dim=1, descending=True)[0]
    top = torch.sort(torch.from_numpy(rng.uniform(0.29, 0.31, (nq, k)).astype(np.float32)), dim=1, descending=True)[0]
    eq = [torch.from_numpy(rng.uniform(1e-3, 2e-3, nq).astype(np.float32)) for _ in range(2)]
    ec = [2.0e-3, 2.2e-3]
    slack, alpha = 9e-5, 20.0
    tv = dev(top.clone())
    fail, eps, thr_out, n_fail = ops.exact_certificate(dev(filt), tv, [dev(e) for e in eq], ec, slack, alpha, True)
    c = np.float32(1.0 + 1e-6)
    e0 = eq[0].numpy() * c + (c + eq[0].numpy()) * np.float32(ec[0])
    e1 = eq[1].numpy() * c + (c + eq[1].numpy()) * np.float32(ec[1])
    want_eps = (e0 + e1) * np.float32(0.5) + np.float32(slack)
    np.testing.assert_allclose(eps.cpu().numpy(), want_eps, rtol=1e-6)
    want_fail = ~(filt[:, -1].numpy() + want_eps < top[:, -1].numpy())
    got_fail = fail.cpu().numpy().astype(bool)
    edge = np.abs(filt[:, -1].numpy() + want_eps - top[:, -1].numpy()) < 1e-6          # (one-ulp boundary cases may differ)
    assert np.array_equal(got_fail[~edge], want_fail[~edge]) and int(n_fail.item()) == int(got_fail.sum())
    np.testing.assert_allclose(tv.cpu().numpy(), np.exp(alpha * top.numpy()), rtol=2e-6)
    np.testing.assert_allclose(thr_out.cpu().numpy(), top[:, -1].numpy() - want_eps, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_attention_block(ops, seed):
    rng = np.random.default_rng(10000 + seed)
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    n, l = int(rng.integers(1, 24)), int(rng.integers(1, 129))
    h = int(rng.choice([128, 256, 512, 768]))
    nh = 4
    x = rnd(n, l, h, seed=10100 + seed)
    mask = _ragged_mask(n, l, 10200 + seed)
    w = _att_weights(h, 10300 + seed)
    want = O.bert_attention(x, mask.unsqueeze(1), O.Weights(w), nh)
    wqkv = torch.cat([w["self.query.weight"], w["self.key.weight"], w["self.value.weight"]], 0)
    bqkv = torch.cat([w["self.query.bias"], w["self.key.bias"], w["self.value.bias"]], 0)
    got = ops.attention_block(dev(x, dtype), dev(mask), dev(wqkv, dtype), dev(bqkv), dev(w["output.dense.weight"], dtype),
                              dev(w["output.dense.bias"]), dev(w["output.LayerNorm.weight"]), dev(w["output.LayerNorm.bias"]), nh)
    close("attention_block", got, want, _tol(dtype, 2e-4, 8e-2))


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_exact_rank_mode_equals_f32_path(seed):
    """Whole exact-rank searches on random small worlds (context mode, hidden size, clip padding, corpus and candidate
    counts, ragged or full videos): the lists must be the plain f32 path's, tie-aware at f32 rounding."""
    from tvretrieval_amd import inference as inf
    from test_gpu_exact import _lists_equal
    from test_gpu_model import _feats, _synthetic_model
    rng = np.random.default_rng(11000 + seed)
    ctx_mode = str(rng.choice(["video_sub", "video", "sub"]))
    hidden = int(rng.choice([128, 256]))
    l = int(rng.choice([128, 128, 64, 48]))
    nv, nq = int(rng.integers(30, 900)), int(rng.integers(1, 90))
    kv = int(rng.integers(1, min(12, nv) + 1))
    n_cand = int(rng.integers(kv, min(nv, 4 * kv + 8) + 1))
    m, cfg = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, torch.float32, seed=100 + seed)
    lens = rng.integers(4, l + 1, nv) if seed % 2 else np.full(nv, l)
    lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1 + seed)
    sf, sm = _feats(nv, lens, 128, 2 + seed)
    qf, qm = _feats(nq, rng.integers(1, 31, nq), 128, 3 + seed)
    bs = int(rng.integers(20, 200))

    def batches():
        for b in range(0, nv, bs):
            yield (vf[b:b + bs].to(DEV), vm[b:b + bs].to(DEV), sf[b:b + bs].to(DEV), sm[b:b + bs].to(DEV))
    n_mom = 100
    with torch.no_grad():
        plain = inf.build_corpus_index(m, batches(), l_ref=l)
        ref = inf.vcmr_search(m, plain, qf.to(DEV), qm.to(DEV), max_vcmr_video=kv, max_before_nms=n_mom)
        exact = inf.build_corpus_index(m, batches(), l_ref=l, exact_filter=True)
        exact.exact.n_candidates = n_cand
        out = inf.vcmr_search(m, exact, qf.to(DEV), qm.to(DEV), max_vcmr_video=kv, max_before_nms=n_mom)
    n_v, n_m, n_same = _lists_equal(out, ref, l, kv, n_mom, "fuzz %d exact vs f32" % seed)
    assert n_same >= nq - max(1, nq // 20)
    print("seed %d: %s h=%d l=%d nv=%d nq=%d k=%d M=%d: %d fell back, %d / %d tie swaps"
          % (seed, ctx_mode, hidden, l, nv, nq, kv, n_cand, out["exact"]["n_fail"], n_v, n_m))


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_search_vs_oracle_fp32(seed):
    """Whole VCMR searches (encode, K6, top-k, ConvSE, moment top-n; f32) against the reference formulation on random
    configurations: context mode, cross attention on / off, merged or per-stream span predictors, hidden size, clip
    count, ragged lengths, feature widths, list sizes."""
    from tvretrieval_amd import inference as inf
    from oracle.listcmp import moment_keys, tie_aware_equal
    from test_gpu_model import _feats, _synthetic_model
    rng = np.random.default_rng(12000 + seed)
    ctx_mode = str(rng.choice(["video_sub", "video_sub", "video", "sub"]))
    cross, merge = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    hidden = int(rng.choice([128, 256, 384]))
    l = int(rng.choice([128, 100, 64, 40, 17]))
    dv, ds_, dq = int(rng.choice([256, 512, 1024])), int(rng.choice([128, 256])), int(rng.choice([128, 256]))
    nv, nq = int(rng.integers(3, 60)), int(rng.integers(1, 40))
    kv = int(rng.integers(1, min(10, nv) + 1))
    n_mom = int(rng.integers(5, 150))
    m, cfg = _synthetic_model(ctx_mode, hidden, dv, ds_, dq, l, torch.float32, seed=200 + seed, cross=cross, merge=merge)
    lens = rng.integers(max(2, l // 6), l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, dv, 1 + seed)
    sf, sm = _feats(nv, lens, ds_, 2 + seed)
    qf, qm = _feats(nq, rng.integers(1, 31, nq), dq, 3 + seed)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    min_l, max_l = 2, int(rng.choice([8, 16]))
    with torch.no_grad():
        ov1, ov2, os1, os2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm if om.use_video else None, os1, os2,
                                                 sm if om.use_sub else None, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, kv, min_l, max_l, n_mom + 16)
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=kv, max_before_nms=n_mom, min_pred_l=min_l,
                              max_pred_l=max_l)
    close("q2c", out["q2c"], q2c, 1e-4)
    gi, wi = out["top_indices"].cpu().numpy(), want["top_indices"].numpy()
    ww, wi2 = torch.topk(torch.exp(20.0 * q2c), min(kv + 6, nv), dim=1)
    tie_aware_equal(gi, out["top_scores"].cpu().numpy(), wi2.numpy(), ww.numpy(), kv, 4e-3, "fuzz %d videos" % seed)
    same = np.nonzero((gi == wi).all(1))[0]             # same ranked videos => same (k, L, L) candidate tensor
    fs, fi = out["flat_scores"].cpu().numpy(), out["flat_indices"].cpu().numpy()
    gk, wk = moment_keys(fi, gi, l), moment_keys(want["flat_indices"].numpy(), wi, l)
    ws = want["flat_scores"].numpy()
    for q in same:                                      # compare the positive-score prefix (zero-score rows are dropped)
        npos = int((ws[q][:n_mom] > 0).sum())
        got_n = int((fi[q] >= 0).sum())
        assert got_n == npos, (seed, q, got_n, npos)
        if npos > 2:
            tie_aware_equal(gk[q:q + 1, :npos], fs[q:q + 1, :npos], wk[q:q + 1], ws[q:q + 1], max(1, npos - 2), 1e-3,
                            "fuzz %d moments of query %d" % (seed, q))
    print("seed %d: %s cross=%s merge=%s h=%d l=%d nv=%d nq=%d k=%d n=%d: %d / %d queries with the oracle's video order"
          % (seed, ctx_mode, cross, merge, hidden, l, nv, nq, kv, n_mom, len(same), nq))
    assert len(same) >= 0.8 * nq


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_training_forward_backward_vs_oracle_autograd(seed):
    """XML.forward + backward (f32, dropout off) on random configurations against the oracle restatement differentiated
    by torch autograd on the CPU: the three loss terms and every parameter gradient, with the two negative-sample draws
    injected on both sides (as in the golden train-step fixtures)."""
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.train import xml_forward_train
    from test_gpu_model import _feats
    rng = np.random.default_rng(13000 + seed)
    ctx_mode = str(rng.choice(["video_sub", "video_sub", "video", "sub"]))
    cross, merge = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if ctx_mode != "video_sub":
        cross = merge = False
    hidden = int(rng.choice([128, 256]))
    l = int(rng.integers(8, 101))
    dv, ds_, dq = int(rng.choice([256, 512])), int(rng.choice([128, 256])), int(rng.choice([128, 256]))
    bsz = int(rng.integers(3, 25))
    cfg = dict(merge_two_stream=merge, cross_att=cross, span_predictor_type="conv", encoder_type="transformer",
               visual_input_size=dv, sub_input_size=ds_, query_input_size=dq, hidden_size=hidden, conv_kernel_size=5,
               stack_conv_predictor_conv_kernel_sizes=-1, conv_stride=1, max_ctx_l=l, max_desc_l=30, input_drop=0.0, drop=0.0,
               n_heads=4, initializer_range=0.02, ctx_mode=ctx_mode, margin=float(rng.choice([0.1, 0.2])),
               ranking_loss_type=str(rng.choice(["hinge", "lse"])), lw_neg_q=float(rng.choice([1, 0.5])),
               lw_neg_ctx=float(rng.choice([1, 2])), lw_st_ed=float(rng.choice([0.01, 0.1])), use_hard_negative=False,
               hard_pool_size=20, use_self_attention=True, no_modular=False)
    torch.manual_seed(300 + seed)
    m = XML(cfg, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(301 + seed)
    with torch.no_grad():        # richer than N(0, 0.02): scores that differ, losses with active terms
        for n_, p in m.named_parameters():
            if n_.lower().endswith("layernorm.weight"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n_.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "predictor" in n_:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[-1]))
    m = m.to(DEV).train()
    lens = rng.integers(3, l + 1, bsz); lens[0] = l
    vf, vm = _feats(bsz, lens, dv, 1 + seed)
    sf, sm = _feats(bsz, lens, ds_, 2 + seed)
    qf, qm = _feats(bsz, rng.integers(1, 31, bsz), dq, 3 + seed)
    st = np.array([rng.integers(0, n) for n in lens]); ed = np.array([rng.integers(s, n) for s, n in zip(st, lens)])
    st_ed = torch.from_numpy(np.stack([st, ed], 1)).long()
    neg_ctx, neg_q = rng.integers(1, bsz, bsz), rng.integers(1, bsz, bsz)
    # oracle + autograd
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    om = O.OracleXML(cfg, sd)
    want_loss, want_parts = om.forward_loss(qf, qm, vf, vm, sf, sm, st_ed, neg_ctx, neg_q)
    want_loss.backward()
    # HIP
    T = lambda t: t.to(DEV)      # noqa: E731
    loss, parts = xml_forward_train(m, T(qf), T(qm), T(vf), T(vm), T(sf), T(sm), T(st_ed), neg_ctx_rank=neg_ctx,
                                    neg_q_rank=neg_q)
    m.zero_grad()
    loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-4 * max(1.0, abs(float(want_loss))), (float(loss), float(want_loss))
    for k in ("loss_st_ed", "loss_neg_ctx", "loss_neg_q"):
        assert abs(parts[k] - want_parts[k]) <= 1e-4 * max(1.0, abs(want_parts[k])), (k, parts[k], want_parts[k])
    worst = []
    for n_, p in m.named_parameters():
        wg = sd[n_].grad
        if wg is None:                      # parameters of an unused modality
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n_
            continue
        assert p.grad is not None, n_
        err = float((p.grad.cpu() - wg).abs().max())
        worst.append((err / max(float(wg.abs().max()), 1e-4), n_, err))
    worst.sort(reverse=True)
    print("seed %d: %s cross=%s merge=%s %s h=%d l=%d bsz=%d: loss %.5f, worst gradient errors %s"
          % (seed, ctx_mode, cross, merge, cfg["ranking_loss_type"], hidden, l, bsz, float(loss), worst[:2]))
    assert worst[0][0] < 1e-3, [w for w in worst if w[0] >= 1e-3][:40]


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_cross_attention(ops, seed):
    rng = np.random.default_rng(14000 + seed)
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    n, lq, lk = int(rng.integers(1, 16)), int(rng.integers(1, 129)), int(rng.integers(1, 129))
    h = int(rng.choice([128, 256, 768]))
    nh = 4
    main, side = rnd(n, lq, h, seed=14100 + seed), rnd(n, lk, h, seed=14200 + seed)
    mm, sm = _ragged_mask(n, lq, 14300 + seed), _ragged_mask(n, lk, 14400 + seed)
    sd = _att_weights(h, 14500 + seed)
    att = {k[5:]: v for k, v in sd.items() if k.startswith("self.")}
    g, b = sd["output.LayerNorm.weight"], sd["output.LayerNorm.bias"]
    cross = O.bert_self_attention(main, side, side, torch.einsum("bm,bn->bmn", mm, sm), O.Weights(att), nh)
    want = torch.nn.functional.layer_norm(cross + main, (h,), g, b, 1e-5)
    wkv = torch.cat([att["key.weight"], att["value.weight"]], 0)
    bkv = torch.cat([att["key.bias"], att["value.bias"]], 0)
    got = ops.cross_attention(dev(main, dtype), dev(mm), dev(side, dtype), dev(sm), dev(att["query.weight"], dtype),
                              dev(att["query.bias"]), dev(wkv, dtype), dev(bkv), dev(g), dev(b), nh)
    # padded QUERY rows see (score - 10000) on every key: f32 keeps ~1e-3 of the score there, in the reference too, so their
    # softmax -- and with it the row -- carries that much rounding noise; valid rows are compared at the tight tolerance
    valid = mm.bool()
    close("cross_attention, valid query rows", got.float().cpu()[valid], want[valid], _tol(dtype, 2e-4, 8e-2))
    close("cross_attention, padded query rows", got.float().cpu()[~valid], want[~valid], _tol(dtype, 3e-3, 8e-2))


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_packed_query_ops_equal_padded(ops, seed):
    """xml_attention_block_varlen / xml_modular_pool_varlen on packed tokens against the padded entries (and the padded
    modular pooling against the oracle), random sequence counts and lengths <= 32."""
    rng = np.random.default_rng(15000 + seed)
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    n, lq = int(rng.integers(1, 300)), int(rng.integers(1, 33))
    h = int(rng.choice([128, 256, 768]))
    n_mod = int(rng.integers(1, 3))
    lens = rng.integers(1, lq + 1, n); lens[0] = lq
    mask = torch.from_numpy((np.arange(lq)[None, :] < lens[:, None]).astype(np.float32))
    x = rnd(n, lq, h, seed=15100 + seed)
    sd = _att_weights(h, 15200 + seed)
    wqkv = torch.cat([sd["self.query.weight"], sd["self.key.weight"], sd["self.value.weight"]], 0)
    bqkv = torch.cat([sd["self.query.bias"], sd["self.key.bias"], sd["self.value.bias"]], 0)
    wargs = (dev(wqkv, dtype), dev(bqkv), dev(sd["output.dense.weight"], dtype), dev(sd["output.dense.bias"]),
             dev(sd["output.LayerNorm.weight"]), dev(sd["output.LayerNorm.bias"]))
    cu, src, rows = ops.pack_plan(dev(mask))
    xp = dev(x, dtype).reshape(n * lq, h)[src[:rows].long()].contiguous()
    padded = ops.attention_block(dev(x, dtype), dev(mask), *wargs, 4)
    packed = ops.attention_block_varlen(xp, cu, n, lq, *wargs, 4)
    sel = padded.reshape(n * lq, h)[src[:rows].long()]
    close("attention varlen vs padded", packed, sel, 2e-5 if dtype == torch.float32 else 3e-2, 0.0 if dtype == torch.float32 else 1.6e-2)
    wm = rnd(n_mod, h, seed=15300 + seed, scale=h ** -0.5)
    pool_pad = ops.modular_pool(padded, dev(mask), dev(wm))
    pool_pack = ops.modular_pool_varlen(sel.contiguous(), cu, n, lq, dev(wm))
    close("modular pool varlen vs padded", pool_pack, pool_pad, 2e-5 if dtype == torch.float32 else 2e-2)
    pf = padded.float().cpu()
    att = torch.softmax(O.mask_logits(torch.einsum("nld,md->nlm", pf, wm), mask.unsqueeze(2)), dim=1)
    want = torch.einsum("nlm,nld->mnd", att, pf)
    close("modular pool vs the reference formulation", pool_pad, want, _tol(dtype, 2e-5, 2e-2))


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_svmr_vs_oracle_fp32(seed):
    """SVMR (moments inside each query's ground-truth video; xml/inference.py:195-241, 330-340) on random configurations
    against the reference formulation: the softmaxed span probabilities of the given video and its banded top-n."""
    from tvretrieval_amd import inference as inf
    from oracle.listcmp import tie_aware_equal
    from test_gpu_model import _feats, _synthetic_model
    rng = np.random.default_rng(16000 + seed)
    ctx_mode = str(rng.choice(["video_sub", "video", "sub"]))
    cross, merge = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    hidden = int(rng.choice([128, 256]))
    l = int(rng.choice([128, 96, 50, 24]))
    nv, nq = int(rng.integers(2, 40)), int(rng.integers(1, 30))
    n_mom = int(rng.integers(5, 120))
    m, cfg = _synthetic_model(ctx_mode, hidden, 256, 128, 128, l, torch.float32, seed=400 + seed, cross=cross, merge=merge)
    lens = rng.integers(max(3, l // 5), l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1 + seed)
    sf, sm = _feats(nv, lens, 128, 2 + seed)
    qf, qm = _feats(nq, rng.integers(1, 31, nq), 128, 3 + seed)
    gt = torch.from_numpy(rng.integers(0, nv, nq)).int()
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        ov1, ov2, os1, os2 = om.encode_context(vf, vm, sf, sm)
        _, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm if om.use_video else None, os1, os2,
                                               sm if om.use_sub else None, cross=True)
        ar = torch.arange(nq)
        st_p, ed_p = torch.softmax(st, -1)[ar, gt.long()], torch.softmax(ed, -1)[ar, gt.long()]
        want = O.svmr_tail(st_p.numpy(), ed_p.numpy(), 2, 16, n_mom + 8)
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=min(5, nv), max_before_nms=n_mom,
                              svmr_video=gt.to(DEV))
    close("svmr st prob", out["svmr_st"][:, :l], st_p, 1e-6, 2e-3)
    close("svmr ed prob", out["svmr_ed"][:, :l], ed_p, 1e-6, 2e-3)
    gs, gf = out["svmr_scores"].cpu().numpy(), out["svmr_flat"].cpu().numpy()
    for q in range(nq):
        ws, wf = want[q, :, 2], (want[q, :, 0] * l + want[q, :, 1]).astype(np.int64)
        npos = int((ws[:n_mom] > 0).sum())
        assert int((gf[q] >= 0).sum()) == npos, (seed, q)
        if npos > 2:
            tie_aware_equal(gf[q:q + 1, :npos], gs[q:q + 1, :npos], wf[None], ws[None], max(1, npos - 2), 2e-3,
                            "fuzz %d SVMR moments of query %d" % (seed, q))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_k7_candidate_summaries_feed_k9(ops, seed):
    """xml_convse_rerank_ex emits, per pair, 8 banded row maxima (st * w) * max_d ed (one per group of 16 rows) -- checked against a torch
    restatement on the probabilities the kernel itself returned -- and xml_moment_topk_ex started from them returns the SAME
    lists, bit for bit, as the kernel that makes its own first pass: peaky and flat distributions, skipped pairs, weights
    of 0, one pair per query (SVMR), every band / clip padding."""
    rng = np.random.default_rng(31000 + seed)
    n_mod = int(rng.integers(1, 3))
    merged = bool(n_mod == 2 and rng.integers(0, 2))
    lpad = int(rng.choice([16, 48, 64, 112, 128]))
    l_ref = int(rng.integers(max(2, lpad - 15), lpad + 1))
    hidden = int(rng.choice([64, 128, 256]))
    nq, nv = int(rng.integers(1, 60)), int(rng.integers(1, 25))
    kp = 1 if seed % 5 == 0 else int(rng.integers(1, 101))
    min_l = int(rng.integers(0, 4))
    max_l = min_l + int(rng.integers(1, 20))
    n_out = int(rng.integers(1, 300))
    dtype = torch.float32 if seed % 2 else torch.bfloat16
    g = torch.Generator().manual_seed(500 + seed)
    lens = torch.randint(1, l_ref + 1, (nv,), generator=g)
    mask = (torch.arange(lpad)[None] < lens[:, None]).float()
    sharp = float(rng.choice([0.05, 1.0, 6.0]))              # flat ... peaky start / end distributions
    q_lin = [(torch.randn(nq, hidden, generator=g) * sharp).to(dtype) for _ in range(n_mod)]
    feat2 = [(torch.randn(nv, lpad, hidden, generator=g) * mask[..., None]).to(dtype) for _ in range(n_mod)]
    conv_w = torch.randn(2 * (1 if merged else n_mod) * 5, generator=g) * 0.5
    pair = torch.randint(-1 if seed % 3 == 0 else 0, nv, (nq, kp), generator=g).int()
    w = None
    if kp > 1:
        w = torch.exp(20.0 * (0.1 + 0.02 * torch.rand(nq, kp, generator=g))).sort(1, descending=True)[0].contiguous()
        w = torch.where(pair >= 0, w, torch.zeros_like(w))   # skipped pairs carry weight 0 (the sharded pass)
    dev = lambda ts: [t.to(DEV) for t in ts]                 # noqa: E731
    masks = [mask.to(DEV)] * n_mod
    wd = None if w is None else w.to(DEV)
    st, ed, summ = ops.convse_rerank(dev(q_lin), dev(feat2), masks, pair.to(DEV), conv_w.to(DEV), l_ref, merged, 5,
                                     pair_w=wd, band=(min_l, max_l))
    st0, ed0 = ops.convse_rerank(dev(q_lin), dev(feat2), masks, pair.to(DEV), conv_w.to(DEV), l_ref, merged, 5)
    # the summaries change nothing else.  (Not bit for bit any more: without summaries the 5-tap case runs the half-wave
    # epilogue, whose softmax sums its 128 terms in another order -- the last bit of a probability may differ.)
    assert torch.allclose(st, st0, rtol=4e-6, atol=1e-9) and torch.allclose(ed, ed0, rtol=4e-6, atol=1e-9)
    # restatement: m[i] = (st[i] * w) * max_{min_l <= d < max_l, i + d < l_ref} ed[i + d]; top 8 per pair, descending
    s_, e_ = st.cpu()[..., :l_ref], ed.cpu()[..., :l_ref]
    a = s_ * (w[..., None] if w is not None else 1.0)
    best = torch.zeros_like(a)
    for d_ in range(min_l, max_l):
        if d_ < l_ref:
            best[..., :l_ref - d_] = torch.maximum(best[..., :l_ref - d_], e_[..., d_:])
    m = (a * best).clamp_min(0)
    mp = torch.cat([m, torch.zeros(nq, kp, 128 - l_ref)], -1)
    want = torch.maximum(mp[..., :64], mp[..., 64:]).view(nq, kp, 8, 8).amax(-1)      # group g: rows 8 g .. + 7 and + 64
    live = (pair >= 0)
    assert torch.equal(summ.cpu()[live], want[live]), float((summ.cpu()[live] - want[live]).abs().max())
    a1 = ops.moment_topk(st, ed, wd, l_ref, min_l, max_l, n_out, summ=summ)
    a0 = ops.moment_topk(st, ed, wd, l_ref, min_l, max_l, n_out)
    assert torch.equal(a1[0], a0[0]) and torch.equal(a1[1], a0[1]), seed
