"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI, via tvretrieval_amd.ops) against the CPU
oracle on the same seeded inputs.  fp32 storage: 1e-4 absolute on similarity scores / logits (BASELINE.json);
bf16 storage: looser, stated per test.  Run on the MI355X box:  pytest -m gpu"""
import numpy as np
import pytest
import torch

from oracle import xml_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype, f32=1e-4, bf16=3e-2):
    return f32 if dtype == torch.float32 else bf16


def close(name, got, want, atol, rtol=0.0):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    err = np.abs(got - want)
    lim = atol + rtol * np.abs(want)
    bad = err > lim
    assert not bad.any(), "%s: %d/%d off, max err %.3e at %s (got %r want %r)" % (
        name, bad.sum(), bad.size, err.max(), np.unravel_index(err.argmax(), err.shape),
        got.flat[err.argmax()], want.flat[err.argmax()])


def bf16_grid(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_grid(torch.randn(*shape, generator=g) * scale)


@pytest.fixture(scope="module")
def ops():
    from tvretrieval_amd import ops as o
    o._lib.load()
    return o


@pytest.fixture
def dbg_lib(ops, monkeypatch):
    """The -DXML_DEBUG_VARIANTS build (kernel-variant switches) standing in for the product library during one test.
    The product library has no such switches (stateless dispatch); __graft_entry__.build() builds both."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(ops._lib.LIB_PATH), "libxmlhip_dbg.so")
    if not os.path.isfile(path):
        pytest.skip("libxmlhip_dbg.so not built (XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh)")
    lib = ops._lib.bind(ctypes.CDLL(path))
    monkeypatch.setattr(ops._lib, "_lib", lib)
    return lib


def dev(t, dtype=None):
    t = t.to(DEV)
    return t.to(dtype).contiguous() if dtype is not None else t.contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(5, 24, 40), (300, 136, 96), (257, 768, 3072), (128, 128, 64), (1000, 2304, 768),
                                   (513, 200, 128), (256, 128, 64)])
def test_linear(ops, dtype, shape):
    m, n, k = shape
    x, w, b = rnd(m, k, seed=1), rnd(n, k, seed=2, scale=k ** -0.5), rnd(n, seed=3)
    want = torch.nn.functional.linear(x, w, b)
    got = ops.linear(dev(x, dtype), dev(w, dtype), dev(b))
    close("linear", got, want, _tol(dtype, 2e-5, 2e-2), 1e-2 if dtype == torch.bfloat16 else 1e-5)
    got = ops.linear(dev(x, dtype), dev(w, dtype), None, relu=True)
    close("linear relu", got, torch.relu(torch.nn.functional.linear(x, w)), _tol(dtype, 2e-5, 2e-2),
          1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_persistent_gemm_equals_tile_per_workgroup_gemm(ops, dbg_lib, dtype):
    """large projections run on the persistent 256x256 kernel (gemm256p.hip): same MFMA sequence per accumulator as the
    one-tile-per-workgroup kernel -> bitwise the same outputs, for every epilogue (bias, ReLU, position / residual
    addend, bf16 / f32 out, ragged M, N not a multiple of 8)."""
    import ctypes
    lib = dbg_lib
    g = torch.Generator(device=DEV).manual_seed(7)

    def both(fn):
        res = []
        try:
            for variant in (0, 2):
                lib.xml_debug_set_gemm_variant(ctypes.c_int(variant))
                res.append(fn())
        finally:
            lib.xml_debug_set_gemm_variant(ctypes.c_int(0))
        return res

    for m, n, k in ((270001, 768, 256), (90000, 2304, 768), (270040, 700, 128 if dtype == torch.bfloat16 else 64)):
        x = (torch.randn(m, k, device=DEV, generator=g) * 0.3).to(dtype)
        w = (torch.randn(n, k, device=DEV, generator=g) * k ** -0.5).to(dtype)
        b = torch.randn(n, device=DEV, generator=g)
        a, c = both(lambda: ops.linear(x, w, None, relu=True))
        assert torch.equal(a, c), (m, n, k, "relu")
        a, c = both(lambda: ops.linear(x, w, b))
        assert torch.equal(a, c), (m, n, k)
        if m in (270001, 270040):      # spot check against torch on a corner that includes the ragged last row tile
            want = torch.nn.functional.linear(x[-300:].float(), w.float(), b)
            close("persistent gemm", a[-300:], want.cpu(), _tol(dtype, 2e-5, 2e-2), 1e-2 if dtype == torch.bfloat16 else 1e-5)
    # K1+K2 (position-embedding addend, f32 pre-LN output) and a whole BertAttention block (QKV, residual addend)
    n_seq, l, d_in, h = 2200, 128, 256, 768
    x = torch.randn(n_seq, l, d_in, device=DEV, generator=g)
    wts = [torch.randn(h, d_in, device=DEV, generator=g).mul(d_in ** -0.5).to(dtype), torch.randn(h, device=DEV, generator=g) * 0.1,
           torch.randn(l, h, device=DEV, generator=g).mul(0.5).to(dtype)]
    ones, zeros = torch.ones(d_in, device=DEV), torch.zeros(d_in, device=DEV)
    oh, zh = torch.ones(h, device=DEV), torch.zeros(h, device=DEV)
    a, c = both(lambda: ops.linear_ln_relu_pos(x, ones, zeros, wts[0], wts[1], wts[2], oh, zh))
    assert torch.equal(a, c)
    mask = torch.ones(n_seq, l, device=DEV)
    wqkv = torch.randn(3 * h, h, device=DEV, generator=g).mul(h ** -0.5).to(dtype)
    wo = torch.randn(h, h, device=DEV, generator=g).mul(h ** -0.5).to(dtype)
    bqkv, bo = torch.randn(3 * h, device=DEV, generator=g) * 0.1, torch.randn(h, device=DEV, generator=g) * 0.1
    y, z = both(lambda: ops.attention_block(a, mask, wqkv, bqkv, wo, bo, oh, zh, 4))
    assert torch.equal(y, z)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_l2norm_convert(ops, dtype):
    a, b = rnd(37, 200, seed=4), rnd(37, 200, seed=5)
    g, beta = 1 + 0.1 * rnd(200, seed=6), 0.1 * rnd(200, seed=7)
    want = torch.nn.functional.layer_norm(a + b, (200,), g, beta, 1e-5)
    got = ops.add_layernorm(dev(a), dev(b, dtype), dev(g), dev(beta), out_dtype=dtype)
    close("add_layernorm", got, want, _tol(dtype, 1e-5, 2e-2))
    want = torch.nn.functional.normalize(a, dim=-1)
    got = ops.l2norm_rows(dev(a, dtype))
    close("l2norm", got, want, _tol(dtype, 1e-6, 4e-3))
    z = torch.zeros(3, 64)
    close("l2norm zero rows", ops.l2norm_rows(dev(z, dtype)), z, 0)
    close("convert", ops.convert(dev(a), dtype), a.to(dtype).float(), 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 17, 96, 128), (2, 128, 3072, 768), (9, 30, 64, 256),
                                   (3, 40, 770, 256), (2, 100, 3074, 768), (2, 9, 37, 64)])   # TEF dims: d_in % 8 != 0
def test_linear_ln_relu_pos(ops, dtype, shape):
    n, l, d_in, h = shape
    x = rnd(n, l, d_in, seed=10)
    sd = {"LayerNorm.weight": 1 + 0.1 * rnd(d_in, seed=11), "LayerNorm.bias": 0.1 * rnd(d_in, seed=12),
          "net.1.weight": rnd(h, d_in, seed=13, scale=d_in ** -0.5), "net.1.bias": 0.1 * rnd(h, seed=14)}
    pe = {"position_embeddings.weight": rnd(l + 3, h, seed=15, scale=0.5), "LayerNorm.weight": 1 + 0.1 * rnd(h, seed=16),
          "LayerNorm.bias": 0.1 * rnd(h, seed=17)}
    want = O.trainable_pos_enc(O.linear_layer(x, O.Weights(sd)), O.Weights(pe))
    w_pad = torch.nn.functional.pad(sd["net.1.weight"], (0, -d_in % 8))      # pack-time K padding (LinearLayer.packed)
    got = ops.linear_ln_relu_pos(dev(x), dev(sd["LayerNorm.weight"]), dev(sd["LayerNorm.bias"]),
                                 dev(w_pad, dtype), dev(sd["net.1.bias"]),
                                 dev(pe["position_embeddings.weight"], dtype), dev(pe["LayerNorm.weight"]),
                                 dev(pe["LayerNorm.bias"]))
    close("linear_ln_relu_pos", got, want, _tol(dtype, 5e-5, 6e-2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("h,n_seq", [(768, 600), (256, 1600), (512, 800)])
def test_gemm_layernorm_epilogue_matches_unfused(ops, dtype, h, n_seq):
    """Large batches run K1+K2 and BertSelfOutput with the LayerNorm inside the persistent GEMM's epilogue (cross-workgroup
    row statistics, gemm256p.hip LNE); half-size batches of the same rows fall below the eligibility threshold and take
    the GEMM (f32 out) + LayerNorm launches.  Rows are independent, so the two must agree: f32 to rounding of the
    statistics, bf16 to one rounding of the pre-LayerNorm value.  The first sequences are also checked against the oracle."""
    if dtype == torch.float32 and h == 768:
        n_seq = 608
    l, d_in, nh = 128, 128, 4
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(n_seq, l, d_in, device=DEV, generator=g)
    sd = {"LayerNorm.weight": 1 + 0.1 * rnd(d_in, seed=11), "LayerNorm.bias": 0.1 * rnd(d_in, seed=12),
          "net.1.weight": rnd(h, d_in, seed=13, scale=d_in ** -0.5), "net.1.bias": 0.1 * rnd(h, seed=14)}
    pe = {"position_embeddings.weight": rnd(l, h, seed=15, scale=0.5), "LayerNorm.weight": 1 + 0.1 * rnd(h, seed=16),
          "LayerNorm.bias": 0.1 * rnd(h, seed=17)}
    args = (dev(sd["LayerNorm.weight"]), dev(sd["LayerNorm.bias"]), dev(sd["net.1.weight"], dtype), dev(sd["net.1.bias"]),
            dev(pe["position_embeddings.weight"], dtype), dev(pe["LayerNorm.weight"]), dev(pe["LayerNorm.bias"]))
    half = n_seq // 2
    fused = ops.linear_ln_relu_pos(x, *args)
    parts = torch.cat([ops.linear_ln_relu_pos(x[:half].contiguous(), *args), ops.linear_ln_relu_pos(x[half:].contiguous(), *args)])
    tol = 2e-5 if dtype == torch.float32 else 3e-2            # bf16: + 2 ulp relative (one extra rounding before the LN)
    rtol = 0.0 if dtype == torch.float32 else 1.6e-2
    close("K1+K2 fused vs unfused", fused, parts, tol, rtol)
    want = O.trainable_pos_enc(O.linear_layer(x[:3].cpu(), O.Weights(sd)), O.Weights(pe))
    close("K1+K2 fused vs oracle", fused[:3], want, _tol(dtype, 5e-5, 6e-2))
    # BertAttention block on the encoder output
    sdA = _att_weights(h, 40)
    wqkv = torch.cat([sdA["self.%s.weight" % k] for k in ("query", "key", "value")])
    bqkv = torch.cat([sdA["self.%s.bias" % k] for k in ("query", "key", "value")])
    aargs = (dev(wqkv, dtype), dev(bqkv), dev(sdA["output.dense.weight"], dtype), dev(sdA["output.dense.bias"]),
             dev(sdA["output.LayerNorm.weight"]), dev(sdA["output.LayerNorm.bias"]), nh)
    mask = (torch.arange(l, device=DEV)[None] < torch.randint(20, l + 1, (n_seq, 1), device=DEV, generator=g)).float()
    xin = fused
    fusedA = ops.attention_block(xin, mask, *aargs)
    partsA = torch.cat([ops.attention_block(xin[:half].contiguous(), mask[:half].contiguous(), *aargs),
                        ops.attention_block(xin[half:].contiguous(), mask[half:].contiguous(), *aargs)])
    close("BertAttention fused vs unfused", fusedA, partsA, 5e-5 if dtype == torch.float32 else 3e-2, rtol)
    wantA = O.bert_attention(xin[:2].float().cpu(), mask[:2].cpu().unsqueeze(1), O.Weights(sdA), nh)
    close("BertAttention fused vs oracle", fusedA[:2], wantA, _tol(dtype, 2e-4, 8e-2))


def _att_weights(h, seed):
    s = h ** -0.5
    sd = {}
    for i, nm in enumerate(["query", "key", "value"]):
        sd["self.%s.weight" % nm] = rnd(h, h, seed=seed + i, scale=s)
        sd["self.%s.bias" % nm] = 0.1 * rnd(h, seed=seed + 10 + i)
    sd["output.dense.weight"] = rnd(h, h, seed=seed + 20, scale=s)
    sd["output.dense.bias"] = 0.1 * rnd(h, seed=seed + 21)
    sd["output.LayerNorm.weight"] = 1 + 0.1 * rnd(h, seed=seed + 22)
    sd["output.LayerNorm.bias"] = 0.1 * rnd(h, seed=seed + 23)
    return sd


def _ragged_mask(n, l, seed, full_first=True):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, l + 1, (n,), generator=g)
    if full_first:
        lens[0] = l
    return (torch.arange(l)[None] < lens[:, None]).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(5, 40, 128, 4), (3, 128, 768, 4), (4, 100, 256, 4), (6, 13, 128, 4), (2, 128, 512, 4)])
def test_attention_block(ops, dtype, shape):
    n, l, h, nh = shape
    if dtype == torch.float32 and h == 768 and l > 100:
        pass  # 135 KB of LDS: still within the 160 KB CU budget
    x = rnd(n, l, h, seed=30)
    mask = _ragged_mask(n, l, 31)
    sd = _att_weights(h, 40)
    want = O.bert_attention(x, mask.unsqueeze(1), O.Weights(sd), nh)
    wqkv = torch.cat([sd["self.query.weight"], sd["self.key.weight"], sd["self.value.weight"]], 0)
    bqkv = torch.cat([sd["self.query.bias"], sd["self.key.bias"], sd["self.value.bias"]], 0)
    got = ops.attention_block(dev(x, dtype), dev(mask), dev(wqkv, dtype), dev(bqkv), dev(sd["output.dense.weight"], dtype),
                              dev(sd["output.dense.bias"]), dev(sd["output.LayerNorm.weight"]),
                              dev(sd["output.LayerNorm.bias"]), nh)
    close("attention_block (incl. padded rows)", got, want, _tol(dtype, 1e-4, 8e-2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4, 33, 33, 128, 4), (2, 128, 128, 768, 4), (3, 20, 20, 256, 4)])
def test_cross_attention(ops, dtype, shape):
    n, lq, lk, h, nh = shape
    main, side = rnd(n, lq, h, seed=50), rnd(n, lk, h, seed=51)
    mm, sm = _ragged_mask(n, lq, 52), _ragged_mask(n, lk, 53)
    sd = _att_weights(h, 60)
    att = {k[5:]: v for k, v in sd.items() if k.startswith("self.")}
    g, b = sd["output.LayerNorm.weight"], sd["output.LayerNorm.bias"]
    cross_mask = torch.einsum("bm,bn->bmn", mm, sm)
    cross = O.bert_self_attention(main, side, side, cross_mask, O.Weights(att), nh)
    want = torch.nn.functional.layer_norm(cross + main, (h,), g, b, 1e-5)
    wkv = torch.cat([att["key.weight"], att["value.weight"]], 0)
    bkv = torch.cat([att["key.bias"], att["value.bias"]], 0)
    got = ops.cross_attention(dev(main, dtype), dev(mm), dev(side, dtype), dev(sm), dev(att["query.weight"], dtype),
                              dev(att["query.bias"]), dev(wkv, dtype), dev(bkv), dev(g), dev(b), nh)
    # padded query rows see (score - 10000): fp32 absorbs ~1e-3 of the score there (same in the reference)
    close("cross_attention (incl. padded rows)", got, want, _tol(dtype, 2e-4, 8e-2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_mod", [1, 2])
def test_modular_pool(ops, dtype, n_mod):
    n, lq, h = 11, 30, 256
    enc = rnd(n, lq, h, seed=70)
    mask = _ragged_mask(n, lq, 71)
    wm = rnd(n_mod, h, seed=72, scale=h ** -0.5)
    sc = torch.softmax(O.mask_logits(enc @ wm.t(), mask.unsqueeze(2)), dim=1)
    want = torch.einsum("blm,bld->mbd", sc, enc)
    got = ops.modular_pool(dev(enc, dtype), dev(mask), dev(wm))
    close("modular_pool", got, want, _tol(dtype, 1e-5, 2e-2))


def _normed(*shape, seed):
    return bf16_grid(torch.nn.functional.normalize(rnd(*shape, seed=seed), dim=-1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(7, 10, 40, 128), (130, 9, 128, 768), (300, 37, 100, 256), (1, 1, 128, 768),
                                   (256, 33, 16, 128)])
def test_q2c_scores(ops, dtype, shape):
    nq, nv, l, h = shape
    lpad = (l + 15) // 16 * 16
    q, c = _normed(nq, h, seed=80), _normed(nv, l, h, seed=81)
    mask = _ragged_mask(nv, l, 82)
    s = torch.einsum("md,nld->mln", q, c)
    want = torch.max(O.mask_logits(s, mask.t().unsqueeze(0)), dim=1)[0]
    cpad = torch.zeros(nv, lpad, h); cpad[:, :l] = c
    mpad = torch.zeros(nv, lpad); mpad[:, :l] = mask
    got = ops.q2c_scores(dev(q, dtype), dev(cpad, dtype), dev(mpad))
    close("q2c", got, want, _tol(dtype, 1e-5, 1e-5))   # inputs are bf16-exact, accumulation is f32 in both modes
    # second modality averaged in place: (a + b) / 2
    got2 = ops.q2c_scores(dev(q, dtype), dev(cpad, dtype), dev(mpad), out=got.clone(), combine=True)
    close("q2c combine", got2, (want + want) / 2, 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(700, 37, 128, 768), (300, 2, 128, 256), (1, 1, 128, 128), (520, 1030, 128, 128),
                                   (33, 9, 100, 256), (40, 12, 48, 128)])
@pytest.mark.parametrize("n_mod", [1, 2])
def test_q2c_fused(ops, dtype, shape, n_mod):
    """the engine's entry point: all modalities in one launch (persistent kernel at lpad == 128) vs the oracle."""
    nq, nv, l, h = shape
    lpad = (l + 15) // 16 * 16
    qs = [_normed(nq, h, seed=180 + m) for m in range(n_mod)]
    cs = [_normed(nv, l, h, seed=190 + m) for m in range(n_mod)]
    masks = [_ragged_mask(nv, l, 200 + m) for m in range(n_mod)]
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


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(700, 37, 768), (300, 2, 256), (1, 1, 128), (520, 1030, 192), (2600, 65, 768)])
@pytest.mark.parametrize("n_mod", [1, 2])
def test_q2c_tiled_equals_row_major(ops, dtype, shape, n_mod):
    """slice-major operand tiles (the resident index layout) vs row-major operands: the same kernel with a different
    source addressing -> bitwise the same scores; tile -> rows round trip is the identity."""
    nq, nv, h = shape
    if h * (2 if dtype == torch.bfloat16 else 4) < 384:
        pytest.skip("below the persistent kernel's minimum K")
    qs = [dev(_normed(nq, h, seed=280 + m), dtype) for m in range(n_mod)]
    cs = [dev(_normed(nv, 128, h, seed=290 + m), dtype) for m in range(n_mod)]
    masks = [dev(_ragged_mask(nv, 128, 300 + m)) for m in range(n_mod)]
    assert ops.q2c_tiled_ok(128, h, dtype)
    want = ops.q2c_scores_fused(qs, cs, masks)
    tiles = [ops.pack_q2c_corpus(c) for c in cs]
    assert all(isinstance(t, ops.TiledRows) for t in tiles)
    for t, c in zip(tiles, cs):
        assert torch.equal(t.to_rows(), c)
        assert t.numel() == (nv * 128 + 255) // 256 * 256 * h
    out = torch.full((nq, nv), float("nan"), device=DEV)
    got = ops.q2c_scores_fused(qs, tiles, masks, out=out)
    assert torch.equal(got, want)
    # full-length videos: the tiles are marked all_valid, K6 skips the masks and runs its five-slot ring
    ones = [torch.ones_like(m) for m in masks]
    want1 = ops.q2c_scores_fused(qs, cs, ones)
    tiles1 = [ops.pack_q2c_corpus(c, o) for c, o in zip(cs, ones)]
    assert all(t.all_valid for t in tiles1)
    assert all(ops.pack_q2c_corpus(c, m).all_valid == bool((m == 1).all()) for c, m in zip(cs, masks))
    out1 = torch.full((nq, nv), float("nan"), device=DEV)
    assert torch.equal(ops.q2c_scores_fused(qs, tiles1, ones, out=out1), want1)
    # the same corpus through the packed-bit masks (mode 2, all bits set) and through the float masks (mode 0)
    for t in tiles1:
        t.all_valid = False
        t.mask_bits = torch.full((nv, 4), -1, dtype=torch.int32, device=DEV)
    assert torch.equal(ops.q2c_scores_fused(qs, tiles1, ones), want1)
    for t in tiles1:
        t.mask_bits = None
    assert torch.equal(ops.q2c_scores_fused(qs, tiles1, ones), want1)
    # non-binary masks cannot be packed: float path, still equal to the row-major kernel
    soft = [m * 0.5 + 0.5 * (m > 0) * (torch.arange(128, device=DEV)[None] % 2 == 0) for m in masks]
    tsoft = [ops.pack_q2c_corpus(c, m) for c, m in zip(cs, soft)]
    if not all(bool(((m == 0) | (m == 1)).all()) for m in soft):
        assert all(t.mask_bits is None and not t.all_valid for t, m in zip(tsoft, soft) if not bool(((m == 0) | (m == 1)).all()))
    assert torch.equal(ops.q2c_scores_fused(qs, tsoft, soft), ops.q2c_scores_fused(qs, cs, soft))
    # padding rows of the last tile are zero (an odd number of videos leaves half a tile)
    flat = tiles[0].data.view(-1, h // (64 // tiles[0].element_size()), 256, 64 // tiles[0].element_size())
    if (nv * 128) % 256:
        assert float(flat[-1, :, 128:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_normalise_and_tile_in_one_pass_is_bitwise_l2norm_then_tile(ops, dtype):
    """index build: xml_q2c_tile_rows_l2norm (F.normalize inside the tiling pass, plain and through the length-bucket row map)
    == xml_l2norm_rows followed by the tiling kernels, bit for bit; the normalised rows agree with F.normalize."""
    nv, h = 301, 768
    g = torch.Generator().manual_seed(77)
    lens = torch.randint(1, 129, (nv,), generator=g)
    mask = dev((torch.arange(128)[None] < lens[:, None]).float())
    x = dev(torch.randn(nv, 128, h, generator=g) * 3.0, dtype) * mask[..., None].to(dtype)
    n1 = ops.l2norm_rows(x)
    close("l2norm_rows", n1, torch.nn.functional.normalize(x.float().cpu(), dim=-1), _tol(dtype, 2e-6, 8e-3), 1e-2 if dtype == torch.bfloat16 else 1e-6)
    ones = torch.ones_like(mask)
    for m, plan in ((ones, None), (mask, None), (mask, ops.q2c_pack_plan([mask]))):
        a = ops.pack_q2c_corpus(x, m, plan, normalize=True)
        b = ops.pack_q2c_corpus(n1, m, plan)
        assert torch.equal(a.data, b.data)
        assert torch.equal(a.to_rows(), n1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_mod", [1, 2])
def test_q2c_length_buckets_bitwise(ops, dtype, n_mod):
    """Ragged corpus (TVR-like lengths, mean ~51 of 128 clips): the packed image (videos padded to 16 clips, back to back,
    straddling the wave tiles; xml_q2c_scores_packed) gives BITWISE the scores of the plain tiled image with packed bit
    masks, in the videos' original columns -- so every list downstream is identical -- and both equal the oracle formulation."""
    nq, nv, h = 700, 611, 256
    g = torch.Generator().manual_seed(31)
    lens = torch.cat([torch.randint(1, 33, (40,), generator=g), torch.randint(33, 65, (500,), generator=g),
                      torch.randint(65, 129, (71,), generator=g)])[torch.randperm(nv, generator=g)]
    qs = [_normed(nq, h, seed=280 + m) for m in range(n_mod)]
    cs = [_normed(nv, 128, h, seed=290 + m) for m in range(n_mod)]
    masks = [(torch.arange(128)[None] < (lens - (m if m else 0)).clamp_min(1)[:, None]).float() for m in range(n_mod)]
    cs = [c * mk[..., None] for c, mk in zip(cs, masks)]              # zero rows beyond each video's length
    qd = [dev(q, dtype) for q in qs]
    cd = [dev(c, dtype) for c in cs]
    md = [dev(m) for m in masks]
    plain = [ops.pack_q2c_corpus(c, m) for c, m in zip(cd, md)]
    assert all(t.mask_bits is not None and t.plan is None for t in plain)
    plan = ops.q2c_pack_plan(md)
    assert plan is not None and plan.n_straddles > 0 and plan.n_tiles * 256 < 0.6 * nv * 128
    packed = [ops.pack_q2c_corpus(c, m, plan) for c, m in zip(cd, md)]
    assert all(torch.equal(t.to_rows(), c) for t, c in zip(packed, cd))         # un-bucketing gives the rows back
    want = ops.q2c_scores_fused(qd, plain, md)
    got = ops.q2c_scores_fused(qd, packed, md, out=torch.full((nq, nv), float("nan"), device=DEV))
    assert torch.equal(got, want)
    ref = None
    for m in range(n_mod):
        s = torch.einsum("md,nld->mln", qs[m], cs[m])
        s = torch.max(O.mask_logits(s, masks[m].t().unsqueeze(0)), dim=1)[0]
        ref = s if ref is None else (ref + s) / 2
    close("bucketed q2c vs oracle", got, ref, _tol(dtype, 1e-5, 1e-2))
    print("padded clips %d of %d (%.0f %% of the MFMA work of the unbucketed layout)"
          % (plan.padded_clips, nv * 128, 100.0 * plan.n_tiles * 256 / (nv * 128)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["tiny_videos", "holes_and_empties", "all_sizes", "few_queries"])
def test_q2c_packed_shapes_bitwise(ops, dtype, case):
    """xml_q2c_scores_packed on the layouts the plan can produce: eight one-block videos per wave tile, videos with no
    valid clip / with holes in their masks / whose second modality is shorter or empty, every size 1..8 blocks with
    straddles, fewer queries than a query tile.  Bitwise the plain tiled kernel (mask_logits then max, both modalities)."""
    g = torch.Generator().manual_seed({"tiny_videos": 1, "holes_and_empties": 2, "all_sizes": 3, "few_queries": 4}[case])
    nq, h = (37 if case == "few_queries" else 300), 256
    if case == "tiny_videos":
        lens = torch.randint(1, 17, (203,), generator=g)
    elif case == "holes_and_empties":
        lens = torch.randint(0, 100, (150,), generator=g)
        lens[::7] = 0
    else:
        lens = torch.cat([torch.arange(1, 129), torch.randint(1, 129, (85,), generator=g)])[torch.randperm(213, generator=g)]
    nv = lens.numel()
    masks = [(torch.arange(128)[None] < lens[:, None]).float(), (torch.arange(128)[None] < (lens - 3).clamp_min(0)[:, None]).float()]
    if case == "holes_and_empties":
        masks[0][3, 2:9] = 0
        masks[1][5] = 0
        masks[0][8, :40] = 0
    qs = [_normed(nq, h, seed=380 + m) for m in range(2)]
    cs = [_normed(nv, 128, h, seed=390 + m) * masks[m][..., None] for m in range(2)]
    qd, cd, md = [dev(q, dtype) for q in qs], [dev(c, dtype) for c in cs], [dev(m) for m in masks]
    plain = [ops.pack_q2c_corpus(c, m) for c, m in zip(cd, md)]
    plan = ops.PackPlan(md)
    packed = [ops.pack_q2c_corpus(c, m, plan) for c, m in zip(cd, md)]
    for n_mod in (1, 2):
        want = ops.q2c_scores_fused(qd[:n_mod], plain[:n_mod], md[:n_mod])
        got = ops.q2c_scores_fused(qd[:n_mod], packed[:n_mod], md[:n_mod], out=torch.full((nq, nv), float("nan"), device=DEV))
        assert torch.equal(got, want), (case, n_mod, int((got != want).sum()))
    if case == "holes_and_empties":
        assert bool((got[:, 0] == -1e10).all())                    # no valid clip in either modality: mask_logits' constant


def test_q2c_fused_full_scale_property(ops):
    """BASELINE-size property check (10 000 x 21 793 x 128 x 768 bf16, both modalities): every output element is
    written, and random (query, video) samples equal an fp32 recomputation; max over clips of a cosine of
    unit vectors lies in [-1, 1]."""
    nq, nv, h = 10000, 21793, 768
    g = torch.Generator(device=DEV).manual_seed(5)
    qs, cs = [], []
    for m in range(2):
        qs.append(torch.nn.functional.normalize(torch.randn(nq, h, device=DEV, generator=g), dim=-1).to(torch.bfloat16))
        c = torch.empty(nv, 128, h, device=DEV, dtype=torch.bfloat16)
        for b in range(0, nv, 4096):
            e = min(nv, b + 4096)
            c[b:e] = torch.nn.functional.normalize(torch.randn(e - b, 128, h, device=DEV, generator=g), dim=-1).to(torch.bfloat16)
        cs.append(c)
    lens = torch.randint(1, 129, (nv,), device=DEV, generator=g)
    mask = (torch.arange(128, device=DEV)[None] < lens[:, None]).float().contiguous()
    out = torch.full((nq, nv), float("nan"), device=DEV)
    ops.q2c_scores_fused(qs, [ops.pack_q2c_corpus(c) for c in cs], [mask, mask], out=out)   # the index's tiled layout
    assert not torch.isnan(out).any()
    assert float(out.max()) <= 1.0 + 1e-3 and float(out.min()) >= -1.0 - 1e-3
    qi = torch.randint(0, nq, (64,), device=DEV, generator=g)
    vi = torch.randint(0, nv, (96,), device=DEV, generator=g)
    qi[0], qi[1], vi[0], vi[1] = 0, nq - 1, 0, nv - 1
    want = 0
    for m in range(2):
        s = torch.einsum("qd,vld->qvl", qs[m][qi].float(), cs[m][vi].float())
        s = s.masked_fill(mask[vi][None] == 0, -1e10).max(-1)[0]
        want = want + s
    close("sampled entries", out[qi][:, vi], want / 2, 2e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("lpad", [128, 48])
def test_q2c_variants_equivalent(ops, dbg_lib, dtype, lpad):
    """128x128 register-staged kernel (with / without XCD swizzle) == 256x256 LDS-DMA kernels (ring / double buffer), bit for bit:
    every accumulator sees the same MFMA sequence over K."""
    import ctypes
    lib = dbg_lib
    q, c = _normed(1100, 256, seed=83), _normed(70, lpad, 256, seed=84)
    mask = _ragged_mask(70, lpad, 85)
    args = (dev(q, dtype), dev(c, dtype), dev(mask))
    res = []
    try:
        for variant, swz in ((4, 1), (3, 1), (2, 1), (1, 1), (1, 0)):
            lib.xml_debug_set_q2c_variant(ctypes.c_int(variant))
            lib.xml_debug_set_q2c_swizzle(ctypes.c_int(swz))
            res.append(ops.q2c_scores(*args))
            res.append(ops.q2c_scores(*args, out=res[-1].clone(), combine=True))
    finally:
        lib.xml_debug_set_q2c_variant(ctypes.c_int(0))
        lib.xml_debug_set_q2c_swizzle(ctypes.c_int(1))
    for r in res[2::2]:
        assert torch.equal(res[0], r)
    for r in res[3::2]:
        assert torch.equal(res[1], r)


@pytest.mark.parametrize("shape", [(5, 10, 4), (9, 2179, 100), (3, 21793, 100), (4, 300, 256), (2, 7, 7)])
def test_topk_rows(ops, shape):
    rows, n, k = shape
    s = rnd(rows, n, seed=90) * 0.3
    s[0, : min(n, 50)] = 0.125          # heavy ties at the top
    if rows > 1:
        s[1] = torch.round(s[1] * 8) / 8    # ties everywhere, threshold inside a tie group
    vals, idx = ops.topk_rows(dev(s), k, alpha=20.0)
    vals, idx = vals.cpu(), idx.cpu().long()
    wv, wi = torch.topk(torch.exp(20.0 * s), k, dim=1)
    close("topk values", vals, wv, 0, 1e-5)
    assert torch.equal(torch.gather(s, 1, idx), torch.gather(s, 1, wi)), "selected scores differ"
    for r in range(rows):       # (score desc, index asc) order; no duplicates
        key = [(-float(s[r, i]), int(i)) for i in idx[r]]
        assert key == sorted(key) and len(set(idx[r].tolist())) == k
        thr = float(s[r, idx[r, -1]])  # ties at the threshold resolved to the lowest columns
        tied = (s[r] == thr).nonzero().flatten().tolist()
        took = sorted(i for i in idx[r].tolist() if float(s[r, i]) == thr)
        assert took == tied[:len(took)]
    # payload + raw values
    pay = torch.randperm(n, generator=torch.Generator().manual_seed(1)).int().repeat(rows, 1)
    v2, i2 = ops.topk_rows(dev(s), k, alpha=0.0, idx_in=dev(pay))
    close("topk raw", v2, torch.topk(s, k, dim=1)[0], 0)
    assert torch.equal(torch.gather(s, 1, torch.argsort(pay.long(), dim=1).gather(1, i2.cpu().long())),
                       torch.topk(s, k, dim=1)[0])


@pytest.mark.parametrize("n,k", [(4096, 100), (5000, 256), (21793, 100), (21793, 200), (50000, 37)])
def test_topk_long_rows_prefilter_and_fallbacks(ops, n, k):
    """long rows are pre-filtered through a sample threshold (one pass over the row); rows that defeat the sample --
    sorted either way, constant, mostly -inf, ties across the threshold -- must fall back and still give the exact
    top-k with the (score desc, column asc) tie rule."""
    g = torch.Generator().manual_seed(n + k)
    base = (torch.randn(n, 16, generator=g) * 0.036).max(-1)[0]           # K6-like scores
    rows = [base,
            torch.sort(base)[0],                                          # ascending: the sample holds the smallest
            torch.sort(base, descending=True)[0],                         # descending: the sample holds the largest
            torch.full((n,), 0.25),                                       # constant row
            torch.where(torch.rand(n, generator=g) < 0.999, torch.tensor(float("-inf")), base),   # < k finite values?
            torch.round(base * 64) / 64,                                  # few distinct values: ties across the threshold
            -base]                                                        # negative scores
    rows[4][:k] = base[:k]                                                # keep at least k finite entries
    s = torch.stack(rows)
    vals, idx = ops.topk_rows(dev(s), k, alpha=0.0)
    vals, idx = vals.cpu(), idx.cpu().long()
    wv = torch.topk(s, k, dim=1)[0]
    assert torch.equal(vals, wv)
    assert torch.equal(torch.gather(s, 1, idx), wv)
    for r in range(s.shape[0]):
        key = [(-float(s[r, i]), int(i)) for i in idx[r]]
        assert key == sorted(key) and len(set(idx[r].tolist())) == k, r
        thr = float(s[r, idx[r, -1]])
        tied = (s[r] == thr).nonzero().flatten().tolist()
        took = sorted(i for i in idx[r].tolist() if float(s[r, i]) == thr)
        assert took == tied[:len(took)], r
    pay = torch.randperm(n, generator=g).int().repeat(s.shape[0], 1)
    v2, i2 = ops.topk_rows(dev(s), k, alpha=20.0, idx_in=dev(pay))
    assert torch.allclose(v2.cpu(), torch.exp(20.0 * wv), rtol=1e-6)
    # payloads name the right columns (ties are ordered by payload here, so compare the scores they point at)
    assert torch.equal(torch.gather(s, 1, torch.argsort(pay.long(), dim=1).gather(1, i2.cpu().long())), wv)


def _conv_case(nq, nv, l, h, merged, n_mod, seed):
    g = torch.Generator().manual_seed(seed)
    q = [rnd(nq, h, seed=seed + 1 + m, scale=0.3) for m in range(n_mod)]
    f = [rnd(nv, l, h, seed=seed + 5 + m, scale=0.3) for m in range(n_mod)]
    mask = _ragged_mask(nv, l, seed + 9)
    n_conv = 1 if merged else n_mod
    cw = rnd(2 * n_conv * 5, seed=seed + 11, scale=0.5)
    return q, f, mask, cw


def _conv_oracle(q, f, mask, cw, merged, pair_vid, softmax):
    n_mod = len(q)
    n_conv = 1 if merged else n_mod
    wst, wed = cw[:n_conv * 5].view(n_conv, 1, 1, 5), cw[n_conv * 5:].view(n_conv, 1, 1, 5)
    sims = [torch.einsum("md,nld->mnl", q[m], f[m]) for m in range(n_mod)]
    nq, nv, l = sims[0].shape
    conv = lambda x, w: torch.nn.functional.conv1d(x.reshape(nq * nv, 1, l), w, padding=2).view(nq, nv, l)
    if merged:
        x = (sims[0] + sims[1]) / 2
        st, ed = O.mask_logits(conv(x, wst[0]), mask), O.mask_logits(conv(x, wed[0]), mask)
    else:
        st = sum(O.mask_logits(conv(sims[m], wst[m]), mask) for m in range(n_mod)) / n_mod
        ed = sum(O.mask_logits(conv(sims[m], wed[m]), mask) for m in range(n_mod)) / n_mod
    if softmax:
        st, ed = torch.softmax(st, -1), torch.softmax(ed, -1)
    rows = torch.arange(nq).unsqueeze(1)
    return st[rows, pair_vid.long()], ed[rows, pair_vid.long()]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(7, 10, 40, 128, True, 2), (40, 6, 128, 768, True, 2), (9, 12, 100, 256, False, 1),
                                  (5, 8, 24, 128, False, 2)])
@pytest.mark.parametrize("softmax", [False, True])
def test_convse_rerank(ops, dtype, case, softmax):
    nq, nv, l, h, merged, n_mod = case
    lpad = (l + 15) // 16 * 16
    q, f, mask, cw = _conv_case(nq, nv, l, h, merged, n_mod, 100)
    g = torch.Generator().manual_seed(7)
    k = min(5, nv)
    pair = torch.stack([torch.randperm(nv, generator=g)[:k] for _ in range(nq)]).int()
    pair[0, :] = pair[0, 0]       # many queries on one video + repeated pairs
    pair[:, 0] = 1                 # one video selected by every query (chunks of 32 when nq > 32)
    want_st, want_ed = _conv_oracle(q, f, mask, cw, merged, pair, softmax)
    fp = [torch.zeros(nv, lpad, h) for _ in f]
    for a, b in zip(fp, f):
        a[:, :l] = b
    mp = torch.zeros(nv, lpad); mp[:, :l] = mask
    pair_skip = pair.clone()
    pair_skip[-1, -1] = -1
    st, ed = ops.convse_rerank([dev(x, dtype) for x in q], [dev(x, dtype) for x in fp], [dev(mp)] * n_mod,
                               dev(pair_skip), dev(cw), l, merged, 5, softmax=softmax)
    assert float(st[-1, -1].abs().max()) == 0 and float(ed[-1, -1].abs().max()) == 0, "skipped pair not zeroed"
    st, ed = st[..., :l].cpu(), ed[..., :l].cpu()
    want_st[-1, -1] = 0; want_ed[-1, -1] = 0
    if softmax:
        close("convse st prob", st, want_st, _tol(dtype, 1e-5, 1e-5), 1e-4)
        close("convse ed prob", ed, want_ed, _tol(dtype, 1e-5, 1e-5), 1e-4)
    else:
        close("convse st logits", st, want_st, 1e-4, 1e-5)   # inputs bf16-exact, f32 accumulation in both modes
        close("convse ed logits", ed, want_ed, 1e-4, 1e-5)


def test_convse_unzeroed_skipped_rows_never_reach_k9(ops):
    """Sharded pass: K7 leaves the rows of pairs owned by another rank unwritten (zero_skipped=False) and K9 skips
    pairs of weight 0 -- poisoning those rows must not change the moments."""
    nq, nv, l, h = 12, 9, 64, 128
    q, f, mask, cw = _conv_case(nq, nv, l, h, True, 2, 100)
    g = torch.Generator().manual_seed(11)
    pair = torch.stack([torch.randperm(nv, generator=g)[:6] for _ in range(nq)]).int()
    own = torch.rand(nq, 6, generator=g) < 0.4
    own[:, 0] = True
    pair_local = torch.where(own, pair, torch.full_like(pair, -1))
    args = ([dev(x, torch.bfloat16) for x in q], [dev(x, torch.bfloat16) for x in f], [dev(mask)] * 2, dev(pair_local),
            dev(cw), l, True, 5)
    st0, ed0 = ops.convse_rerank(*args, softmax=True, zero_skipped=True)
    st1, ed1 = ops.convse_rerank(*args, softmax=True, zero_skipped=False)
    o = dev(own)
    assert torch.equal(st0[o], st1[o]) and torch.equal(ed0[o], ed1[o])
    st1[~o] = float("nan"); ed1[~o] = float("nan")
    w = torch.where(o, dev(torch.rand(nq, 6, generator=g) + 0.5), torch.zeros(nq, 6, device=DEV)).contiguous()
    a = ops.moment_topk(st0, ed0, w, l, 2, 16, 60)
    b = ops.moment_topk(st1.contiguous(), ed1.contiguous(), w, l, 2, 16, 60)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def _check_moment_lists(sc, fl, want_s, want_i, l_ref):
    """scores equal to 1e-6 relative; indices equal except inside groups of (near-)tied scores."""
    sc, fl = sc.cpu().numpy(), fl.cpu().numpy()
    want_s, want_i = want_s.numpy(), want_i.numpy()
    for q in range(len(sc)):
        npos = int((want_s[q] > 0).sum())
        np.testing.assert_allclose(sc[q][:npos], want_s[q][:npos], rtol=2e-6, atol=0)
        assert (sc[q][npos:] == 0).all() and (fl[q][npos:] == -1).all()
        mism = np.nonzero(fl[q][:npos] != want_i[q][:npos])[0]
        for i in mism:   # allowed only where the neighbouring scores are within rounding of each other
            j = int(np.nonzero(want_i[q][:npos] == fl[q][i])[0][0]) if fl[q][i] in want_i[q][:npos] else None
            assert j is not None and abs(want_s[q][j] - want_s[q][i]) <= 4e-6 * want_s[q][i], (q, i, j)


@pytest.mark.parametrize("case", [(6, 5, 40, 60), (4, 100, 128, 200), (3, 1, 100, 200), (2, 3, 7, 50), (3, 100, 100, 200)])
def test_moment_topk(ops, case):
    nq, k, l, n_out = case
    lpad = (l + 15) // 16 * 16
    g = torch.Generator().manual_seed(120)
    mask = (torch.arange(l)[None, None] < torch.randint(3, l + 1, (nq, k, 1), generator=g)).float()
    st = torch.softmax(O.mask_logits(torch.randn(nq, k, l, generator=g) * 3, mask), -1)
    ed = torch.softmax(O.mask_logits(torch.randn(nq, k, l, generator=g) * 3, mask), -1)
    w = torch.exp(20 * (torch.rand(nq, k, generator=g) * 0.3))
    w, _ = torch.sort(w, dim=1, descending=True)
    prod = torch.einsum("qvm,qv,qvn->qvmn", st, w, ed) * torch.from_numpy(O.min_max_length_mask(l, 2, 16))
    ws, wi = torch.sort(prod.reshape(nq, -1), dim=1, descending=True)
    stp, edp = torch.zeros(nq, k, lpad), torch.zeros(nq, k, lpad)
    stp[..., :l], edp[..., :l] = st, ed
    sc, fl = ops.moment_topk(dev(stp), dev(edp), dev(w), l, 2, 16, n_out)
    _check_moment_lists(sc, fl, ws[:, :n_out], wi[:, :n_out], l)


def test_moment_topk_skipped_pairs_and_near_flat(ops):
    """w == 0 marks pairs owned by another shard (never candidates); near-flat probabilities with slowly decaying
    video weights make row maxima useless as a bound (the leading-pair bound + overflow refinement must work)."""
    nq, k, l, n_out = 5, 100, 128, 200
    g = torch.Generator().manual_seed(121)
    st = torch.softmax(torch.randn(nq, k, l, generator=g) * 0.05, -1)
    ed = torch.softmax(torch.randn(nq, k, l, generator=g) * 0.05, -1)
    w, _ = torch.sort(torch.exp(20 * (0.2 + 0.01 * torch.rand(nq, k, generator=g))), dim=1, descending=True)
    w[:, 1::3] = 0.0            # every third slot lives on another rank
    w[0, 0] = 0.0
    prod = torch.einsum("qvm,qv,qvn->qvmn", st, w, ed) * torch.from_numpy(O.min_max_length_mask(l, 2, 16))
    ws, wi = torch.sort(prod.reshape(nq, -1), dim=1, descending=True)
    sc, fl = ops.moment_topk(dev(st), dev(ed), dev(w), l, 2, 16, n_out)
    _check_moment_lists(sc, fl, ws[:, :n_out], wi[:, :n_out], l)


def test_moment_topk_flat_distribution_fallback(ops):
    """uniform probabilities: every candidate ties -> list overflow -> exact radix-select fallback."""
    nq, k, l, n_out = 2, 100, 128, 200
    st = torch.full((nq, k, l), 1.0 / l)
    ed = torch.full((nq, k, l), 1.0 / l)
    w = torch.ones(nq, k)
    w[:, 0] = 2.0
    sc, fl = ops.moment_topk(dev(st), dev(ed), dev(w), l, 2, 16, n_out)
    sc, fl = sc.cpu(), fl.cpu()
    assert torch.allclose(sc, torch.full_like(sc, 2.0 / l / l))
    assert ((fl >= 0) & (fl < l * l)).all(), "all winners must come from the boosted video 0"
    for q in range(nq):
        assert len(set(fl[q].tolist())) == n_out


@pytest.mark.parametrize("nq", [2, 300])
def test_moment_topk_more_ties_than_the_list_holds(ops, nq):
    """179 200 candidates per query tie at the n_out-th best score (uniform probabilities, equal weights) and 154 lie above it
    (pair 0 boosted, 11 start clips): the 154 come first, in order, then exact ties -- distinct, valid, whichever the list held
    (both workgroup shapes: 2 queries -> 1024 threads, 300 -> 256)."""
    k, l, n_out = 100, 128, 200
    st = torch.full((nq, k, l), 1.0 / l)
    ed = torch.full((nq, k, l), 1.0 / l)
    st[:, 0, 11:] = 0.0
    w = torch.ones(nq, k)
    w[:, 0] = 2.0
    sc, fl = ops.moment_topk(dev(st), dev(ed), dev(w), l, 2, 16, n_out)
    sc, fl = sc.cpu(), fl.cpu().long()
    n_top = 11 * 14
    want_top = torch.tensor([i * l + j for i in range(11) for j in range(i + 2, i + 16)])
    for q in range(nq):
        assert torch.equal(fl[q, :n_top], want_top), q
        assert torch.allclose(sc[q, :n_top], torch.full((n_top,), 2.0 / l / l))
        rest = fl[q, n_top:]
        assert (rest >= 0).all() and len(set(rest.tolist())) == n_out - n_top and not (set(rest.tolist()) & set(want_top.tolist()))
        r, i, j = rest // (l * l), (rest // l) % l, rest % l
        assert ((j - i >= 2) & (j - i < 16) & ((r > 0) | (i >= 11))).all()
        assert torch.allclose(sc[q, n_top:], torch.full((n_out - n_top,), 1.0 / l / l))


@pytest.mark.parametrize("nq,k,n_out", [(50, 100, 200), (100, 100, 200), (7, 33, 1000), (64, 16, 200), (128, 128, 1024),
                                        (3, 100, 40)])
def test_moment_topk_small_batches_bitwise(ops, nq, k, n_out):
    """Batches too small to fill the chip (the reference's eval_query_bsz = 50) run K9 with 1024-thread workgroups (sixteen waves
    walk a query's pairs); larger ones with 256.  The lists of the SAME rows inside a 200-query batch and alone: bit for bit
    -- peaky and near-flat distributions, skipped pairs (w = 0), ragged valid lengths, lists shorter and (n_out = 1000 of few
    rows) much longer than the ranked-in-place epilogue takes (the bitonic path), and against the oracle's order."""
    l = 128
    g = torch.Generator().manual_seed(300 + nq + k)
    lens = torch.randint(5, l + 1, (37,), generator=g).int()
    pv = torch.randint(0, 37, (200, k), generator=g).int()
    mask = (torch.arange(l)[None, None] < lens[pv.long()][..., None]).float()
    temp = torch.where(torch.arange(200) % 2 == 0, 3.0, 0.05)[:, None, None]
    st = torch.softmax(O.mask_logits(torch.randn(200, k, l, generator=g) * temp, mask), -1) * mask
    ed = torch.softmax(O.mask_logits(torch.randn(200, k, l, generator=g) * temp, mask), -1) * mask
    w, _ = torch.sort(torch.exp(20 * (torch.rand(200, k, generator=g) * 0.3)), dim=1, descending=True)
    w[:, 2::5] = 0.0
    big = ops.moment_topk(dev(st), dev(ed), dev(w), l, 2, 16, n_out, pair_vid=dev(pv), vid_len=dev(lens))
    small = ops.moment_topk(dev(st[:nq].contiguous()), dev(ed[:nq].contiguous()), dev(w[:nq].contiguous()), l, 2, 16, n_out,
                            pair_vid=dev(pv[:nq].contiguous()), vid_len=dev(lens))
    assert torch.equal(small[0], big[0][:nq]) and torch.equal(small[1], big[1][:nq])
    assert int((small[1] >= 0).sum()) > nq * min(50, n_out // 4)
    # the order itself: scores descending, equal scores by ascending flat index; every entry is the product it names
    sc, fl = small[0].cpu(), small[1].cpu().long()
    ok = fl >= 0
    assert (sc[:, :-1] >= sc[:, 1:]).all() and ((sc[:, :-1] > sc[:, 1:]) | (fl[:, :-1] < fl[:, 1:]) | ~ok[:, 1:]).all()
    r, i, j = fl // (l * l), (fl // l) % l, fl % l
    qi = torch.arange(nq)[:, None].expand_as(fl)
    prod = (st[:nq][qi, r.clamp(min=0), i.clamp(min=0)] * w[:nq][qi, r.clamp(min=0)]) * ed[:nq][qi, r.clamp(min=0), j.clamp(min=0)]
    assert torch.equal(torch.where(ok, prod, torch.zeros(())), sc)


@pytest.mark.parametrize("seed", range(6))
def test_moment_topk_forms_agree_on_random_shapes(ops, seed):
    """K9's two workgroup forms (256 threads per query in a 200-query batch, 1 024 threads for the same rows alone) on random
    shapes: l_ref below the padded row length, every band, with / without pair weights, with / without ragged lengths --
    bit for bit, and every entry is the product it names, in (score desc, flat asc) order."""
    rng = np.random.default_rng(900 + seed)
    lpad = int(rng.choice([32, 64, 112, 128]))
    l = int(rng.integers(max(9, lpad - 15), lpad + 1))
    k = int(rng.integers(32, 129))
    nq = int(rng.integers(1, 129))
    min_l = int(rng.integers(0, 4))
    max_l = int(rng.integers(min_l + 1, min(l, 33)))
    n_out = int(rng.choice([1, 37, 200, 513, 1024]))
    g = torch.Generator().manual_seed(901 + seed)
    lens = torch.randint(1, l + 1, (29,), generator=g).int()
    pv = torch.randint(0, 29, (200, k), generator=g).int()
    ragged = bool(seed % 2)
    lim = lens[pv.long()] if ragged else torch.full((200, k), l)
    mask = (torch.arange(lpad)[None, None] < lim[..., None]).float()
    temp = torch.where(torch.arange(200) % 3 == 0, 4.0, 0.1)[:, None, None]
    st = torch.softmax(O.mask_logits(torch.randn(200, k, lpad, generator=g) * temp, mask), -1) * mask
    ed = torch.softmax(O.mask_logits(torch.randn(200, k, lpad, generator=g) * temp, mask), -1) * mask
    w = torch.exp(20 * torch.rand(200, k, generator=g) * 0.2) if seed % 3 else None
    if w is not None:
        w[:, 1::7] = 0.0
    rk = dict(pair_vid=dev(pv), vid_len=dev(lens)) if ragged else {}
    big = ops.moment_topk(dev(st), dev(ed), None if w is None else dev(w), l, min_l, max_l, n_out, **rk)
    rk = dict(pair_vid=dev(pv[:nq].contiguous()), vid_len=dev(lens)) if ragged else {}
    small = ops.moment_topk(dev(st[:nq].contiguous()), dev(ed[:nq].contiguous()), None if w is None else dev(w[:nq].contiguous()),
                            l, min_l, max_l, n_out, **rk)
    assert torch.equal(small[0], big[0][:nq]) and torch.equal(small[1], big[1][:nq]), (lpad, l, k, nq, min_l, max_l, n_out)
    sc, fl = small[0].cpu(), small[1].cpu().long()
    ok = fl >= 0
    assert (sc[:, :-1] >= sc[:, 1:]).all() and ((sc[:, :-1] > sc[:, 1:]) | (fl[:, :-1] < fl[:, 1:]) | ~ok[:, 1:]).all()
    r, i, j = (fl // (l * l)).clamp(min=0), ((fl // l) % l).clamp(min=0), (fl % l).clamp(min=0)
    qi = torch.arange(nq)[:, None].expand_as(fl)
    a = st[:nq][qi, r, i] * (w[:nq][qi, r] if w is not None else 1.0)
    assert torch.equal(torch.where(ok, a * ed[:nq][qi, r, j], torch.zeros(())), sc)
    assert (~ok | ((j - i >= min_l) & (j - i < max_l))).all()
    # nothing better was left out: the smallest listed score bounds every unlisted candidate of the band
    d = torch.arange(lpad)[None, :] - torch.arange(lpad)[:, None]
    band = ((d >= min_l) & (d < max_l))[:l, :l]
    for q in range(min(nq, 4)):
        prod = (st[q, :, :l, None] * (w[q, :, None, None] if w is not None else 1.0)) * ed[q, :, None, :l] * band[None]
        n_pos = int((prod > 0).sum())
        assert int(ok[q].sum()) == min(n_out, n_pos), (q, int(ok[q].sum()), n_pos)
        if n_pos > n_out:
            assert float(torch.topk(prod.flatten(), n_out)[0][-1]) == float(sc[q, n_out - 1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nq", [1, 50, 256, 257, 1000])
def test_query_vectors_of_both_modalities_tiled_in_one_launch(ops, dtype, nq):
    """ops.q2c_tile_rows_l2norm_pair (the halves of the pooled (2, nq, H) tensor, one launch through a row map) = the two
    single-modality launches, bit for bit; layouts that are not two halves of one tensor are refused."""
    g = torch.Generator().manual_seed(40 + nq)
    both = dev(torch.randn(2, nq, 256, generator=g), dtype)
    pair = ops.q2c_tile_rows_l2norm_pair(both[0], both[1])
    assert pair is not None
    for m in range(2):
        one = ops.q2c_tile_rows_l2norm(both[m].contiguous())
        assert torch.equal(pair[m].data, one.data) and pair[m].rows == one.rows
    assert ops.q2c_tile_rows_l2norm_pair(both[1], both[0]) is None
    assert ops.q2c_tile_rows_l2norm_pair(both[0], both[0].clone()) is None


@pytest.mark.parametrize("n,lq", [(1, 30), (3, 5), (1025, 64), (4097, 17), (10000, 30)])
def test_pack_plan_shapes(ops, n, lq):
    """xml_pack_plan (packing plan of a padded token batch): cu_seqlens / source rows for batches smaller and larger than
    its one-workgroup scan, the widest supported sequence (64 = one wave), and the masks it must refuse."""
    rng = np.random.default_rng(n * 131 + lq)
    lens = rng.integers(1, lq + 1, n)
    lens[0] = lq
    mask = (np.arange(lq)[None, :] < lens[:, None]).astype(np.float32)
    cu, src, rows = ops.pack_plan(torch.from_numpy(mask).to(DEV))
    assert rows == int(lens.sum())
    assert np.array_equal(cu.cpu().numpy(), np.concatenate([[0], np.cumsum(lens)]))
    want = np.concatenate([i * lq + np.arange(k) for i, k in enumerate(lens)])
    assert np.array_equal(src[:rows].cpu().numpy(), want)
    bad = mask.copy(); bad[n // 2, :] = 0                                   # an empty sequence
    assert ops.pack_plan(torch.from_numpy(bad).to(DEV))[2] == -1
    if lq > 2:
        bad = mask.copy(); bad[n - 1, 0] = 0; bad[n - 1, 1] = 1             # ones that are not a prefix
        assert ops.pack_plan(torch.from_numpy(bad).to(DEV))[2] == -1
        bad = mask.copy(); bad[0, 1] = 0.5                                  # a value that is neither 0 nor 1
        assert ops.pack_plan(torch.from_numpy(bad).to(DEV))[2] == -1


def test_convse_pair_inversion_corner_cases(ops):
    """K7's pair inversion (count with the atomic's return as the pair's rank, one-workgroup scan, scatter): one video
    selected by every query (a bucket longer than a chunk), videos nobody selected, skipped pairs, and more videos than the
    scan has threads -- the rows must equal the plain one-pair-at-a-time evaluation."""
    h, l, nq, nv, k = 128, 48, 150, 1500, 7
    g = torch.Generator(device=DEV).manual_seed(5)
    q = [torch.randn(nq, h, device=DEV, generator=g) for _ in range(2)]
    f2 = [torch.randn(nv, l, h, device=DEV, generator=g) for _ in range(2)]
    lens = torch.randint(5, l + 1, (nv,), generator=torch.Generator().manual_seed(1))
    mask = (torch.arange(l)[None, :] < lens[:, None]).float().to(DEV)
    pv = torch.randint(0, nv, (nq, k), generator=torch.Generator().manual_seed(2)).int()
    pv[:, 0] = 77                       # every query selects video 77: 150 pairs = three chunks
    pv[::3, 1] = -1                     # skipped pairs
    pv[pv == 5] = 6                     # nobody selects video 5
    pv = pv.to(DEV)
    conv_w = torch.randn(10, device=DEV, generator=g) * 0.3
    st, ed = ops.convse_rerank(q, f2, [mask, mask], pv, conv_w, l, True, 5, softmax=True)
    # reference: each query alone against its own pair list (buckets of one pair: no inversion effects)
    for qi in (0, 1, 3, 149):
        st1, ed1 = ops.convse_rerank([t[qi:qi + 1] for t in q], f2, [mask, mask], pv[qi:qi + 1].contiguous(), conv_w, l, True, 5,
                                     softmax=True)
        assert torch.equal(st[qi], st1[0]) and torch.equal(ed[qi], ed1[0])
    assert float(st[::3, 1].abs().max()) == 0.0 and float(ed[::3, 1].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lq,d_in,h", [(40, 30, 128, 128), (9000, 30, 768, 768)])
def test_linear_ln_relu_pos_packed_vs_oracle(ops, dtype, n, lq, d_in, h):
    """K1+K2 on packed tokens (xml_pack_plan + xml_linear_ln_relu_pos_packed): token t of sequence i is row i * lq + t of
    the PADDED batch read through the source-row map, with positional row t -- the oracle's LinearLayer +
    TrainablePositionalEncoding at the valid positions (small batch: the three-launch path; large: LayerNorm in the GEMM)."""
    rng = np.random.default_rng(n)
    lens = rng.integers(1, lq + 1, n)
    lens[0] = lq
    x = rnd(n, lq, d_in, seed=20)
    mask = torch.from_numpy((np.arange(lq)[None, :] < lens[:, None]).astype(np.float32))
    sd = {"LayerNorm.weight": 1 + 0.1 * rnd(d_in, seed=11), "LayerNorm.bias": 0.1 * rnd(d_in, seed=12),
          "net.1.weight": rnd(h, d_in, seed=13, scale=d_in ** -0.5), "net.1.bias": 0.1 * rnd(h, seed=14)}
    pe = {"position_embeddings.weight": rnd(lq + 2, h, seed=15, scale=0.5), "LayerNorm.weight": 1 + 0.1 * rnd(h, seed=16),
          "LayerNorm.bias": 0.1 * rnd(h, seed=17)}
    cu, src, rows = ops.pack_plan(dev(mask))
    assert rows == int(lens.sum())
    got = ops.linear_ln_relu_pos_packed(dev(x).reshape(n * lq, d_in), src, rows, lq, dev(sd["LayerNorm.weight"]),
                                        dev(sd["LayerNorm.bias"]), dev(sd["net.1.weight"], dtype), dev(sd["net.1.bias"]),
                                        dev(pe["position_embeddings.weight"], dtype), dev(pe["LayerNorm.weight"]),
                                        dev(pe["LayerNorm.bias"]))
    ns = min(n, 64)                                            # the oracle on the first sequences
    want = O.trainable_pos_enc(O.linear_layer(x[:ns], O.Weights(sd)), O.Weights(pe))
    want = torch.cat([want[i, :lens[i]] for i in range(ns)])
    close("linear_ln_relu_pos_packed", got[:want.shape[0]], want, _tol(dtype, 5e-5, 6e-2))
    # and the padded entry on the same batch, every valid token
    full = ops.linear_ln_relu_pos(dev(x), dev(sd["LayerNorm.weight"]), dev(sd["LayerNorm.bias"]), dev(sd["net.1.weight"], dtype),
                                  dev(sd["net.1.bias"]), dev(pe["position_embeddings.weight"], dtype),
                                  dev(pe["LayerNorm.weight"]), dev(pe["LayerNorm.bias"]))
    sel = full.reshape(n * lq, h)[src[:rows].long()]
    close("packed vs padded K1+K2", got, sel, 2e-5 if dtype == torch.float32 else 3e-2, 0.0 if dtype == torch.float32 else 1.6e-2)


@pytest.mark.parametrize("case", [(7, 50, 5, 32, 1.5), (33, 200, 100, 128, 1.5), (5, 1000, 3, 100, 0.7),
                                  (300, 17, 100, 64, 2.0 / 3.0)])
def test_moments_decode_vs_oracle(ops, case):
    """K10 (xml_moments_decode) against the oracle's restatement of the reference's numpy tail (np.unravel_index, the local
    rank -> meta index gather, float32 seconds; xml/inference.py:423-431): video ids and spans bit for bit -- also for clip
    lengths whose float32 products round (the kernel must not contract mul + add into an fma) -- empty tails marked and
    counted, the VR form, the SVMR form in clip units, and writing into a slice of a larger result buffer."""
    from tvretrieval_amd.results import MOMENT_DTYPE
    nq, n, k, l, clip = case
    g = torch.Generator().manual_seed(nq * 1000 + n)
    flat = torch.randint(0, k * l * l, (nq, n), generator=g).int()
    cnt = torch.randint(0, n + 1, (nq,), generator=g)
    cnt[0], cnt[-1] = n, 0
    flat[torch.arange(n)[None, :] >= cnt[:, None]] = -1
    score = torch.rand(nq, n, generator=g)
    nv = 500
    top = torch.stack([torch.randperm(nv, generator=g)[:k] for _ in range(nq)]).int()
    meta2vid = torch.randperm(100000, generator=g)[:nv].int()
    big = torch.full((nq + 5, n + 3, 4), -7, dtype=torch.int32, device=DEV)
    bigc = torch.full((nq + 5,), -7, dtype=torch.int32, device=DEV)
    rec, c = ops.moments_decode(score.to(DEV), flat=flat.to(DEV), top_idx=top.to(DEV), meta2vid=meta2vid.to(DEV), l_ref=l,
                                clip_length=clip, seconds=True, out=big[2:2 + nq, :n], out_count=bigc[2:2 + nq])
    torch.cuda.synchronize()
    assert (big[:2] == -7).all() and (big[2 + nq:] == -7).all() and (big[:, n:] == -7).all() and (bigc[:2] == -7).all()
    h = np.ascontiguousarray(rec.cpu().numpy()).view(MOMENT_DTYPE)[..., 0]
    np.testing.assert_array_equal(c.cpu().numpy(), cnt.numpy())
    valid = (flat >= 0).numpy()
    vid, st_s, ed_s = O.unravel_moments(np.where(valid, flat.numpy(), 0), top.numpy(), l, clip_length=clip)
    assert st_s.dtype == np.float32 and ed_s.dtype == np.float32
    np.testing.assert_array_equal(h["vid"], np.where(valid, meta2vid.numpy()[vid], -1))
    np.testing.assert_array_equal(h["st"], np.where(valid, st_s, 0))
    np.testing.assert_array_equal(h["ed"], np.where(valid, ed_s, 0))
    np.testing.assert_array_equal(h["score"], np.where(valid, score.numpy(), 0))
    # SVMR form: one video per query, clip units (st_idx, ed_idx + 1); identity meta map
    fl1 = torch.where(flat >= 0, flat % (l * l), flat)
    rv = torch.randint(0, nv, (nq,), generator=g).int()
    rec, c = ops.moments_decode(score.to(DEV), flat=fl1.to(DEV), row_vid=rv.to(DEV), l_ref=l, seconds=False)
    h = rec.cpu().numpy().view(MOMENT_DTYPE)[..., 0]
    si, ei = np.where(valid, fl1.numpy() // l, 0), np.where(valid, fl1.numpy() % l, 0)
    np.testing.assert_array_equal(h["vid"], np.where(valid, rv.numpy()[:, None], -1))
    np.testing.assert_array_equal(h["st"], np.where(valid, si, 0).astype(np.float32))
    np.testing.assert_array_equal(h["ed"], np.where(valid, ei + 1, 0).astype(np.float32))
    # VR form: [video_idx, 0, 0, score] for the first n_vr columns of the top-k lists
    n_vr = min(k, 100)
    w = torch.rand(nq, k, generator=g)
    rec, c = ops.moments_decode(w.to(DEV), top_idx=top.to(DEV), meta2vid=meta2vid.to(DEV), n=n_vr)
    h = rec.cpu().numpy().view(MOMENT_DTYPE)[..., 0]
    assert h.shape == (nq, n_vr) and (c.cpu().numpy() == n_vr).all()
    np.testing.assert_array_equal(h["vid"], meta2vid.numpy()[top.numpy()[:, :n_vr]])
    np.testing.assert_array_equal(h["score"], w.numpy()[:, :n_vr])
    assert (h["st"] == 0).all() and (h["ed"] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(40, 9, 100, 256, True, 2, 100, 60), (300, 6, 128, 128, True, 2, 5, 40),     # 250 pairs per video: the sub-counter inversion
                                 
                                  (9, 12, 64, 256, False, 1, 12, 30), (13, 7, 40, 128, False, 2, 7, 200)])
def test_ragged_rows_k7_k9_skip_the_zero_tails(ops, dtype, case):
    """Ragged corpora (xml_convse_rerank_ex / xml_moment_topk_ex with vid_len): K7 neither fetches the clip rows nor writes
    the probabilities beyond a video's valid length -- they are exact zeros of the masked softmax -- and K9 does not read
    them.  The destination is pre-filled with NaN: every entry l < vid_len[v] must be BITWISE the full kernel's, every
    entry beyond must still be NaN (unwritten), and K9 on the poisoned rows must return the full path's lists bit for bit
    -- also with skipped pairs (-1), empty videos, non-prefix masks and the 5-tap filter reaching across the valid end."""
    nq, nv, l, h, merged, n_mod, k, n_out = case
    lpad = (l + 15) // 16 * 16
    q, f, mask, cw = _conv_case(nq, nv, l, h, merged, n_mod, 300 + nq)
    mask = mask.clone()
    mask[0] = 1                                           # a full-length video
    mask[1] = 0; mask[1, :3] = 1; mask[1, 1] = 0          # a hole inside the valid range (not a prefix)
    if nv > 4:
        mask[4] = 0                                       # an empty video (uniform probabilities: nothing may be skipped)
    g = torch.Generator().manual_seed(17)
    k = min(k, nv)
    pair = torch.stack([torch.randperm(nv, generator=g)[:k] for _ in range(nq)]).int()
    pair[-1, -1] = -1
    fp = [torch.zeros(nv, lpad, h) for _ in f]
    for a, b in zip(fp, f):
        a[:, :l] = b                                      # (padded clips keep their encoder outputs: the taps read them)
    mp = torch.zeros(nv, lpad); mp[:, :l] = mask
    vlen = ((mp != 0).int() * torch.arange(1, lpad + 1).int()).amax(1).clamp(max=l).int()
    vlen[vlen == 0] = l          # a video without a valid clip: its masked softmax is UNIFORM, not zero -> full rows
    args = ([dev(x, dtype) for x in q], [dev(x, dtype) for x in fp], [dev(mp)] * n_mod, dev(pair), dev(cw), l, merged, 5)
    st_f, ed_f = ops.convse_rerank(*args, softmax=True, zero_skipped=False)
    nan = lambda: torch.full((nq, k, lpad), float("nan"), device=DEV)                     # noqa: E731
    st_r, ed_r = ops.convse_rerank(*args, softmax=True, zero_skipped=False, vid_len=dev(vlen), out=(nan(), nan()))
    pv = pair.long().clamp(min=0)
    live = (torch.arange(lpad)[None, None, :] < vlen[pv][..., None]) & (pair >= 0)[..., None]
    # (whole 16-byte pieces are stored: up to 3 of the exact zeros behind the valid length may be written too)
    maybe = (torch.arange(lpad)[None, None, :] < ((vlen[pv] + 3) // 4 * 4)[..., None]) & (pair >= 0)[..., None]
    for name, full, rag in (("st", st_f, st_r), ("ed", ed_f, ed_r)):
        full, rag = full.cpu(), rag.cpu()
        assert torch.equal(full[live], rag[live]), name + ": stored entries differ from the full kernel's"
        assert torch.isnan(rag[~maybe]).all(), name + ": an entry beyond the valid length was written"
        between = maybe & ~live
        assert ((rag[between] == 0) | torch.isnan(rag[between])).all(), name + ": non-zero behind the valid length"
        assert float(full[(~live) & (pair >= 0)[..., None]].abs().max()) == 0.0, name + ": the skipped tail is not exactly zero"
    w = torch.rand(nq, k, generator=g) + 0.1
    w[-1, -1] = 0.0                                       # (a skipped pair carries weight 0 in the sharded pass)
    st_f[-1, -1] = 0; ed_f[-1, -1] = 0
    want = ops.moment_topk(st_f, ed_f, dev(w), l, 2, 16, n_out)
    got = ops.moment_topk(st_r, ed_r, dev(w), l, 2, 16, n_out, pair_vid=dev(pair), vid_len=dev(vlen))
    assert torch.equal(got[1], want[1]) and torch.equal(got[0], want[0]), "K9 on ragged rows != K9 on full rows"


@pytest.mark.parametrize("shape", [(9, 1600, 200), (5, 300, 100), (3, 5000, 256), (4, 64, 17)])
def test_topk_threshold_ties_go_to_the_lowest_payloads(ops, shape):
    """With a payload (xml_topk_rows idx_in: the merge of per-shard lists) the order is (score desc, payload asc) -- also AT
    the list's last position: when more elements tie with the k-th score than fit, the ones with the smallest payloads are
    taken, wherever they sit in the row.  (A merged sharded list must not depend on which rank a tied candidate came from.)"""
    rows, n, k = shape
    g = torch.Generator().manual_seed(n + k)
    s = torch.round(torch.rand(rows, n, generator=g) * 6) / 6              # 7 distinct values: every threshold is a tie group
    pay = torch.stack([torch.randperm(10 * n, generator=g)[:n] for _ in range(rows)]).int()
    vals, idx = ops.topk_rows(dev(s), k, alpha=0.0, idx_in=dev(pay))
    vals, idx = vals.cpu(), idx.cpu()
    for r in range(rows):
        order = sorted(range(n), key=lambda i: (-float(s[r, i]), int(pay[r, i])))[:k]
        assert idx[r].tolist() == [int(pay[r, i]) for i in order], r
        assert vals[r].tolist() == [float(s[r, i]) for i in order], r


def test_convse_pair_inversion_paths_vs_oracle(ops):
    """The three ways K7 inverts its pair list -- one workgroup (small batches), global counters, 16 sub-counters per video
    (many pairs on few videos: TVR val's 500 pairs per video) -- against the oracle on the same inputs."""
    for nq, nv, k, l, h in ((60, 9, 5, 48, 128),            # 300 pairs: the single-workgroup path
                            (700, 5000, 50, 32, 128),       # 35 000 pairs on 5 000 videos: global counters
                            (700, 100, 50, 64, 128)):       # 35 000 pairs on 100 videos: sub-counters
        lpad = (l + 15) // 16 * 16
        g = torch.Generator().manual_seed(nq + nv)
        q = [rnd(nq, h, seed=3, scale=0.3)]
        mask = _ragged_mask(nv, l, 5)
        cw = rnd(2 * 5, seed=7, scale=0.5)
        f = [rnd(nv, l, h, seed=9, scale=0.3)]
        pair = torch.stack([torch.randperm(nv, generator=g)[:k] for _ in range(nq)]).int()
        # oracle on the selected pairs only (the (nq, nv, l) tensor of the dense restatement would be 0.9 GB at nv = 5 000)
        fsel = f[0][pair.long()]                                                 # (nq, k, l, h)
        sim = torch.einsum("qd,qkld->qkl", q[0], fsel)
        conv = lambda x, w: torch.nn.functional.conv1d(x.reshape(nq * k, 1, l), w.view(1, 1, 5), padding=2).view(nq, k, l)   # noqa: E731
        msel = mask[pair.long()]
        want_st = torch.softmax(O.mask_logits(conv(sim, cw[:5]), msel), -1)
        want_ed = torch.softmax(O.mask_logits(conv(sim, cw[5:]), msel), -1)
        fp = torch.zeros(nv, lpad, h); fp[:, :l] = f[0]
        mp = torch.zeros(nv, lpad); mp[:, :l] = mask
        st, ed = ops.convse_rerank([dev(q[0], torch.float32)], [dev(fp, torch.float32)], [dev(mp)], dev(pair), dev(cw), l, False,
                                   5, softmax=True)
        close("st (%d pairs on %d videos)" % (nq * k, nv), st[..., :l], want_st, 1e-5, 1e-4)
        close("ed (%d pairs on %d videos)" % (nq * k, nv), ed[..., :l], want_ed, 1e-5, 1e-4)
