"""Feature ingest ("next" row 8f-3): container round trip and the reference's truncate / pad / normalise semantics."""
import numpy as np
import pytest
import torch


def _make(tmp_path, dims=(48, 32), n=7, seed=0):
    from tvretrieval_amd import ingest
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, 60, n)
    names = ["vid_%d" % i for i in range(n)]
    vf = {k: rng.standard_normal((l, dims[0])).astype(np.float32) for k, l in zip(names, lens)}
    sf = {k: rng.standard_normal((l, dims[1])).astype(np.float32) for k, l in zip(names, lens)}
    ingest.write_feature_store(str(tmp_path / "vid"), vf, dtype="float32")
    ingest.write_feature_store(str(tmp_path / "sub"), sf, dtype="float16")
    return names, vf, sf, ingest.FeatureStore(str(tmp_path / "vid")), ingest.FeatureStore(str(tmp_path / "sub"))


def _expected(feats, names, max_l, normalize):
    """what start_end_dataset.py:311-321 + start_end_collate give for this batch"""
    seqs = []
    for n in names:
        a = np.asarray(feats[n][:max_l], dtype=np.float32)
        if normalize:
            a = a / (np.linalg.norm(a, axis=-1, keepdims=True) + 1e-5)      # utils/basic_utils.py:82-84
        seqs.append(a)
    l = max(len(s) for s in seqs)
    out = np.zeros((len(seqs), l, seqs[0].shape[1]), np.float32)
    mask = np.zeros((len(seqs), l), np.float32)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
        mask[i, :len(s)] = 1
    return out, mask


def test_store_roundtrip_and_dataset_contract(tmp_path):
    from tvretrieval_amd import ingest
    names, vf, sf, vs, ss = _make(tmp_path)
    for n in names:
        np.testing.assert_array_equal(np.asarray(vs[n]), vf[n])
        np.testing.assert_array_equal(np.asarray(ss[n]), sf[n].astype(np.float16))
    desc = {str(10 + i): np.random.default_rng(i).standard_normal((5 + i, 32)).astype(np.float32) for i in range(4)}
    ingest.write_feature_store(str(tmp_path / "desc"), desc, dtype="float32")
    ds = ingest.StoreEvalDataset([dict(desc_id=10 + i, desc="d", vid_name=names[i]) for i in range(4)],
                                 [dict(vid_name=n, duration=1.0) for n in names], {n: i for i, n in enumerate(names)},
                                 ingest.FeatureStore(str(tmp_path / "desc")), vs, ss, max_desc_len=6, max_ctx_len=20)
    ds.set_data_mode("context")
    assert len(ds) == len(names)
    item = ds[2]
    want, _ = _expected(vf, [names[2]], 20, True)
    np.testing.assert_allclose(item["model_inputs"]["video_feat"], want[0], rtol=1e-6)
    ds.set_data_mode("query")
    ds.load_gt_vid_name_for_query(True)
    q = ds[3]
    assert q["meta"]["vid_name"] == names[3] and q["model_inputs"]["query_feat"].shape == (6, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [True, False])
def test_context_feeder_matches_reference_collate(tmp_path, normalize):
    from tvretrieval_amd import ingest
    names, vf, sf, vs, ss = _make(tmp_path, n=11)
    feeder = ingest.ContextFeeder(names, vs, ss, max_ctx_len=40, batch_size=4, normalize_vfeat=normalize,
                                  normalize_tfeat=normalize, device="cuda:0")
    assert len(feeder) == 3
    sf16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sf.items()}
    for bi, (v, vm, s, sm) in enumerate(feeder):
        bn = names[bi * 4:bi * 4 + 4]
        wv, wm = _expected(vf, bn, 40, normalize)
        ws, _ = _expected(sf16, bn, 40, normalize)
        np.testing.assert_array_equal(vm.cpu().numpy(), wm)
        np.testing.assert_array_equal(sm.cpu().numpy(), wm)
        np.testing.assert_allclose(v.cpu().numpy(), wv, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=2e-6, atol=1e-7)


def _golden_stores(tmp_path):
    """The raw arrays of tests/golden/ingest_collate.npz (made by the reference's own l2_normalize_np_array +
    pad_sequences_1d, tools/make_golden.py::gen_ingest_case) written into FeatureStores."""
    from conftest import GOLDEN
    from tvretrieval_amd import ingest
    import os
    z = np.load(os.path.join(GOLDEN, "ingest_collate.npz"))
    lens = z["lens"]
    names = ["vid_%d" % i for i in range(len(lens))]
    stores = {}
    for tag in ("video", "sub"):
        off = np.concatenate([[0], np.cumsum(lens)])
        feats = {n: z["raw/" + tag][off[i]:off[i + 1]] for i, n in enumerate(names)}
        ingest.write_feature_store(str(tmp_path / tag), feats, dtype="float32")
        stores[tag] = ingest.FeatureStore(str(tmp_path / tag))
    return z, names, stores


def test_dataset_items_match_reference_fixture(tmp_path):
    """CPU: StoreEvalDataset items (truncate + l2_normalize_np_array) against the reference-made fixture."""
    from tvretrieval_amd import ingest
    z, names, stores = _golden_stores(tmp_path)
    max_l, bsz = int(z["max_l"]), int(z["bsz"])
    desc = {"0": np.ones((3, 8), np.float32)}
    ingest.write_feature_store(str(tmp_path / "desc"), desc, dtype="float32")
    ds = ingest.StoreEvalDataset([dict(desc_id=0, desc="d", vid_name=names[0])], [dict(vid_name=n, duration=1.0) for n in names],
                                 {n: i for i, n in enumerate(names)}, ingest.FeatureStore(str(tmp_path / "desc")),
                                 stores["video"], stores["sub"], max_desc_len=6, max_ctx_len=max_l)
    ds.set_data_mode("context")
    for i in range(len(names)):
        item = ds[i]["model_inputs"]
        for tag in ("video", "sub"):
            want = z["%s/norm/batch%d/feat" % (tag, i // bsz)][i % bsz]
            n = min(int(z["lens"][i]), max_l)
            np.testing.assert_allclose(item[tag + "_feat"], want[:n], rtol=1e-6, atol=1e-8)
            assert (want[n:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("normalize", [True, False])
def test_context_feeder_matches_reference_fixture(tmp_path, normalize):
    """GPU: ContextFeeder batches (pinned staging, device-side xml_l2norm_rows_eps) against padded features + masks made
    by the reference's l2_normalize_np_array + pad_sequences_1d (tests/golden/ingest_collate.npz)."""
    from tvretrieval_amd import ingest
    z, names, stores = _golden_stores(tmp_path)
    max_l, bsz = int(z["max_l"]), int(z["bsz"])
    feeder = ingest.ContextFeeder(names, stores["video"], stores["sub"], max_ctx_len=max_l, batch_size=bsz,
                                  normalize_vfeat=normalize, normalize_tfeat=normalize, device="cuda:0")
    kind = "norm" if normalize else "raw"
    for bi, (v, vm, s, sm) in enumerate(feeder):
        for tag, got, gm in (("video", v, vm), ("sub", s, sm)):
            want, wm = z["%s/%s/batch%d/feat" % (tag, kind, bi)], z["%s/%s/batch%d/mask" % (tag, kind, bi)]
            np.testing.assert_array_equal(gm.cpu().numpy(), wm)
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=1e-7)
    assert bi == (len(names) - 1) // bsz
