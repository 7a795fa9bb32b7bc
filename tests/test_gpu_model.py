"""GPU parity tests at the drop-in boundary: tvretrieval_amd.model_xml.XML / tvretrieval_amd.inference against
(a) the golden vectors captured from the reference and (b) the CPU oracle on larger seeded inputs.
fp32 compute: scores within 1e-4, top-k indices / spans identical (ties within rounding excepted).
bf16 compute: reported as top-k overlap (BASELINE.json: bf16 configs are judged by overlap / R@1)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import xml_oracle as O
from test_gpu_kernels import close, DEV

pytestmark = pytest.mark.gpu

MODEL_CASES = ["xml_video_sub_cross_h128", "xml_video_only_h256", "xml_sub_only_h128",
               "xml_video_sub_nocross_nomerge_h128"]


def build_model(cfg, sd, dtype=torch.float32):
    from tvretrieval_amd.model_xml import XML
    m = XML(cfg, compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(DEV).eval()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_golden_fp32(name):
    d, cfg, sd = load_golden(name)
    m = build_model(cfg, sd)
    dummy = torch.zeros(len(d["ctx_lens"]), 2, 2, device=DEV)
    vf = T(d["video_feat"]) if m.use_video else dummy
    vm = T(d["video_mask"]) if m.use_video else dummy
    sf = T(d["sub_feat"]) if m.use_sub else dummy
    sm = T(d["sub_mask"]) if m.use_sub else dummy
    with torch.no_grad():
        # intermediates first: localise a failure to one kernel
        first = "video" if m.use_video else "sub"
        proj, enc1 = getattr(m, first + "_input_proj"), getattr(m, first + "_encoder1")
        feat, mask = (vf, vm) if m.use_video else (sf, sm)
        from tvretrieval_amd import ops
        p, e = proj.packed(torch.float32), m.ctx_pos_embed.packed(torch.float32)
        x = ops.linear_ln_relu_pos(feat, p["ln_g"], p["ln_b"], p["w"], p["b"], e["pos"], e["ln_g"], e["ln_b"])
        close("ctx_pos_embed out", x, d["int/ctx_pos_embed"], 1e-4)
        v1, v2, s1, s2 = m.encode_context(vf, vm, sf, sm)
        for k, t in (("vf1", v1), ("vf2", v2), ("sf1", s1), ("sf2", s2)):
            if k in d:
                close(k + " (incl. padded rows)", t, d[k], 2e-4)
            else:
                assert t is None
        vq, sq = m.encode_query(T(d["query_feat"]), T(d["query_mask"]))
        close("video_query", vq, d["video_query"], 1e-4)
        close("sub_query", sq, d["sub_query"], 1e-4)
        q2c, st, ed = m.get_pred_from_raw_query(T(d["query_feat"]), T(d["query_mask"]), v1, v2,
                                                vm if m.use_video else None, s1, s2, sm if m.use_sub else None,
                                                cross=True)
        close("q2c cross", q2c, d["q2c_cross"], 1e-4)
        close("st cross", st, d["st_cross"], 1e-3, 1e-6)     # logits are O(1..10); -1e10 fills compare exactly
        close("ed cross", ed, d["ed_cross"], 1e-3, 1e-6)
        n = d["q2c_pair"].shape[0]
        sel = lambda t: None if t is None else t[:n].contiguous()
        q2c_p, st_p, ed_p = m.get_pred_from_raw_query(T(d["query_feat"])[:n], T(d["query_mask"])[:n], sel(v1), sel(v2),
                                                      sel(vm) if m.use_video else None, sel(s1), sel(s2),
                                                      sel(sm) if m.use_sub else None, cross=False)
        close("q2c pair", q2c_p, d["q2c_pair"], 1e-4)
        close("st pair", st_p, d["st_pair"], 1e-3, 1e-6)
        close("ed pair", ed_p, d["ed_pair"], 1e-3, 1e-6)


class GoldenDataset(object):
    """The reference's eval-dataset contract (xml/start_end_dataset.py:171-343) over fixture arrays."""

    def __init__(self, d, use_video, use_sub):
        self.d = d
        self.n_v = len(d["ctx_lens"])
        self.n_q = len(d["query_gt_video"])
        self.use_video, self.use_sub = use_video, use_sub
        self.video2idx = {"vid_%03d" % i: int(d["video_idx"][i]) for i in range(self.n_v)}
        self.mode, self.gt = "query", False

    def set_data_mode(self, mode):
        self.mode = mode

    def load_gt_vid_name_for_query(self, flag):
        self.gt = flag

    def __len__(self):
        return self.n_q if self.mode == "query" else self.n_v

    def __getitem__(self, i):
        if self.mode == "context":
            mi = dict(video_feat=self.d["video_feat/%d" % i], sub_feat=self.d["sub_feat/%d" % i])
            return dict(meta=dict(vid_name="vid_%03d" % i, duration=0.0), model_inputs=mi)
        meta = dict(desc_id=5000 + i, desc="query %d" % i,
                    vid_name="vid_%03d" % int(self.d["query_gt_video"][i]) if self.gt else None)
        return dict(meta=meta, model_inputs=dict(query_feat=self.d["query_feat/%d" % i]))


def _compare_lists(got, want, task):
    """[video_idx, st, ed, score] lists: scores to 1e-4 relative; ids/spans identical unless scores tie to rounding."""
    assert len(got) == want.shape[0]
    for qi, e in enumerate(got):
        g = np.array(e["predictions"], dtype=np.float64).reshape(-1, 4)
        w = want[qi]
        w = w[w[:, 3] > 0]
        assert len(g) >= min(len(w), len(g)) and len(g) <= want.shape[1]
        n = min(len(g), len(w))
        assert n == len(w) or len(g) == want.shape[1], (task, qi, len(g), len(w))
        np.testing.assert_allclose(g[:n, 3], w[:n, 3], rtol=1e-4, err_msg="%s q%d scores" % (task, qi))
        for i in np.nonzero((g[:n, :3] != w[:n, :3]).any(1))[0]:
            j = np.nonzero((w[:n, :3] == g[i, :3]).all(1))[0]
            assert len(j) == 1 and abs(w[j[0], 3] - w[i, 3]) <= 2e-4 * w[i, 3], (task, qi, i, g[i], w[i])


@pytest.mark.parametrize("name", ["pipeline_video_sub_h128", "pipeline_video_only_h128"])
def test_golden_pipeline_fp32(name):
    """compute_context_info -> compute_query2ctx_info(VCMR, SVMR, VR) vs the reference driver's own output."""
    import argparse
    from tvretrieval_amd import inference as inf
    d, cfg, sd = load_golden(name)
    o = json.loads(str(d["opt"]))
    m = build_model(cfg, sd)
    ds = GoldenDataset(d, m.use_video, m.use_sub)
    opt = argparse.Namespace(eval_context_bsz=o["eval_context_bsz"], eval_query_bsz=o["eval_query_bsz"],
                             device=torch.device(DEV), q2c_alpha=o["q2c_alpha"], min_pred_l=o["min_pred_l"],
                             max_pred_l=o["max_pred_l"], clip_length=o["clip_length"], debug=False,
                             external_inference_vr_res_path=None, max_ctx_l=cfg["max_ctx_l"])
    with torch.no_grad():
        ctx = inf.compute_context_info(m, ds, opt)
        for k in ["video_feat1", "video_feat2", "video_mask", "sub_feat1", "sub_feat2", "sub_mask"]:
            if ("ctx/" + k) in d:
                want = d["ctx/" + k]
                if want.ndim == 3:
                    # valid clips: tight.  Padded clips of cross-attended streams see (score - 10000) in fp32, i.e.
                    # scores quantised to 2^-10 in the reference as well as here: compare those at 1e-3.
                    valid = d["ctx/" + k.split("_")[0] + "_mask"][..., None] > 0
                    got = ctx[k].float().cpu().numpy()
                    close("ctx " + k + " (valid clips)", np.where(valid, got, 0), np.where(valid, want, 0), 2e-4)
                    close("ctx " + k + " (padded clips)", np.where(valid, 0, got), np.where(valid, 0, want), 1e-3)
                else:
                    close("ctx " + k, ctx[k], want, 0)
        res = inf.compute_query2ctx_info(m, ds, opt, ctx, max_before_nms=o["max_before_nms"],
                                         max_n_videos=o["max_vcmr_video"], tasks=("SVMR", "VCMR", "VR"))
    for task in ("VR", "VCMR", "SVMR"):
        _compare_lists(res[task], d["res/" + task], task)
        span = int if task == "VR" else float          # the reference's VR entries are [video_idx, 0, 0, score]
        assert all(type(p[0]) is int and type(p[1]) is span and type(p[2]) is span and type(p[3]) is float
                   for e in res[task] for p in e["predictions"]), "the reference's element types"
    # the array form of the same results (what eval_epoch works on) and the device tuple vs the oracle's numpy tail
    from tvretrieval_amd.results import MomentResults
    with torch.no_grad():
        arr = inf.compute_query2ctx_info(m, ds, opt, ctx, max_before_nms=o["max_before_nms"],
                                         max_n_videos=o["max_vcmr_video"], tasks=("SVMR", "VCMR", "VR"), as_arrays=True)
    for task in ("VR", "VCMR", "SVMR"):
        assert isinstance(arr[task], MomentResults) and arr[task].to_list() == res[task]
    # opt.graph_search: the same batches padded to (eval_query_bsz, max_desc_l) and replayed through one captured graph -- the
    # same lists, bit for bit (a last, shorter batch is filled up with dummy rows that are not decoded)
    import copy
    opt_g = copy.copy(opt)
    opt_g.graph_search = True
    opt_g.max_desc_l = int(m.config.max_desc_l)        # the positional table: batches are padded up to it
    assert opt_g.max_desc_l >= max(int(d["query_feat/%d" % i].shape[0]) for i in range(ds.n_q))
    with torch.no_grad():
        arr_g = inf.compute_query2ctx_info(m, ds, opt_g, ctx, max_before_nms=o["max_before_nms"],
                                           max_n_videos=o["max_vcmr_video"], tasks=("SVMR", "VCMR", "VR"), as_arrays=True)
    for task in ("VR", "VCMR", "SVMR"):
        a, g_ = arr[task], arr_g[task]
        np.testing.assert_array_equal(a.count, g_.count, err_msg=task + " counts (graphed batches)")
        for col in ("vid", "st", "ed", "score"):
            np.testing.assert_array_equal(getattr(a, col), getattr(g_, col), err_msg="%s.%s (graphed batches)" % (task, col))
    meta_vid = np.array([ds.video2idx["vid_%03d" % i] for i in range(ds.n_v)])
    v = arr["VCMR"]
    bs = o["eval_query_bsz"]
    for b in range(0, ds.n_q, bs):          # the driver's own batching: the same launches, bit for bit
        e = min(ds.n_q, b + bs)
        with torch.no_grad():
            qf, qm = inf.pad_batch([d["query_feat/%d" % i] for i in range(b, e)], DEV)
            out = inf.vcmr_search(m, ctx["index"], qf, qm, max_vcmr_video=o["max_vcmr_video"],
                                  max_before_nms=o["max_before_nms"], q2c_alpha=o["q2c_alpha"],
                                  min_pred_l=o["min_pred_l"], max_pred_l=o["max_pred_l"])
        fi, ti = out["flat_indices"].cpu().numpy(), out["top_indices"].cpu().numpy()
        ok = fi >= 0
        vid, st_s, ed_s = O.unravel_moments(np.where(ok, fi, 0), ti, ctx["index"].l_ref, clip_length=o["clip_length"])
        n = fi.shape[1]
        np.testing.assert_array_equal(v.count[b:e], ok.sum(1))
        np.testing.assert_array_equal(v.vid[b:e, :n], np.where(ok, meta_vid[vid], -1))
        np.testing.assert_array_equal(v.st[b:e, :n], np.where(ok, st_s, 0).astype(np.float64))
        np.testing.assert_array_equal(v.ed[b:e, :n], np.where(ok, ed_s, 0).astype(np.float64))
        np.testing.assert_array_equal(v.score[b:e, :n], np.where(ok, out["flat_scores"].cpu().numpy(), 0).astype(np.float64))


def test_golden_external_vr_fp32(tmp_path):
    """The external-VR hook against the REFERENCE's own run of that branch (xml/inference.py:244-249,
    264-273,349-355; fixture made by tools/make_golden.py::gen_external_vr_case): another model's VR submission -- 8
    videos per query, trimmed to max_vcmr_video = 6 by get_submission_top_n -- replaces K6 + K8; VCMR moments are
    re-ranked inside those videos with weights exp(alpha * s); the VR list is the external one."""
    import argparse
    from tvretrieval_amd import inference as inf
    d, cfg, sd = load_golden("pipeline_external_vr_h128")
    o = json.loads(str(d["opt"]))
    m = build_model(cfg, sd)
    ds = GoldenDataset(d, m.use_video, m.use_sub)
    ext = dict(video2idx=ds.video2idx,
               VR=[dict(desc_id=5000 + i, desc="query %d" % i,
                        predictions=[[int(v), 0, 0, float(s)] for v, s in zip(d["ext/video_idx"][i], d["ext/score"][i])])
                   for i in range(ds.n_q)])
    path = str(tmp_path / "external_vr.json")
    with open(path, "w") as f:
        json.dump(ext, f)
    opt = argparse.Namespace(eval_context_bsz=o["eval_context_bsz"], eval_query_bsz=o["eval_query_bsz"],
                             device=torch.device(DEV), q2c_alpha=o["q2c_alpha"], min_pred_l=o["min_pred_l"],
                             max_pred_l=o["max_pred_l"], clip_length=o["clip_length"], debug=False,
                             external_inference_vr_res_path=path, max_ctx_l=cfg["max_ctx_l"])
    with torch.no_grad():
        ctx = inf.compute_context_info(m, ds, opt)
        res = inf.compute_query2ctx_info(m, ds, opt, ctx, max_before_nms=o["max_before_nms"],
                                         max_n_videos=o["max_vcmr_video"], tasks=("SVMR", "VCMR", "VR"))
    for task in ("VR", "VCMR", "SVMR"):
        _compare_lists(res[task], d["res/" + task], task + " (external VR)")
    for e, want in zip(res["VR"], d["res/VR"]):      # the external list itself, position by position
        assert [p[0] for p in e["predictions"]] == [int(v) for v in want[:, 0]]


@pytest.mark.parametrize("name", ["xml_video_sub_cross_h128", "xml_video_only_h256"])
def test_golden_bf16_overlap(name):
    """bf16 compute on the same fixtures: features within bf16 rounding of the fp32 reference, and the video
    ranking agrees with the reference wherever the reference's own margin exceeds bf16 noise."""
    d, cfg, sd = load_golden(name)
    m = build_model(cfg, sd, torch.bfloat16)
    dummy = torch.zeros(len(d["ctx_lens"]), 2, 2, device=DEV)
    with torch.no_grad():
        v1, v2, s1, s2 = m.encode_context(T(d["video_feat"]) if m.use_video else dummy,
                                          T(d["video_mask"]) if m.use_video else dummy,
                                          T(d["sub_feat"]) if m.use_sub else dummy,
                                          T(d["sub_mask"]) if m.use_sub else dummy)
        for k, t in (("vf1", v1), ("vf2", v2), ("sf1", s1), ("sf2", s2)):
            if k in d:
                assert t.dtype == torch.bfloat16
                close(k, t, d[k], 0.12, 0.02)
        q2c, st, ed = m.get_pred_from_raw_query(T(d["query_feat"]), T(d["query_mask"]), v1, v2,
                                                T(d["video_mask"]) if m.use_video else None, s1, s2,
                                                T(d["sub_mask"]) if m.use_sub else None, cross=True)
    close("q2c bf16", q2c, d["q2c_cross"], 2e-2)
    want = d["q2c_cross"]
    got = q2c.cpu().numpy()
    for q in range(len(want)):
        order = np.argsort(-want[q])
        if want[q][order[0]] - want[q][order[1]] > 5e-2:
            assert int(np.argmax(got[q])) == int(order[0])


def _synthetic_model(ctx_mode, hidden, dv, ds_, dq, max_ctx_l, dtype, seed=0, cross=True, merge=True):
    from tvretrieval_amd.model_xml import XML
    cfg = dict(merge_two_stream=merge, cross_att=cross, span_predictor_type="conv", encoder_type="transformer",
               visual_input_size=dv, sub_input_size=ds_, query_input_size=dq, hidden_size=hidden, conv_kernel_size=5,
               stack_conv_predictor_conv_kernel_sizes=-1, conv_stride=1, max_ctx_l=max_ctx_l, max_desc_l=30,
               input_drop=0.1, drop=0.1, n_heads=4, initializer_range=0.02, ctx_mode=ctx_mode, margin=0.1,
               ranking_loss_type="hinge", lw_neg_q=1, lw_neg_ctx=1, lw_st_ed=0.01, use_hard_negative=False,
               hard_pool_size=20, use_self_attention=True, no_modular=False)
    if ctx_mode != "video_sub":
        cfg["merge_two_stream"] = False
        cfg["cross_att"] = False
    torch.manual_seed(seed)
    m = XML(cfg, compute_dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # richer than N(0, 0.02): avoids near-identical videos (see tools/make_golden.py)
        for n_, p in m.named_parameters():
            if n_.lower().endswith("layernorm.weight"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n_.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "predictor" in n_:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[-1]))
            p.copy_(p.to(torch.bfloat16).float())
    return m.to(DEV).eval(), cfg


def _feats(n, lens, dim, seed):
    g = torch.Generator().manual_seed(seed)
    l = int(max(lens))
    x = torch.zeros(n, l, dim)
    mask = torch.zeros(n, l)
    for i, li in enumerate(lens):
        v = torch.randn(int(li), dim, generator=g)
        x[i, :li] = (v / (v.norm(dim=-1, keepdim=True) + 1e-5)).to(torch.bfloat16).float()
        mask[i, :li] = 1
    return x, mask


@pytest.mark.parametrize("ctx_mode,hidden", [("video_sub", 768), ("video", 256)])
def test_search_vs_oracle_fp32(ctx_mode, hidden):
    """BASELINE configs[0]/[1]-shaped case at a size the oracle finishes in seconds: full VCMR search (encode,
    K6, top-k, ConvSE, moment top-n) against the reference formulation on CPU."""
    from tvretrieval_amd import inference as inf
    nv, nq, l = 48, 33, 128
    m, cfg = _synthetic_model(ctx_mode, hidden, 512, 256, 256, l, torch.float32, seed=5)
    rng = np.random.default_rng(3)
    lens = rng.integers(20, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 512, 1)
    sf, sm = _feats(nv, lens, 256, 2)
    qf, qm = _feats(nq, rng.integers(5, 31, nq), 256, 3)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    om = O.OracleXML(cfg, sd)
    with torch.no_grad():
        ov1, ov2, os1, os2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm if om.use_video else None, os1, os2,
                                                 sm if om.use_sub else None, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, 10, 2, 16, 216)
        bs = 16
        batches = [(vf[b:b + bs].to(DEV), vm[b:b + bs].to(DEV), sf[b:b + bs].to(DEV), sm[b:b + bs].to(DEV))
                   for b in range(0, nv, bs)]
        # all batches are padded to the same L here, so batching does not change padded-row semantics
        index = inf.build_corpus_index(m, batches)
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=200)
    close("q2c", out["q2c"], q2c, 1e-4)
    close("top scores", out["top_scores"], want["top_scores"], 0, 2e-3)   # exp(20 s): 1e-4 on s -> 2e-3 relative
    gi, wi = out["top_indices"].cpu().numpy(), want["top_indices"].numpy()
    ws = want["top_scores"].numpy()
    for q in range(nq):
        for i in np.nonzero(gi[q] != wi[q])[0]:
            j = np.nonzero(wi[q] == gi[q][i])[0]
            assert len(j) == 1 and abs(ws[q][j[0]] - ws[q][i]) <= 4e-3 * ws[q][i], ("video rank", q, i)
    from oracle.listcmp import moment_keys, tie_aware_equal
    same = np.nonzero((np.sort(gi, 1) == np.sort(wi, 1)).all(1))[0]      # same top-10 SET => same candidate pool
    assert len(same) >= 0.9 * nq
    fs, fi = out["flat_scores"].cpu().numpy(), out["flat_indices"].cpu().numpy()
    gk, wk = moment_keys(fi, gi, l), moment_keys(want["flat_indices"].numpy(), wi, l)
    assert (fi[same] >= 0).all()
    n_sw = tie_aware_equal(gk[same], fs[same], wk[same], want["flat_scores"].numpy()[same], 200, 5e-4, "moments")
    assert n_sw <= 0.02 * len(same) * 200, n_sw


def test_video_with_one_empty_modality_matches_the_reference_lists():
    """Two unmerged streams whose masks differ per video: video 1 has 4 subtitle clips and 9 visual ones, video 2 no visual
    clip at all.  The reference averages the two streams' masked LOGITS and applies the softmax afterwards
    (xml/model_xml.py:436-453, xml/inference.py:365-370): a position masked in one stream sits at -5e9, in both at -1e10, so
    the probability past the SHORTER stream is exactly 0 and CorpusIndex.set_valid_lengths' max-over-modalities length skips
    nothing that counts (ADVICE r5 asked); a video with an empty stream has similarity -5e9, i.e. weight exp(20 s) = 0.
    Top-1000 of three videos against the oracle: identical positive-score lists, video 1's moments among them."""
    from tvretrieval_amd import inference as inf
    from oracle.listcmp import moment_keys, tie_aware_equal
    nv, nq, l, n = 3, 9, 64, 1000
    m, cfg = _synthetic_model("video_sub", 128, 256, 128, 128, l, torch.float32, seed=9, cross=False, merge=False)
    lens = np.array([l, 4, 4])
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    vf9, vm9 = _feats(1, [9], 256, 7)
    vf[1, :9], vm[1, :9] = vf9[0], vm9[0]       # video 1: 9 visual clips, 4 subtitle clips
    vf[2], vm[2] = 0, 0                         # video 2: no visual clip at all
    qf, qm = _feats(nq, [5, 30, 17, 9, 30, 12, 8, 21, 3], 128, 3)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        ov1, ov2, os1, os2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm, os1, os2, sm, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, nv, 2, 16, n)
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=nv, max_before_nms=n)
    assert index.vlen.cpu().tolist() == [l, 9, 4]
    close("q2c", out["q2c"], q2c, 1e-4)
    gi, wi = out["top_indices"].cpu().numpy(), want["top_indices"].numpy()
    assert (gi == wi).all()
    gk, wk = moment_keys(out["flat_indices"].cpu().numpy(), gi, l), moment_keys(want["flat_indices"].numpy(), wi, l)
    gs, ws = out["flat_scores"].cpu().numpy(), want["flat_scores"].numpy()
    seen = 0
    for q in range(nq):
        npos = int((ws[q] > 0).sum())                     # the reference pads its list with zero-score moments; K9 stops
        assert npos < n and (gk[q][npos:] == -1).all() and (gk[q][:npos] >= 0).all(), (q, npos)
        tie_aware_equal(gk[q:q + 1, :npos], gs[q:q + 1], wk[q:q + 1, :npos], ws[q:q + 1], npos, 5e-4, "moments")
        seen += int(((wk[q][:npos] // (l * l)) == 1).sum())
    assert seen >= nq, "video 1 never reached a list"


def test_ragged_corpus_buckets_give_identical_lists(monkeypatch):
    """TVR-like clip counts: the index built with length buckets and the one built without return the same top-100 /
    top-200 lists, bit for bit (bf16, the headline dtype)."""
    from tvretrieval_amd import inference as inf
    nv, nq, l = 300, 70, 128
    m, cfg = _synthetic_model("video_sub", 256, 512, 256, 256, l, torch.bfloat16, seed=6)
    rng = np.random.default_rng(4)
    lens = np.clip(np.round(rng.normal(51, 14, nv)), 8, 128).astype(int); lens[0] = l
    vf, vm = _feats(nv, lens, 512, 1)
    sf, sm = _feats(nv, lens, 256, 2)
    qf, qm = _feats(nq, rng.integers(5, 31, nq), 256, 3)
    batches = [(vf[b:b + 100].to(DEV), vm[b:b + 100].to(DEV), sf[b:b + 100].to(DEV), sm[b:b + 100].to(DEV))
               for b in range(0, nv, 100)]
    with torch.no_grad():
        bucketed = inf.build_corpus_index(m, batches, l_ref=l)
        monkeypatch.setenv("XML_Q2C_NO_BUCKETS", "1")
        flat = inf.build_corpus_index(m, batches, l_ref=l)
        assert bucketed.feat1n["video"].plan is not None and flat.feat1n["video"].plan is None
        a = inf.vcmr_search(m, bucketed, qf.to(DEV), qm.to(DEV))
        b = inf.vcmr_search(m, flat, qf.to(DEV), qm.to(DEV))
    for k in ("q2c", "top_scores", "top_indices", "flat_scores", "flat_indices"):
        assert torch.equal(a[k], b[k]), k


def test_tef_context_mode_dims_vs_oracle():
    """video_sub_tef checkpoints (xml/config.py:108-110,251-254): visual 3072+2 / sub 768+2 input dims (the dataset
    concatenates the two TEF columns, xml/start_end_dataset.py:127-142).  Encode + search vs the oracle, fp32."""
    from tvretrieval_amd import inference as inf
    nv, nq, l = 10, 7, 64
    m, cfg = _synthetic_model("video_sub", 256, 3074, 770, 768, l, torch.float32, seed=9)
    rng = np.random.default_rng(5)
    lens = rng.integers(10, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 3074, 1)
    sf, sm = _feats(nv, lens, 770, 2)
    qf, qm = _feats(nq, rng.integers(5, 31, nq), 768, 3)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        ov1, ov2, os1, os2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm, os1, os2, sm, cross=True)
        v1, v2, s1, s2 = m.encode_context(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=5, max_before_nms=50)
    for name, a, b in (("vf1", v1, ov1), ("vf2", v2, ov2), ("sf1", s1, os1), ("sf2", s2, os2)):
        close(name, a, b, 2e-4)
    close("q2c", out["q2c"], q2c, 1e-4)
    want = O.vcmr_tail(q2c, st, ed, 20.0, 5, 2, 16, 50)
    assert torch.equal(out["top_indices"].cpu().long(), want["top_indices"])


def test_svmr_only_external_vr_and_eval_epoch(tmp_path):
    """"next" rows 8f-1/2/4 end to end on the golden pipeline fixture: SVMR-only path == the SVMR list of the full
    run, external-VR re-ranking fed with this model's own VR output reproduces its VCMR list, eval_epoch runs NMS
    + evaluator on top."""
    import argparse
    import copy
    from tvretrieval_amd import inference as inf
    d, cfg, sd = load_golden("pipeline_video_sub_h128")
    o = json.loads(str(d["opt"]))
    m = build_model(cfg, sd)
    ds = GoldenDataset(d, m.use_video, m.use_sub)
    opt = argparse.Namespace(eval_context_bsz=o["eval_context_bsz"], eval_query_bsz=o["eval_query_bsz"],
                             device=torch.device(DEV), q2c_alpha=o["q2c_alpha"], min_pred_l=o["min_pred_l"],
                             max_pred_l=o["max_pred_l"], clip_length=o["clip_length"], debug=False,
                             external_inference_vr_res_path=None, max_ctx_l=cfg["max_ctx_l"],
                             max_before_nms=o["max_before_nms"], max_vcmr_video=o["max_vcmr_video"], nms_thd=o["nms_thd"],
                             dset_name="tvr")
    with torch.no_grad():
        ctx = inf.compute_context_info(m, ds, opt)
        full = inf.compute_query2ctx_info(m, ds, opt, ctx, max_before_nms=o["max_before_nms"],
                                          max_n_videos=o["max_vcmr_video"], tasks=("SVMR", "VCMR", "VR"))
        svmr = inf.compute_query2ctx_info_svmr_only(m, ds, opt, ctx, max_before_nms=o["max_before_nms"])
        assert svmr["SVMR"] == full["SVMR"]
        _compare_lists(svmr["SVMR"], d["res/SVMR"], "SVMR-only")
        # external VR: cosine-like scores s with exp(alpha*s) == this model's weights
        ext = dict(video2idx=ds.video2idx, VR=copy.deepcopy(full["VR"]))
        for e in ext["VR"]:
            for p in e["predictions"]:
                p[3] = float(np.log(p[3]) / o["q2c_alpha"])
        path = str(tmp_path / "ext_vr.json")
        json.dump(ext, open(path, "w"))
        opt.external_inference_vr_res_path = path
        rer = inf.compute_query2ctx_info(m, ds, opt, ctx, max_before_nms=o["max_before_nms"],
                                         max_n_videos=o["max_vcmr_video"], tasks=("VCMR", "VR"))
        opt.external_inference_vr_res_path = None
        for a, b in zip(rer["VCMR"], full["VCMR"]):
            ga, gb = np.array(a["predictions"]), np.array(b["predictions"])
            np.testing.assert_allclose(ga[:, 3], gb[:, 3], rtol=2e-5)
            assert (ga[:, :3] == gb[:, :3]).mean() > 0.95
        gt = [dict(desc_id=5000 + i, desc="", type=["v", "t", "vt"][i % 3], vid_name="vid_%03d" % int(d["query_gt_video"][i]),
                   ts=[3.0, 9.0]) for i in range(ds.n_q)]
        sub, met, sub_nms, met_nms = inf.eval_epoch(m, ds, opt, tasks=("SVMR", "VCMR", "VR"), ground_truth=gt)
    assert set(met) == {"VCMR", "SVMR", "VR", "VCMR_by_type", "SVMR_by_type", "VR_by_type"}
    assert 0.0 <= met["VR"]["r100"] <= 100.0 and met["VR"]["r1"] <= met["VR"]["r100"]
    assert set(met_nms) == {"VCMR", "SVMR", "VCMR_by_type", "SVMR_by_type"}
    for i in range(ds.n_q):
        np.testing.assert_allclose(np.array(sub_nms["VCMR"][i]["predictions"]).reshape(-1, 4)[:, :3],
                                   d["nms/VCMR/%d" % i][:, :3], atol=1e-6)


def test_c1_single_query_single_video_svmr():
    """BASELINE configs[0]: SVMR with 1 query x 1 video (128 clips, d=768), video-only context, against the
    reference formulation: span logits, softmaxed probabilities, the banded top-n moments of the given video."""
    from tvretrieval_amd import inference as inf
    l = 128
    m, cfg = _synthetic_model("video", 768, 768, 768, 768, l, torch.float32, seed=9)
    vf, vm = _feats(1, [l], 768, 11)
    qf, qm = _feats(1, [17], 768, 12)
    dummy = torch.zeros(1, 2, 2)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    om = O.OracleXML(cfg, sd)
    with torch.no_grad():
        ov1, ov2, _, _ = om.encode_context(vf, vm, dummy, dummy)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, ov1, ov2, vm, None, None, None, cross=True)
        st_p, ed_p = torch.softmax(st[:, 0], -1), torch.softmax(ed[:, 0], -1)
        want = O.svmr_tail(st_p.numpy(), ed_p.numpy(), 2, 16, 200)          # (1, n, 3): st idx, ed idx, score
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), None, None)])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=100, max_before_nms=200,
                              svmr_video=torch.zeros(1, dtype=torch.int32, device=DEV))
    assert out["top_indices"].shape == (1, 1) and int(out["top_indices"][0, 0]) == 0
    close("q2c", out["q2c"], q2c, 1e-4)
    close("svmr st prob", out["svmr_st"][:, :l], st_p, 0, 2e-3)
    close("svmr ed prob", out["svmr_ed"][:, :l], ed_p, 0, 2e-3)
    gs, gf = out["svmr_scores"].cpu().numpy()[0], out["svmr_flat"].cpu().numpy()[0]
    ws, wf = want[0, :, 2], (want[0, :, 0] * l + want[0, :, 1]).astype(np.int64)
    n = int((ws > 0).sum())
    assert n > 100 and (gf[n:] == -1).all()
    from oracle.listcmp import tie_aware_equal
    tie_aware_equal(gf[None, :n], gs[None, :n], wf[None, :n], ws[None, :n], n, 2e-4, "SVMR moments")
    # VCMR over a one-video corpus ranks the same spans (weight exp(20 s) is a common factor)
    vf_ = out["flat_indices"].cpu().numpy()[0][:n]
    tie_aware_equal(vf_[None], out["flat_scores"].cpu().numpy()[0][None, :n] / float(out["top_scores"][0, 0]),
                    wf[None, :n], ws[None, :n], n, 2e-4, "VCMR over one video")


def test_short_corpus_zero_score_tail_is_dropped():
    """The one output-shape difference at the boundary, pinned: the reference sorts the WHOLE (k, L, L) product tensor and
    always returns max_before_nms rows (xml/inference.py:381-386); when fewer candidates pass the length mask the tail is
    padding -- rows of score exactly 0 at masked (st, ed) positions, in an order torch.sort leaves unspecified.  The HIP
    path returns the positive-score prefix and marks the rest flat = -1 / score 0 (the drivers drop those rows), so the
    lists are identical up to that meaningless tail."""
    from tvretrieval_amd import inference as inf
    nv, nq, l = 2, 3, 16                      # 2 videos x 16 clips: at most 2 * sum_{i} min(14, 16 - i - 2) candidates
    m, cfg = _synthetic_model("video_sub", 128, 64, 64, 64, l, torch.float32, seed=4)
    vf, vm = _feats(nv, [16, 9], 64, 1)
    sf, sm = _feats(nv, [16, 9], 64, 2)
    qf, qm = _feats(nq, [5, 9, 12], 64, 3)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        v1, v2, s1, s2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, v1, v2, vm, s1, s2, sm, cross=True)
        want = O.vcmr_tail(q2c, st, ed, 20.0, 2, 2, 16, 200)
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=2, max_before_nms=200)
    ws, wf = want["flat_scores"].numpy(), want["flat_indices"].numpy()
    gs, gf = out["flat_scores"].cpu().numpy(), out["flat_indices"].cpu().numpy()
    assert ws.shape == (nq, 200) == gs.shape
    from oracle.listcmp import moment_keys, tie_aware_equal
    for q in range(nq):
        n = int((ws[q] > 0).sum())
        assert 0 < n < 200                                            # the reference's list HAS a padding tail here
        assert (ws[q][n:] == 0).all()                                 # ... of exact zeros
        assert (gf[q][n:] == -1).all() and (gs[q][n:] == 0).all()     # ours: marked empty
        gk = moment_keys(gf[q:q + 1, :n], out["top_indices"].cpu().numpy()[q:q + 1], l)
        wk = moment_keys(wf[q:q + 1, :n], want["top_indices"].numpy()[q:q + 1], l)
        tie_aware_equal(gk, gs[q:q + 1, :n], wk, ws[q:q + 1, :n], n, 2e-4, "positive-score prefix")


def test_pad_tail_gives_the_reference_row_count():
    """Opt-in pad_tail=True: max_before_nms rows like the reference (xml/inference.py:381-386) -- the positive-score prefix
    unchanged, then rows of score 0 at positions whose score is exactly 0 in the reference's product tensor too, no
    position twice; through the driver every prediction list then has max_before_nms entries."""
    import argparse
    from tvretrieval_amd import inference as inf
    nv, nq, l = 2, 3, 16
    m, cfg = _synthetic_model("video_sub", 128, 64, 64, 64, l, torch.float32, seed=4)
    vf, vm = _feats(nv, [16, 9], 64, 1)
    sf, sm = _feats(nv, [16, 9], 64, 2)
    qf, qm = _feats(nq, [5, 9, 12], 64, 3)
    om = O.OracleXML(cfg, {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        v1, v2, s1, s2 = om.encode_context(vf, vm, sf, sm)
        q2c, st, ed = om.get_pred_from_raw_query(qf, qm, v1, v2, vm, s1, s2, sm, cross=True)
        t = O.vcmr_tail(q2c, st, ed, 20.0, 2, 2, 16, 2 * l * l)           # the WHOLE sorted product tensor
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        gt = torch.tensor([0, 1, 1], dtype=torch.int32, device=DEV)
        plain = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=2, max_before_nms=200, svmr_video=gt)
        out = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=2, max_before_nms=200, svmr_video=gt,
                              pad_tail=True)
    ref_score = np.zeros((nq, 2 * l * l), dtype=np.float32)               # reference score of every flat position
    np.put_along_axis(ref_score, t["flat_indices"].numpy(), t["flat_scores"].numpy(), 1)
    for key_f, key_s in (("flat_indices", "flat_scores"), ("svmr_flat", "svmr_scores")):
        gf, gs = out[key_f].cpu().numpy(), out[key_s].cpu().numpy()
        pf = plain[key_f].cpu().numpy()
        assert gf.shape == (nq, 200) and (gf >= 0).all()
        for q in range(nq):
            n = int((pf[q] >= 0).sum())
            assert 0 < n < 200
            assert (gf[q][:n] == pf[q][:n]).all() and (gs[q][:n] == plain[key_s].cpu().numpy()[q][:n]).all()
            assert (gs[q][n:] == 0).all() and len(set(gf[q].tolist())) == 200
            if key_f == "flat_indices":
                assert (ref_score[q][gf[q][n:]] == 0).all()               # zero in the reference's tensor as well
            assert (gf[q][n:][1:] > gf[q][n:][:-1]).all()                 # the tail in ascending flat order (stable sort)
    # the driver: every list has max_before_nms rows
    d, cfg2, sd = load_golden("pipeline_video_only_h128")
    o = json.loads(str(d["opt"]))
    m2 = build_model(cfg2, sd)
    ds = GoldenDataset(d, m2.use_video, m2.use_sub)
    opt = argparse.Namespace(eval_context_bsz=o["eval_context_bsz"], eval_query_bsz=o["eval_query_bsz"],
                             device=torch.device(DEV), q2c_alpha=o["q2c_alpha"], min_pred_l=o["min_pred_l"],
                             max_pred_l=o["max_pred_l"], clip_length=o["clip_length"], debug=False,
                             external_inference_vr_res_path=None, max_ctx_l=cfg2["max_ctx_l"], pad_tail=True)
    with torch.no_grad():
        ctx = inf.compute_context_info(m2, ds, opt)
        res = inf.compute_query2ctx_info(m2, ds, opt, ctx, max_before_nms=800, max_n_videos=2, tasks=("SVMR", "VCMR"))
        opt.pad_tail = False
        short = inf.compute_query2ctx_info(m2, ds, opt, ctx, max_before_nms=800, max_n_videos=2, tasks=("SVMR", "VCMR"))
    for task in ("SVMR", "VCMR"):      # 30 clips: < 800 banded candidates in one video (SVMR), and in two for most queries
        assert [len(e["predictions"]) for e in res[task]] == [800] * ds.n_q
        assert all(e["predictions"][-1][3] == 0.0 for e in res["SVMR"])
        for a, b in zip(res[task], short[task]):
            assert a["predictions"][:len(b["predictions"])] == b["predictions"] and len(b["predictions"]) <= 800
    assert all(len(e["predictions"]) < 800 for e in short["SVMR"])


@pytest.mark.parametrize("dtype,hidden,nq", [(torch.float32, 128, 700), (torch.bfloat16, 768, 3000), (torch.float32, 768, 600)])
def test_packed_query_encode_equals_padded(dtype, hidden, nq, monkeypatch):
    """encode_query on the packed valid tokens (xml_attention_block_varlen / xml_modular_pool_varlen, no padding rows) gives
    the padded path's query vectors: projections and LayerNorms are row-wise, a padded key adds +0 to the softmax."""
    from tvretrieval_amd import model_xml
    m, cfg = _synthetic_model("video_sub", hidden, 256, 128, 128, 64, dtype, seed=2)
    rng = np.random.default_rng(4)
    lens = rng.integers(1, 31, nq)
    lens[0], lens[1] = 30, 1
    qf, qm = _feats(nq, lens, 128, 9)
    qf, qm = qf.to(DEV), qm.to(DEV)
    with torch.no_grad():
        monkeypatch.setattr(model_xml, "PACK_QUERY_TOKENS", False)
        v0, s0 = m.encode_query(qf, qm)
        monkeypatch.setattr(model_xml, "PACK_QUERY_TOKENS", True)
        assert m._encode_query_packed(qf, qm) is not None
        v1, s1 = m.encode_query(qf, qm)
        # a mask that is not a prefix of ones (or an empty query) keeps the padded path
        qm2 = qm.clone()
        qm2[0, 3] = 0
        assert m._encode_query_packed(qf, qm2) is None
        # the plan itself: cu_seqlens = running token counts, src_row = (sequence, token) of every packed row
        from tvretrieval_amd import ops
        cu, src, rows = ops.pack_plan(qm.float().contiguous())
        assert rows == int(lens.sum())
        want_cu = np.concatenate([[0], np.cumsum(lens)])
        assert np.array_equal(cu.cpu().numpy(), want_cu)
        want_src = np.concatenate([i * qm.shape[1] + np.arange(n_) for i, n_ in enumerate(lens)])
        assert np.array_equal(src[:rows].cpu().numpy(), want_src)
        for bad in (qm2, torch.zeros_like(qm), qm * 0.5):
            assert ops.pack_plan(bad.float().contiguous())[2] == -1
        # a caller that built the masks on the host hands the token count over: same plan, same vectors, no read-back
        cu2, src2, rows2 = ops.pack_plan(qm.float().contiguous(), rows=int(lens.sum()))
        assert rows2 == rows and torch.equal(cu2, cu) and torch.equal(src2[:rows], src[:rows])
        v2, s2 = m.encode_query(qf, qm, n_valid_tokens=int(lens.sum()))
        assert torch.equal(v2, v1) and torch.equal(s2, s1)
        with pytest.raises(ValueError, match="valid tokens"):
            ops.pack_plan(qm.float().contiguous(), rows=nq * 30 + 1)
    # f32: the same arithmetic per valid token.  bf16: the packed batch has fewer rows, so a projection may run on another
    # GEMM kernel of the family (LayerNorm in the epilogue or behind it: one rounding of the pre-LN value more or less) --
    # a couple of bf16 ulps on single elements, nothing systematic
    for a, b, nm in ((v1, v0, "video query"), (s1, s0, "sub query")):
        d_ = (a.float() - b.float()).abs()
        scale = max(1.0, float(b.float().abs().max()))
        if dtype == torch.float32:
            assert float(d_.max()) <= 2e-6 * scale, (nm, float(d_.max()))
        else:
            assert float(d_.max()) <= 0.012 * scale and float(d_.mean()) <= 1e-3 * scale, (nm, float(d_.max()), float(d_.mean()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ctx_mode,hidden", [("video_sub", 768), ("video", 256)])
def test_index_bits_do_not_depend_on_the_context_batch(dtype, ctx_mode, hidden):
    """The same corpus encoded in context batches of 2 048 / 200 / 37 videos gives a BITWISE equal CorpusIndex (feat1n, feat2,
    masks) in bf16 and f32: which kernel a projection takes is a property of its shape class, never of how many rows the
    batch holds (the LayerNorm-epilogue GEMM is a per-workgroup affair and runs from 2 048 rows on; every GEMM tiling
    accumulates over K in the same order; the attention kernels' per-video arithmetic does not depend on the grid).  An
    index built on 8 GPUs from 2 724-video shards therefore equals the single-GPU index of the same corpus bit for bit."""
    from tvretrieval_amd import inference as inf
    nv, l = 2100, 128
    m, cfg = _synthetic_model(ctx_mode, hidden, 512, 256, 256, l, dtype, seed=21)
    g = torch.Generator(device=DEV).manual_seed(5)
    lens = torch.randint(20, l + 1, (nv,), device=DEV, generator=g)
    lens[0] = l
    mask = (torch.arange(l, device=DEV)[None] < lens[:, None]).float()

    def feats(d):
        x = torch.nn.functional.normalize(torch.randn(nv, l, d, device=DEV, generator=g), dim=-1)
        return (x * mask[..., None]).contiguous()
    vf, sf = feats(512), feats(256)

    def batches(bs):
        for b in range(0, nv, bs):
            yield vf[b:b + bs], mask[b:b + bs], (sf[b:b + bs] if m.use_sub else None), (mask[b:b + bs] if m.use_sub else None)
    with torch.no_grad():
        ref = inf.build_corpus_index(m, batches(2048), l_ref=l)
        for bs in (200, 37):
            other = inf.build_corpus_index(m, batches(bs), l_ref=l)
            for mod in ref.modalities:
                for name, a_, b_ in (("feat1n", ref.feat1n_rows(mod), other.feat1n_rows(mod)), ("feat2", ref.feat2[mod], other.feat2[mod]),
                                     ("mask", ref.mask[mod], other.mask[mod])):
                    same = torch.equal(a_, b_)
                    assert same, "%s[%s]: context batches of %d videos differ from batches of 2048 in %d elements (max |d| %g)" % (
                        name, mod, bs, int((a_ != b_).sum()), float((a_.float() - b_.float()).abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_host_to_host_chunked_pass_equals_the_single_launch(dtype):
    _host_to_host_case(dtype, nq=700, chunk=256, want_chunks=[256, 444])


def test_host_to_host_large_chunks_take_the_packed_encoder_without_a_read_back():
    """Chunks of 768 / 1 632 queries (23 040 / 48 960 padded token rows: above model_xml.PACK_MIN_ROWS): the query encoder
    runs on the packed valid tokens, told its token count by the host (no read-back per chunk) -- still bitwise the single
    launch, which reads its plan back."""
    _host_to_host_case(torch.bfloat16, nq=2400, chunk=768, want_chunks=[768, 1632])


def test_host_to_host_with_an_empty_query():
    """A query without a token (row_start[i + 1] == row_start[i]): its mask row is all zero, the packed encoder does not apply,
    the chunk takes the padded path with its read-back -- same records as the single launch on the same padded batch."""
    _host_to_host_case(torch.float32, nq=600, chunk=256, want_chunks=[256, 344], empty=(0, 17, 300, 599))


def _host_to_host_case(dtype, nq, chunk, want_chunks, empty=()):
    """inference.vcmr_search_host (queries in pinned host memory -> chunked H2D on a side stream overlapped with the previous
    chunk's search -> K10 records -> one D2H) returns, bit for bit, the records of ONE vcmr_search over the whole query set:
    for the padded f32 layout of the reference's collate and for the feature store's ragged f16 token rows (device collate,
    xml_ingest_rows).  Growing chunks (256 + 444 queries; one K7 / K9 launch for all); run twice."""
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops
    from tvretrieval_amd.results import MOMENT_DTYPE
    nv, l = 150, 128
    m, cfg = _synthetic_model("video_sub", 128, 256, 128, 128, l, dtype, seed=15)
    rng = np.random.default_rng(8)
    lens = rng.integers(10, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    qlens = np.concatenate([[30], rng.integers(3, 31, nq - 1)])
    for i in empty:
        qlens[i] = 0
    meta2vid = torch.from_numpy((np.arange(nv) * 7 + 3).astype(np.int32)).to(DEV)
    kw = dict(max_vcmr_video=20, max_before_nms=60)
    with torch.no_grad():
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        # ragged f16 store rows (un-normalised) and the padded batch the device collate makes of them = the reference's input
        g = torch.Generator().manual_seed(4)
        rows16 = (torch.randn(int(qlens.sum()), 128, generator=g) * 2.0).to(torch.float16)
        row_start = torch.from_numpy(np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64))
        qf, qm = ops.ingest_rows(rows16.to(DEV), row_start.to(DEV), nq, 30, 30, normalize=True)
        one = inf.vcmr_search(m, index, qf, qm, **kw)
        rec1, cnt1 = ops.moments_decode(one["flat_scores"], flat=one["flat_indices"], top_idx=one["top_indices"],
                                        meta2vid=meta2vid, l_ref=index.l_ref, clip_length=1.5, seconds=True)
        want = rec1.cpu().numpy().view(MOMENT_DTYPE)[..., 0]
        want_cnt = cnt1.cpu().numpy()
        host = {"padded f32": dict(query_feat=qf.cpu().pin_memory(), query_mask=qm.cpu().pin_memory()),
                "ragged f16": dict(query_feat=rows16.pin_memory(), row_start=row_start.pin_memory(), lq=30)}
        for name, hk in host.items():
            tm = {}
            for rep in range(2):
                rec, cnt = inf.vcmr_search_host(m, index, meta2vid=meta2vid, chunk=chunk, clip_length=1.5, timings=tm, **hk, **kw)
                np.testing.assert_array_equal(cnt, want_cnt, err_msg=name)
                for col in ("vid", "st", "ed", "score"):
                    for q in range(nq):
                        np.testing.assert_array_equal(rec[col][q, :cnt[q]], want[col][q, :want_cnt[q]],
                                                      err_msg="%s: %s of query %d (pass %d)" % (name, col, q, rep))
            assert tm["chunks"] == len(want_chunks) and tm["chunk_queries"] == want_chunks and tm["h2d_s"] > 0 and tm["d2h_s"] > 0
            # two passes in flight (wait=False): each on its own buffer set, the same records
            pend = [inf.vcmr_search_host(m, index, meta2vid=meta2vid, chunk=chunk, clip_length=1.5, wait=False, **hk, **kw)
                    for _ in range(2)]
            assert pend[0].buffers is not pend[1].buffers
            for pnd in pend:
                rec, cnt = pnd.result()
                np.testing.assert_array_equal(cnt, want_cnt, err_msg=name + " (pipelined)")
                for col in ("vid", "st", "ed", "score"):
                    np.testing.assert_array_equal(np.where(np.arange(rec.shape[1])[None] < cnt[:, None], rec[col], 0),
                                                  np.where(np.arange(rec.shape[1])[None] < cnt[:, None], want[col], 0),
                                                  err_msg="%s: %s (pipelined)" % (name, col))
    assert (want_cnt[[i for i in range(nq) if i not in empty]] > 0).all() and len(np.unique(want["vid"][:, 0])) > 10


def test_hip_graph_replay_equals_eager():
    """GraphedVcmrSearch (one HIP graph per query-batch shape) returns exactly what the eager pass returns, for
    several different batches replayed through the same graph."""
    from tvretrieval_amd import inference as inf
    nv, nq, l = 40, 50, 128
    m, cfg = _synthetic_model("video_sub", 256, 512, 256, 256, l, torch.bfloat16, seed=7)
    rng = np.random.default_rng(5)
    lens = rng.integers(20, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 512, 1)
    sf, sm = _feats(nv, lens, 256, 2)
    with torch.no_grad():
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        g = inf.GraphedVcmrSearch(m, index, nq, 30, 256, max_vcmr_video=10, max_before_nms=100)
        for seed in (3, 4, 5):
            qf, qm = _feats(nq, np.concatenate([[30], rng.integers(5, 31, nq - 1)]), 256, seed)
            want = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=100)
            want = {k: v.clone() for k, v in want.items() if v is not None}
            got = g(qf.to(DEV), qm.to(DEV))
            for k in ("q2c", "top_scores", "top_indices", "flat_scores", "flat_indices"):
                assert torch.equal(got[k], want[k]), (seed, k)
    with pytest.raises(ValueError):
        g(qf[:10].to(DEV), qm[:10].to(DEV))


def test_hip_graph_large_batch_keeps_the_padded_query_path():
    """A batch large enough for the packed-token query encoder (nq * lq >= PACK_MIN_ROWS) can still be captured: the graph
    takes the padded path (the packing plan needs a host read-back and its launch shapes depend on the batch), and replays
    with DIFFERENT valid-token counts give the padded path's results."""
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import model_xml
    nv, l, lq = 24, 128, 30
    nq = model_xml.PACK_MIN_ROWS // lq + 40
    assert nq * lq >= model_xml.PACK_MIN_ROWS
    m, cfg = _synthetic_model("video_sub", 128, 256, 128, 128, l, torch.bfloat16, seed=8)
    rng = np.random.default_rng(6)
    lens = rng.integers(20, l + 1, nv); lens[0] = l
    vf, vm = _feats(nv, lens, 256, 1)
    sf, sm = _feats(nv, lens, 128, 2)
    with torch.no_grad():
        index = inf.build_corpus_index(m, [(vf.to(DEV), vm.to(DEV), sf.to(DEV), sm.to(DEV))])
        g = inf.GraphedVcmrSearch(m, index, nq, lq, 128, max_vcmr_video=10, max_before_nms=50)
        assert model_xml.PACK_QUERY_TOKENS            # restored after the capture
        for seed, lo in ((3, 5), (4, 25)):            # short and long queries: very different packed row counts
            qf, qm = _feats(nq, np.concatenate([[lq], rng.integers(lo, lq + 1, nq - 1)]), 128, seed)
            model_xml.PACK_QUERY_TOKENS = False
            try:
                want = inf.vcmr_search(m, index, qf.to(DEV), qm.to(DEV), max_vcmr_video=10, max_before_nms=50)
            finally:
                model_xml.PACK_QUERY_TOKENS = True
            want = {k: v.clone() for k, v in want.items() if v is not None}
            got = g(qf.to(DEV), qm.to(DEV))
            for k in ("q2c", "top_scores", "top_indices", "flat_scores", "flat_indices"):
                assert torch.equal(got[k], want[k]), (seed, k)


def test_full_scale_c3_pipeline_properties():
    """BASELINE configs[2] at FULL size (10 000 queries x 21 793 videos x 128 clips, H=768, video_sub, bf16) through
    the whole search pass, checked by size-independent properties and by sampled comparisons with the reference
    formulation (oracle on CPU for a few (query, video) pairs):
      K6/K8  top-100: scores descending, indices unique and in range, score == exp(20 * q2c[q, idx]);
             idempotence (top-k of the selected scores reproduces the list); sampled q2c entries vs fp32 recomputation;
      K7     start / end rows are probability vectors; sampled rows vs conv1d + softmax on CPU;
      K9     top-200: scores descending, flat indices unique, inside the length band, score == st*w*ed recomputed."""
    import bench
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd import ops as hops
    from tvretrieval_amd.model_xml import XML
    nq, nv, l, hidden, dv, ds_, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
    torch.manual_seed(0)
    cfg = bench.model_config(hidden, dv, ds_, dq, ctx_mode, l)
    m = XML(cfg, compute_dtype=torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        index = inf.build_corpus_index(m, bench.context_batches(0, nv, l, dv, ds_, True, True, torch.device(DEV)),
                                       n_total=nv, l_ref=l)
        qf, qm = bench.synth_queries(nq, dq, torch.device(DEV))
        qvec = inf.stage_query_vectors(m, qf, qm)
        out = inf.vcmr_search(m, index, qf, qm)
        st, ed = inf.stage_span_probs(m, index, qvec, out["top_indices"])
    q2c, tw, ti = out["q2c"], out["top_scores"], out["top_indices"].long()
    fs, fi = out["flat_scores"], out["flat_indices"].long()
    assert q2c.shape == (nq, nv) and tw.shape == (nq, 100) and fs.shape == (nq, 200)
    # ---- K6 / K8
    assert bool(torch.isfinite(q2c).all()) and float(q2c.abs().max()) <= 1.0 + 1e-3          # mean of two cosines
    assert bool((tw[:, :-1] >= tw[:, 1:]).all())
    assert int(ti.min()) >= 0 and int(ti.max()) < nv
    assert bool((torch.sort(ti, dim=1)[0][:, 1:] != torch.sort(ti, dim=1)[0][:, :-1]).all())     # unique per row
    sel = torch.gather(q2c, 1, ti)
    assert torch.allclose(tw, torch.exp(20.0 * sel), rtol=1e-5, atol=0)
    kth = sel[:, -1:]
    assert int((q2c > kth).sum(1).max()) <= 99                      # nothing outside the list beats its last entry
    tw2, pos2 = hops.topk_rows(sel.contiguous(), 100, alpha=20.0)
    assert torch.equal(tw2, tw) and bool((pos2.long() == torch.arange(100, device=DEV)[None]).all())   # idempotent
    g = torch.Generator().manual_seed(1)
    qs = torch.randint(0, nq, (6,), generator=g).tolist()
    f1rows = {mod: index.feat1n_rows(mod) for mod in index.modalities}   # un-tiled K6 operand
    for q in qs:                                                     # sampled exact recomputation, fp32 on the GPU
        vs = ti[q, :8]
        want = 0
        for mod in index.modalities:
            qn = torch.nn.functional.normalize(qvec[mod][q].float(), dim=-1)
            c = f1rows[mod][vs].float()
            want = want + torch.einsum("d,vld->vl", qn, c).max(1)[0]
        want = want / len(index.modalities)
        # bf16 operands (incl. the normalised query rounded to bf16 inside the kernel path): 3 significant digits
        assert float((q2c[q, vs] - want).abs().max()) < 4e-3
    # ---- K7
    assert st.shape == (nq, 100, index.lpad)
    assert torch.allclose(st.sum(-1), torch.ones(nq, 100, device=DEV), atol=2e-3)
    assert torch.allclose(ed.sum(-1), torch.ones(nq, 100, device=DEV), atol=2e-3)
    assert float(st.min()) >= 0 and float(ed.min()) >= 0
    wst = m.merged_st_predictor.weight.detach().float().cpu()
    wed = m.merged_ed_predictor.weight.detach().float().cpu()
    for q in qs[:3]:
        vs = ti[q, :5]
        sims = 0
        for mod in index.modalities:
            lin = getattr(m, mod + "_query_linear")
            ql = torch.nn.functional.linear(qvec[mod][q].float().cpu(), lin.weight.detach().float().cpu(),
                                            lin.bias.detach().float().cpu())
            sims = sims + torch.einsum("d,vld->vl", ql, index.feat2[mod][vs].float().cpu())
        sims = (sims / 2).unsqueeze(1)
        want_st = torch.softmax(torch.nn.functional.conv1d(sims, wst, padding=2).squeeze(1), -1)
        want_ed = torch.softmax(torch.nn.functional.conv1d(sims, wed, padding=2).squeeze(1), -1)
        # bf16 storage of the projected query (the CPU side keeps f32): logits of O(10) move by ~1e-2, peaked
        # softmax probabilities by a few 1e-3
        assert float((st[q, :5, :l].cpu() - want_st).abs().max()) < 1e-2
        assert float((ed[q, :5, :l].cpu() - want_ed).abs().max()) < 1e-2
    # ---- K9
    assert bool((fs[:, :-1] >= fs[:, 1:]).all()) and float(fs.min()) >= 0
    valid = fi >= 0
    assert bool(valid[:, 0].all())
    r = fi // (l * l)
    i = (fi // l) % l
    j = fi % l
    assert bool(((r < 100) & (r >= 0))[valid].all())
    d = (j - i)[valid]
    assert int(d.min()) >= 2 and int(d.max()) < 16                              # min_pred_l <= ed - st < max_pred_l
    srt = torch.sort(torch.where(valid, fi, -1 - torch.arange(200, device=DEV)[None].expand_as(fi)), dim=1)[0]
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                              # unique moments per query
    ar = torch.arange(nq, device=DEV)[:, None].expand_as(fi)
    rr, ii, jj = r.clamp(0, 99), i.clamp(0, l - 1), j.clamp(0, l - 1)
    rec = (st[ar, rr, ii] * tw[ar, rr]) * ed[ar, rr, jj]                        # (st * w) * ed, the reference's order
    assert torch.allclose(fs[valid], rec[valid], rtol=2e-6, atol=0)
    # the best moment of every query can not be beaten by any banded product of its best-weighted video
    best_v0 = ((st[:, 0, :l, None] * tw[:, 0, None, None]) * ed[:, 0, None, :l])
    band = torch.from_numpy(O.min_max_length_mask(l, 2, 16)).to(DEV)
    assert bool((fs[:, 0] >= (best_v0 * band).amax((1, 2)) * (1 - 1e-6)).all())


def test_preallocated_index_equals_list_and_cat_index():
    """build_corpus_index(n_videos=...) (rows written into preallocated index tensors) holds exactly what the list + cat path
    holds, ragged batches included."""
    from tvretrieval_amd import inference as inf
    nv, l = 37, 64
    m, cfg = _synthetic_model("video_sub", 128, 96, 64, 64, l, torch.bfloat16, seed=3)
    rng = np.random.default_rng(8)
    lens = rng.integers(5, l + 1, nv); lens[3] = l
    vf, vm = _feats(nv, lens, 96, 1)
    sf, sm = _feats(nv, lens, 64, 2)

    def batches():      # each batch padded to its OWN maximum, like the reference's collate
        for b in range(0, nv, 10):
            lb = int(lens[b:b + 10].max())
            yield vf[b:b + 10, :lb].contiguous().to(DEV), vm[b:b + 10, :lb].contiguous().to(DEV), \
                sf[b:b + 10, :lb].contiguous().to(DEV), sm[b:b + 10, :lb].contiguous().to(DEV)
    with torch.no_grad():
        a = inf.build_corpus_index(m, batches(), l_ref=l)
        b = inf.build_corpus_index(m, batches(), l_ref=l, n_videos=nv)
    for mod in a.modalities:
        assert torch.equal(a.feat1n_rows(mod), b.feat1n_rows(mod))
        assert torch.equal(a.feat2[mod], b.feat2[mod]) and torch.equal(a.mask[mod], b.mask[mod])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_profile_main_buckets_and_submodule_forwards(dtype):
    """The reference's own profiler (baselines/profiling/profile_main.py:128-209) drives the model through FOUR buckets and
    touches sub-modules directly -- encode_input with model.query_input_proj / query_encoder / query_pos_embed,
    model.video_query_linear(...), model.merged_st_predictor(similarity).  The same four call sequences here (written fresh,
    ctx_mode video_sub_tef like ProfileXML) against the oracle, and every holder module's own forward() -- LinearLayer,
    TrainablePositionalEncoding, BertSelfAttention, BertSelfOutput (xml/model_components.py:76-89,156-163,266-303,313-317) --
    against the oracle's restatement of it."""
    f32 = dtype == torch.float32
    tol = dict(atol=2e-4, rtol=2e-4) if f32 else dict(atol=0.12, rtol=3e-2)
    l, h, nv, nq = 24, 128, 6, 9
    m, cfg = _synthetic_model("video_sub", h, 66, 34, 40, l, dtype, seed=21)     # 64 + 2 / 32 + 2 TEF-style widths
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    om = O.OracleXML(cfg, sd)
    w = om.w
    vf, vm = _feats(nv, [l, 20, 7, 24, 13, 1], 66, 31)
    sf, _ = _feats(nv, [l, 20, 7, 24, 13, 1], 34, 32)
    qf, qm = _feats(nq, [5, 30, 17, 9, 30, 12, 8, 21, 3], 40, 33)

    def chk(name, got, want):
        close(name, got, want, tol["atol"], tol["rtol"])
    with torch.no_grad():
        # bucket 1: context encoding
        got = m.cross_encode_context(vf.to(DEV), vm.to(DEV), sf.to(DEV), vm.to(DEV))
        want = om.cross_encode_context(vf, vm, sf, vm)
        valid = vm[..., None] > 0                    # (padded rows of cross-attended streams: see test_golden_pipeline_fp32)
        for n_, g_, w_ in zip(("vf1", "vf2", "sf1", "sf2"), got, want):
            chk("bucket 1 " + n_, torch.where(valid, g_.float().cpu(), torch.zeros(())), torch.where(valid, w_, torch.zeros(())))
        # bucket 2: query encoding through the sub-modules the profiler names
        enc = m.encode_input(qf.to(DEV), qm.to(DEV), m.query_input_proj, m.query_encoder, m.query_pos_embed)
        vq, sq = m.get_modularized_queries(enc, qm.to(DEV), return_modular_att=False)
        vql, sql = m.video_query_linear(vq), m.sub_query_linear(sq)
        o_enc = om.encode_input(qf, qm, "query_input_proj", "query_encoder", "query_pos_embed")
        o_vq, o_sq = om.get_modularized_queries(o_enc, qm)
        chk("bucket 2 video_query", vq, o_vq)
        chk("bucket 2 sub_query", sq, o_sq)
        lin = lambda x, n_: torch.nn.functional.linear(x, sd[n_ + ".weight"], sd[n_ + ".bias"])      # noqa: E731
        chk("bucket 2 video_query_linear", vql, lin(o_vq, "video_query_linear"))
        chk("bucket 2 sub_query_linear", sql, lin(o_sq, "sub_query_linear"))
        # bucket 3: retrieval (one modality, the profiler doubles the time)
        chk("bucket 3 get_video_level_scores", m.get_video_level_scores(vq, got[0], vm.to(DEV)),
            om.get_video_level_scores(o_vq, want[0], vm))
        # bucket 4: span prediction on a similarity tensor the caller computed (torch einsum: plumbing, like the profiler's)
        sim = torch.einsum("md,nld->mnl", lin(o_vq, "video_query_linear"), want[1])
        sim = ((sim + sim) / 2).reshape(nq * nv, 1, l)
        for name in ("merged_st_predictor", "merged_ed_predictor"):
            g_ = getattr(m, name)(sim.to(DEV)).view(nq, nv, l)
            w_ = torch.nn.functional.conv1d(sim, sd[name + ".weight"], padding=2).view(nq, nv, l)
            close("bucket 4 " + name, g_, w_, 1e-5, 1e-5)
            close("bucket 4 mask_logits", O.mask_logits(g_.cpu(), vm), O.mask_logits(w_, vm), 1e-5, 1e-5)
        # ---- the holder modules' own forward() -----------------------------------------------------------------------
        x_in = vf.to(DEV)
        chk("LinearLayer.forward", m.video_input_proj(x_in), O.linear_layer(vf, w.sub("video_input_proj")))
        hx = O.linear_layer(vf, w.sub("video_input_proj"))
        chk("TrainablePositionalEncoding.forward", m.ctx_pos_embed(hx.to(DEV)), O.trainable_pos_enc(hx, w.sub("ctx_pos_embed")))
        px = O.trainable_pos_enc(hx, w.sub("ctx_pos_embed"))
        sa = m.video_encoder1.self
        g_ = sa(px.to(DEV), px.to(DEV), px.to(DEV), vm.unsqueeze(1).to(DEV))                       # (N, 1, L): key mask
        w_ = O.bert_self_attention(px, px, px, vm.unsqueeze(1), w.sub("video_encoder1.self"), 4)
        chk("BertSelfAttention.forward (self)", g_, w_)
        side = O.trainable_pos_enc(O.linear_layer(sf, w.sub("sub_input_proj")), w.sub("ctx_pos_embed"))
        cm = torch.einsum("bm,bn->bmn", vm, vm)                                                    # cross attention's mask
        g_ = m.video_cross_att(px.to(DEV), side.to(DEV), side.to(DEV), cm.to(DEV)).float().cpu()
        w_ = O.bert_self_attention(px, side, side, cm, w.sub("video_cross_att"), 4)
        chk("BertSelfAttention.forward (cross, valid query rows)", torch.where(valid, g_, torch.zeros(())),
            torch.where(valid, w_, torch.zeros(())))
        chk("BertSelfOutput.forward", m.video_encoder1.output(w_.to(DEV), px.to(DEV)),
            O.bert_self_output(w_, px, w.sub("video_encoder1.output")))
        with pytest.raises(ValueError, match="outer product"):
            bad = cm.clone()
            bad[0, 0, 1] = 0
            m.video_cross_att(px.to(DEV), side.to(DEV), side.to(DEV), bad.to(DEV))
