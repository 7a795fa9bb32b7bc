"""The CPU oracle (oracle/xml_oracle.py) replayed against vectors captured from the reference itself
(tools/make_golden.py).  CPU only; this is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import xml_oracle as O

MODEL_CASES = ["xml_video_sub_cross_h128", "xml_video_only_h256", "xml_sub_only_h128",
               "xml_video_sub_nocross_nomerge_h128"]
TOL = dict(rtol=2e-5, atol=2e-5)


def _close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), **tol)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_encoders_and_scores(name):
    d, cfg, sd = load_golden(name)
    m = O.OracleXML(cfg, sd)
    with torch.no_grad():
        v1, v2, s1, s2 = m.encode_context(d["video_feat"], d["video_mask"], d["sub_feat"], d["sub_mask"])
        for k, t in (("vf1", v1), ("vf2", v2), ("sf1", s1), ("sf2", s2)):
            if k in d:
                _close(t, d[k])       # includes padded rows
            else:
                assert t is None
        vq, sq = m.encode_query(d["query_feat"], d["query_mask"])
        _close(vq, d["video_query"])
        _close(sq, d["sub_query"])
        vm = d["video_mask"] if m.use_video else None
        sm = d["sub_mask"] if m.use_sub else None
        q2c, st, ed = m.get_pred_from_raw_query(d["query_feat"], d["query_mask"], v1, v2, vm, s1, s2, sm, cross=True)
        _close(q2c, d["q2c_cross"])
        _close(st, d["st_cross"], rtol=1e-5, atol=1e-4)
        _close(ed, d["ed_cross"], rtol=1e-5, atol=1e-4)
        n = d["q2c_pair"].shape[0]
        sel = lambda t: None if t is None else t[:n]
        q2c_p, st_p, ed_p = m.get_pred_from_raw_query(
            d["query_feat"][:n], d["query_mask"][:n], sel(v1), sel(v2), sel(vm), sel(s1), sel(s2), sel(sm), cross=False)
        _close(q2c_p, d["q2c_pair"])
        _close(st_p, d["st_pair"], rtol=1e-5, atol=1e-4)
        _close(ed_p, d["ed_pair"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_tail(name):
    d, cfg, sd = load_golden(name)
    alpha, kvid, min_l, max_l, nbefore = d["tail_params"]
    t = O.vcmr_tail(torch.from_numpy(d["q2c_cross"]), torch.from_numpy(d["st_cross"]), torch.from_numpy(d["ed_cross"]),
                    q2c_alpha=float(alpha), max_vcmr_video=int(kvid), min_pred_l=int(min_l), max_pred_l=int(max_l),
                    max_before_nms=int(nbefore))
    np.testing.assert_array_equal(t["top_indices"].numpy(), d["top_indices"])
    np.testing.assert_array_equal(t["top_scores"].numpy(), d["top_scores"])
    np.testing.assert_array_equal(t["flat_scores"].numpy(), d["flat_scores"])
    pos = d["flat_scores"] > 0          # ordering among exact zeros is unspecified
    np.testing.assert_array_equal(t["flat_indices"].numpy()[pos], d["flat_indices"][pos])


def _pipeline_inputs(d, cfg):
    n_v = len(d["ctx_lens"])
    vids = [dict(video=d["video_feat/%d" % i], sub=d["sub_feat/%d" % i]) for i in range(n_v)]
    n_q = len(d["query_gt_video"])
    qs = [d["query_feat/%d" % i] for i in range(n_q)]
    return vids, qs


def _pad(seqs):
    l = max(len(s) for s in seqs)
    out = np.zeros((len(seqs), l, seqs[0].shape[1]), np.float32)
    m = np.zeros((len(seqs), l), np.float32)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
        m[i, :len(s)] = 1
    return out, m


@pytest.mark.parametrize("name", ["pipeline_video_sub_h128", "pipeline_video_only_h128"])
def test_pipeline(name):
    """Replays the driver (xml/inference.py:32-97,252-445) through the oracle: context batches padded to
    the batch max and zero-filled beyond (cat_tensor), query batches, VCMR/SVMR/VR lists, NMS."""
    d, cfg, sd = load_golden(name)
    opt = json.loads(str(d["opt"]))
    m = O.OracleXML(cfg, sd)
    vids, qs = _pipeline_inputs(d, cfg)
    bs = opt["eval_context_bsz"]
    f = dict(v1=[], v2=[], s1=[], s2=[], vm=[], sm=[])
    with torch.no_grad():
        for b in range(0, len(vids), bs):
            vf, vm = _pad([v["video"] for v in vids[b:b + bs]])
            sf, sm = _pad([v["sub"] for v in vids[b:b + bs]])
            v1, v2, s1, s2 = m.encode_context(vf, vm, sf, sm)
            if m.use_video:
                f["v1"].append(v1), f["v2"].append(v2), f["vm"].append(torch.from_numpy(vm))
            if m.use_sub:
                f["s1"].append(s1), f["s2"].append(s2), f["sm"].append(torch.from_numpy(sm))
        ctx = {k: O.cat_pad_context(v) for k, v in f.items()}
        for k, g in (("v1", "video_feat1"), ("v2", "video_feat2"), ("vm", "video_mask"),
                     ("s1", "sub_feat1"), ("s2", "sub_feat2"), ("sm", "sub_mask")):
            if ("ctx/" + g) in d:
                _close(ctx[k], d["ctx/" + g])
        qb = opt["eval_query_bsz"]
        vcmr, vr, svmr = [], [], []
        for b in range(0, len(qs), qb):
            qf, qm = _pad(qs[b:b + qb])
            q2c, st, ed = m.get_pred_from_raw_query(qf, qm, ctx["v1"], ctx["v2"], ctx["vm"], ctx["s1"], ctx["s2"],
                                                    ctx["sm"], cross=True)
            t = O.vcmr_tail(q2c, st, ed, q2c_alpha=opt["q2c_alpha"], max_vcmr_video=opt["max_vcmr_video"],
                            min_pred_l=opt["min_pred_l"], max_pred_l=opt["max_pred_l"],
                            max_before_nms=opt["max_before_nms"])
            vid, st_s, ed_s = O.unravel_moments(t["flat_indices"].numpy(), t["top_indices"].numpy(), t["ctx_l"],
                                                opt["clip_length"])
            gt = d["query_gt_video"][b:b + qb]
            rows = np.arange(len(gt))
            sv = O.svmr_tail(t["st_probs"].numpy()[rows, gt], t["ed_probs"].numpy()[rows, gt], opt["min_pred_l"],
                             opt["max_pred_l"], opt["max_before_nms"])
            for i in range(len(gt)):
                vcmr.append(np.stack([d["video_idx"][vid[i]], st_s[i], ed_s[i], t["flat_scores"].numpy()[i]], 1))
                vr.append(np.stack([d["video_idx"][t["top_indices"].numpy()[i]], 0 * t["top_scores"].numpy()[i],
                                    0 * t["top_scores"].numpy()[i], t["top_scores"].numpy()[i]], 1))
                s = sv[i].copy()
                svmr.append(np.stack([np.full(len(s), d["video_idx"][gt[i]]), s[:, 0] * opt["clip_length"],
                                      (s[:, 1] + 1) * opt["clip_length"], s[:, 2]], 1))
    for got, want in ((vcmr, d["res/VCMR"]), (vr, d["res/VR"]), (svmr, d["res/SVMR"])):
        got = np.stack(got)
        _close(got[..., 3], want[..., 3], rtol=1e-5, atol=1e-9)
        pos = want[..., 3] > 0
        np.testing.assert_array_equal(got[..., :3][pos], want[..., :3][pos])
    # NMS rows
    thd = opt["nms_thd"]
    for i in range(len(qs)):
        want = d["nms/VCMR/%d" % i]
        got = np.array(O.vcmr_nms(d["res/VCMR"][i].tolist(), thd, opt["max_before_nms"], 100)).reshape(-1, 4)
        np.testing.assert_allclose(got, want)
        want = d["nms/SVMR/%d" % i]
        preds = [list(p[1:]) for p in d["res/SVMR"][i].tolist()[:opt["max_before_nms"]]]
        got = np.array([[d["res/SVMR"][i][0][0]] + p for p in O.temporal_nms(preds, thd)[:100]]).reshape(-1, 4)
        np.testing.assert_allclose(got, want)


def test_pipeline_external_vr():
    """The external-VR branch of the driver (xml/inference.py:244-249,264-273,349-355) through the oracle, against the
    lists the reference produced with opt.external_inference_vr_res_path set (tools/make_golden.py::gen_external_vr_case):
    the first max_vcmr_video entries of another model's VR submission replace top-k, their weights are exp(alpha * s)."""
    d, cfg, sd = load_golden("pipeline_external_vr_h128")
    opt = json.loads(str(d["opt"]))
    m = O.OracleXML(cfg, sd)
    vids, qs = _pipeline_inputs(d, cfg)
    bs, kv = opt["eval_context_bsz"], opt["max_vcmr_video"]
    idx2meta = {int(v): i for i, v in enumerate(d["video_idx"])}
    f = dict(v1=[], v2=[], s1=[], s2=[], vm=[], sm=[])
    with torch.no_grad():
        for b in range(0, len(vids), bs):
            vf, vm = _pad([v["video"] for v in vids[b:b + bs]])
            sf, sm = _pad([v["sub"] for v in vids[b:b + bs]])
            v1, v2, s1, s2 = m.encode_context(vf, vm, sf, sm)
            f["v1"].append(v1), f["v2"].append(v2), f["vm"].append(torch.from_numpy(vm))
            f["s1"].append(s1), f["s2"].append(s2), f["sm"].append(torch.from_numpy(sm))
        ctx = {k: O.cat_pad_context(v) for k, v in f.items()}
        qb = opt["eval_query_bsz"]
        vcmr, vr = [], []
        for b in range(0, len(qs), qb):
            qf, qm = _pad(qs[b:b + qb])
            q2c, st, ed = m.get_pred_from_raw_query(qf, qm, ctx["v1"], ctx["v2"], ctx["vm"], ctx["s1"], ctx["s2"],
                                                    ctx["sm"], cross=True)
            ext_i = np.array([[idx2meta[int(v)] for v in row[:kv]] for row in d["ext/video_idx"][b:b + qb]])
            ext_s = d["ext/score"][b:b + qb, :kv].astype(np.float32)
            t = O.vcmr_tail(q2c, st, ed, q2c_alpha=opt["q2c_alpha"], max_vcmr_video=kv, min_pred_l=opt["min_pred_l"],
                            max_pred_l=opt["max_pred_l"], max_before_nms=opt["max_before_nms"],
                            external_top=(ext_i, ext_s))
            vid, st_s, ed_s = O.unravel_moments(t["flat_indices"].numpy(), t["top_indices"].numpy(), t["ctx_l"],
                                                opt["clip_length"])
            for i in range(len(qf)):
                vcmr.append(np.stack([d["video_idx"][vid[i]], st_s[i], ed_s[i], t["flat_scores"].numpy()[i]], 1))
                tw = t["top_scores"].numpy()[i]
                vr.append(np.stack([d["video_idx"][t["top_indices"].numpy()[i]], 0 * tw, 0 * tw, tw], 1))
    for got, want in ((vcmr, d["res/VCMR"]), (vr, d["res/VR"])):
        got = np.stack(got)
        _close(got[..., 3], want[..., 3], rtol=1e-5, atol=1e-9)
        pos = want[..., 3] > 0
        np.testing.assert_array_equal(got[..., :3][pos], want[..., :3][pos])
    # the VR list IS the external one, trimmed to max_vcmr_video, with exp(alpha * s) as score
    np.testing.assert_array_equal(d["res/VR"][..., 0], d["ext/video_idx"][:, :kv])
    np.testing.assert_allclose(d["res/VR"][..., 3], np.exp(opt["q2c_alpha"] * d["ext/score"][:, :kv]), rtol=1e-5)    # (f32 exp)


NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight"]     # xml/train.py:355-362


@pytest.mark.parametrize("name", ["train_step_video_sub_h128", "train_step_nocross_lse_h128",
                                  "train_step_staged_video_sub_h128"])
def test_training_steps(name):
    """loss / gradients of step 1 and the parameters after the BertAdam steps of the reference (three; the staged case
    runs two steps without the span loss and three with it: tensors that have never received a gradient are skipped and
    every tensor counts its own schedule steps)."""
    d, cfg, _ = load_golden(name)
    sd = {k[len("sd_before/"):]: v for k, v in d.items() if k.startswith("sd_before/")}
    params = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    okw = json.loads(str(d["optim"]))
    wd = {k: (0.0 if any(nd in k for nd in NO_DECAY) else okw["weight_decay"]) for k in params}
    state = {}
    sched = d["lw_st_ed_schedule"] if "lw_st_ed_schedule" in d else None
    ever, steps = set(), {k: 0 for k in params}
    for it in range(3 if sched is None else len(sched)):
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        if sched is not None:
            cfg = dict(cfg, lw_st_ed=float(sched[it]))
        m = O.OracleXML(cfg, leaf)
        loss, parts = m.forward_loss(d["query_feat"], d["query_mask"], d["video_feat"], d["video_mask"],
                                     d["sub_feat"], d["sub_mask"], d["st_ed_indices"], d["neg_ctx_rank_steps"][it],
                                     d["neg_q_rank_steps"][it])
        assert abs(float(loss) - float(d["step_losses"][it])) < 2e-5
        loss.backward()
        if it == 0:
            assert abs(float(loss) - float(d["loss"])) < 1e-5
            assert abs(parts["loss_st_ed"] - float(d["loss_st_ed"])) < 1e-5
            assert abs(parts["loss_neg_ctx"] - float(d["loss_neg_ctx"])) < 1e-5
            assert abs(parts["loss_neg_q"] - float(d["loss_neg_q"])) < 1e-5
            for k, p in leaf.items():
                if ("grad/" + k) in d:
                    _close(p.grad, d["grad/" + k], rtol=1e-4, atol=1e-6)
        with torch.no_grad():
            grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in leaf.items()}
            ever |= {k for k in params if leaf[k].grad is not None}
            # staged case: torch 1.4 zero_grad semantics -- a tensor that has EVER had a gradient keeps stepping
            trainable = {k: params[k] for k in params if (leaf[k].grad is not None if sched is None else k in ever)}
            O.bert_adam_step(trainable, grads, state, steps, wd, **{k: v for k, v in okw.items() if k != "weight_decay"})
            for k in trainable:
                steps[k] += 1
    if sched is not None:
        assert steps == json.loads(str(d["final_steps"]))
        assert sorted(set(steps.values())) == [3, 5]
    for k, p in params.items():
        _close(p, d["sd_after3/" + k], rtol=1e-4, atol=2e-6)
