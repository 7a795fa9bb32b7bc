"""CPU tests of the "next" rows: host NMS behind the C ABI and the vectorised evaluator, against
outputs of the reference captured by tools/make_golden.py."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden


@pytest.mark.parametrize("name", ["pipeline_video_sub_h128", "pipeline_video_only_h128"])
def test_nms_matches_reference(name):
    from tvretrieval_amd import postproc
    d, cfg, sd = load_golden(name)
    opt = json.loads(str(d["opt"]))
    n_q = d["res/VCMR"].shape[0]
    vcmr = [dict(desc_id=i, desc="", predictions=[list(p) for p in d["res/VCMR"][i].tolist()]) for i in range(n_q)]
    svmr = [dict(desc_id=i, desc="", predictions=[list(p) for p in d["res/SVMR"][i].tolist()]) for i in range(n_q)]
    got_v = postproc.post_processing_vcmr_nms(copy.deepcopy(vcmr), nms_thd=opt["nms_thd"],
                                              max_before_nms=opt["max_before_nms"], max_after_nms=100)
    got_s = postproc.post_processing_svmr_nms(copy.deepcopy(svmr), nms_thd=opt["nms_thd"],
                                              max_before_nms=opt["max_before_nms"], max_after_nms=100)
    for i in range(n_q):
        np.testing.assert_array_equal(np.array(got_v[i]["predictions"]).reshape(-1, 4), d["nms/VCMR/%d" % i])
        np.testing.assert_array_equal(np.array(got_s[i]["predictions"]).reshape(-1, 4), d["nms/SVMR/%d" % i])


def test_nms_edge_cases():
    from tvretrieval_amd import postproc
    assert postproc.temporal_non_maximum_suppression([], 0.5) == []
    one = [[1.0, 2.0, 0.3]]
    assert postproc.temporal_non_maximum_suppression(one, 0.5) == one
    same = [[0.0, 3.0, 0.9], [0.0, 3.0, 0.8], [10.0, 12.0, 0.1], [0.5, 3.0, 0.85]]
    assert postproc.temporal_non_maximum_suppression(same, 0.5) == [[0.0, 3.0, 0.9], [10.0, 12.0, 0.1]]
    zero_len = [[1.0, 1.0, 0.5], [1.0, 1.0, 0.4]]       # union == 0 -> IoU 0 -> both kept
    assert postproc.temporal_non_maximum_suppression(zero_len, 0.5) == zero_len
    many = [[float(i), float(i) + 0.5, 1.0 / (i + 1)] for i in range(150)]
    assert len(postproc.temporal_non_maximum_suppression(many, 0.5, max_after_nms=100)) == 100
    assert postproc.filter_vcmr_by_nms([], 0.5) == []
    top = postproc.get_submission_top_n(dict(video2idx={}, VR=[dict(predictions=list(range(7)))]), top_n=3)
    assert top["VR"][0]["predictions"] == [0, 1, 2]


@pytest.mark.parametrize("name", ["eval_tvr_style", "eval_didemo_style", "eval_more_tiny", "eval_more_large",
                                  "eval_more_vcmr_only", "eval_more_svmr_vr_didemo"])
def test_evaluator_matches_reference(name):
    from tvretrieval_amd import evaluate
    case = json.load(open(os.path.join(GOLDEN, name + ".json")))
    got = evaluate.eval_retrieval(case["submission"], case["ground_truth"], iou_thds=(0.5, 0.7), verbose=False,
                                  match_number=True, use_desc_type=case["use_desc_type"])
    assert json.loads(json.dumps(got)) == case["metrics"]


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_nms_vs_oracle(seed):
    """Host NMS (C++ behind the ABI) against the oracle's restatement of utils/temporal_nms.py on random prediction lists:
    heavy overlaps, exact score ties (the reference's stable sort order), zero-length spans, more survivors than the cap,
    many videos per query."""
    from oracle import xml_oracle as O
    from tvretrieval_amd import postproc
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(1, 400))
    st = np.round(rng.uniform(0, 60, n) / 1.5) * 1.5
    ln = np.round(rng.uniform(0, 24, n) / 1.5) * 1.5 * (rng.random(n) > 0.05)          # a few zero-length spans
    sc = np.round(rng.random(n), int(rng.choice([2, 3, 8])))                            # coarse scores: exact ties
    preds = [[float(a), float(a + b), float(c)] for a, b, c in zip(st, ln, sc)]
    thd = float(rng.choice([0.3, 0.5, 0.7]))
    cap = int(rng.choice([5, 100]))
    assert postproc.temporal_non_maximum_suppression([list(p) for p in preds], thd, max_after_nms=cap) == \
        O.temporal_nms([list(p) for p in preds], thd, max_after_nms=cap)
    # VCMR: [video, st, ed, score] rows, sorted by score as the search emits them
    vids = rng.integers(0, int(rng.integers(1, 12)), n)
    rows = sorted(([int(v)] + p for v, p in zip(vids, preds)), key=lambda r: -r[3])
    mb = int(rng.choice([50, 1000]))
    got = postproc.filter_vcmr_by_nms([list(r) for r in rows], thd, max_before_nms=mb, max_after_nms=cap)
    want = O.vcmr_nms([list(r) for r in rows], thd, max_before_nms=mb, max_after_nms=cap)
    assert [list(map(float, r)) for r in got] == [list(map(float, r)) for r in want]


def _random_result_set(rng, nq, n, n_vid=9, clip=1.5):
    """(Nq, n) columns shaped like the engine's VCMR output: scores descending per row, coarse enough for exact ties,
    ragged counts including empty and single-entry rows."""
    from tvretrieval_amd.results import MomentResults
    vid = rng.integers(0, n_vid, (nq, n))
    st = np.round(rng.uniform(0, 60, (nq, n)) / clip) * clip
    ed = st + np.round(rng.uniform(0, 24, (nq, n)) / clip) * clip * (rng.random((nq, n)) > 0.05)
    sc = -np.sort(-np.round(rng.random((nq, n)), 3), axis=1)
    cnt = rng.integers(0, n + 1, nq).astype(np.int32)
    cnt[:3] = [0, 1, n][:min(3, nq)]
    return MomentResults(list(range(100, 100 + nq)), ["q%d" % i for i in range(nq)], vid, st, ed, sc, cnt)


@pytest.mark.parametrize("seed", range(4))
def test_batched_nms_equals_per_query_oracle(seed):
    """xml_nms_{vcmr,svmr}_batched_host on (Nq, n) arrays == the oracle's restatement of the reference's per-query NMS,
    for the array form (MomentResults in -> MomentResults out) and the list form (the reference's dicts, mutated)."""
    from oracle import xml_oracle as O
    from tvretrieval_amd import postproc
    rng = np.random.default_rng(900 + seed)
    nq, n = int(rng.integers(70, 200)), int(rng.integers(5, 120))
    res = _random_result_set(rng, nq, n)
    thd = float(rng.choice([0.3, 0.5, 0.7]))
    mb, ma = int(rng.choice([40, 1000])), int(rng.choice([7, 100]))
    lists = res.to_list()
    assert [len(e["predictions"]) for e in lists] == res.count.tolist()
    assert all(type(p[0]) is int and type(p[1]) is float and type(p[3]) is float for e in lists for p in e["predictions"])
    want_v = [O.vcmr_nms([list(p) for p in e["predictions"]], thd, max_before_nms=mb, max_after_nms=ma) for e in lists]
    got = postproc.post_processing_vcmr_nms(res.copy(), nms_thd=thd, max_before_nms=mb, max_after_nms=ma)
    assert [e["predictions"] for e in got.to_list()] == want_v
    got_l = postproc.post_processing_vcmr_nms(copy.deepcopy(lists), nms_thd=thd, max_before_nms=mb, max_after_nms=ma)
    assert [e["predictions"] for e in got_l] == want_v
    # SVMR: one video per query, [vid] + NMS of [st, ed, score]
    res.vid[:] = res.vid[:, :1]
    keep = res.count > 0                       # (the reference indexes predictions[0] and fails on an empty list)
    sv = [e for e, k in zip(res.to_list(), keep) if k]
    want_s = [[[e["predictions"][0][0]] + p for p in
               O.temporal_nms([list(p[1:]) for p in e["predictions"][:mb]], thd)[:ma]] for e in sv]
    got_s = postproc.post_processing_svmr_nms(res.copy(), nms_thd=thd, max_before_nms=mb, max_after_nms=ma).to_list()
    assert [e["predictions"] for e, k in zip(got_s, keep) if k] == want_s
    got_sl = postproc.post_processing_svmr_nms(copy.deepcopy(sv), nms_thd=thd, max_before_nms=mb, max_after_nms=ma)
    assert [e["predictions"] for e in got_sl] == want_s


def test_moment_results_roundtrip_and_protocol():
    from tvretrieval_amd.results import MOMENT_DTYPE, MomentResults, to_lists
    rng = np.random.default_rng(5)
    res = _random_result_set(rng, 12, 9)
    lists = res.to_list()
    back = MomentResults.from_list(lists, width=9)
    for a in ("vid", "st", "ed", "score"):
        m = np.arange(9)[None, :] < res.count[:, None]
        np.testing.assert_array_equal(np.where(m, getattr(back, a), 0), np.where(m, getattr(res, a), 0))
    np.testing.assert_array_equal(back.count, res.count)
    assert len(res) == 12 and res[2] == lists[2] and res[-1] == lists[-1] and list(res) == lists and res[1:3] == lists[1:3]
    assert json.loads(json.dumps(to_lists(dict(VCMR=res, video2idx={"a": 1})))) == dict(VCMR=lists, video2idx={"a": 1})
    t = res.copy().truncate(4)
    assert [e["predictions"] for e in t.to_list()] == [e["predictions"][:4] for e in lists]
    # K10's records: f32 seconds widen exactly; SVMR clip units are scaled in float64 like the reference's tail
    rec = np.zeros((2, 3), MOMENT_DTYPE)
    rec["vid"], rec["st"], rec["ed"], rec["score"] = 7, np.float32(3), np.float32(5), np.float32(0.1)
    r = MomentResults.from_records([1, 2], ["", ""], rec, np.array([3, 1]), scale=0.1)
    assert r.to_list()[0]["predictions"][0] == [7, 3 * 0.1, 5 * 0.1, float(np.float32(0.1))]
    assert len(r.to_list()[1]["predictions"]) == 1
    vr = MomentResults([1], [""], [[4, 9]], np.zeros((1, 2)), np.zeros((1, 2)), [[0.5, 0.25]], [2], int_spans=True)
    assert vr.to_list()[0]["predictions"] == [[4, 0, 0, 0.5], [9, 0, 0, 0.25]] == vr[0]["predictions"]
    assert json.dumps(vr.to_list()[0]["predictions"][0]) == "[4, 0, 0, 0.5]"
    assert MomentResults.from_list(vr.to_list()).int_spans and not MomentResults.from_list(lists).int_spans
    cat = MomentResults.concat([res, t])
    assert len(cat) == 24 and cat.width == 9 and cat.to_list() == lists + t.to_list()


@pytest.mark.parametrize("name", ["eval_tvr_style", "eval_didemo_style", "eval_more_tiny", "eval_more_large",
                                  "eval_more_vcmr_only", "eval_more_svmr_vr_didemo"])
def test_evaluator_array_path_matches_reference(name):
    """eval_retrieval on MomentResults (the engine's arrays) gives the reference's metrics, like the list path."""
    from tvretrieval_amd import evaluate
    from tvretrieval_amd.results import MomentResults
    case = json.load(open(os.path.join(GOLDEN, name + ".json")))
    sub = {k: (v if k == "video2idx" else MomentResults.from_list(v)) for k, v in case["submission"].items()}
    got = evaluate.eval_retrieval(sub, case["ground_truth"], iou_thds=(0.5, 0.7), verbose=False, match_number=True,
                                  use_desc_type=case["use_desc_type"])
    assert json.loads(json.dumps(got)) == case["metrics"]


def test_ground_truth_cache_is_keyed_on_every_desc_id():
    """match_number=False with two prediction sets of the same size and the same first / last desc_id but different middle
    ids: the second call must not be served the first call's ground-truth arrays."""
    from tvretrieval_amd import evaluate
    gt = [dict(desc_id=i, vid_name="v%d" % i, ts=[1.0, 5.0], type="v") for i in range(5)]
    v2i = {"v%d" % i: i for i in range(5)}

    def preds(ids):          # every listed query predicts ITS OWN video at the right span: R@1 = 100 for any subset
        return [dict(desc_id=i, desc="", predictions=[[i, 1.0, 5.0, 1.0]]) for i in ids]
    kw = dict(iou_thds=(0.5,), recall_topks=(1,), task_type="VCMR", match_number=False, verbose=False, use_desc_type=False)
    a = evaluate.eval_by_task_type(preds([0, 1, 2, 4]), v2i, gt, **kw)
    b = evaluate.eval_by_task_type(preds([0, 1, 3, 4]), v2i, gt, **kw)
    a, b = (x[0] if isinstance(x, tuple) else x for x in (a, b))
    assert a == b and all(v == 100.0 for v in a.values()), (a, b)


def test_span_predictor_module_keeps_its_gradient():
    """{merged,video,sub}_{st,ed}_predictor are nn.Conv1d parameters in the reference (xml/model_xml.py:97-130): called
    directly in a training graph they must carry gradient to the taps and to the input."""
    import torch
    from tvretrieval_amd.model_xml import _SpanConv
    m = _SpanConv(1, 1, 5, stride=1, padding=2, bias=False)
    ref = torch.nn.Conv1d(1, 1, 5, stride=1, padding=2, bias=False)
    ref.load_state_dict(m.state_dict())
    x = torch.randn(7, 1, 19, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    m(x).square().sum().backward()
    ref(x2).square().sum().backward()
    assert torch.allclose(m.weight.grad, ref.weight.grad) and torch.allclose(x.grad, x2.grad)
