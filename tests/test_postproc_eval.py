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
