"""TVR retrieval metrics ("next" row 8f-2): R@{1,5,10,100} at IoU {0.5,0.7} for VCMR / SVMR / VR.

Vectorised numpy restatement of standalone_eval/eval.py:83-276 (same function names, arguments and result keys, same
fp32 IoU arithmetic, hull "union", `>=` thresholds, DiDeMo >= 2-of-4 rule, per-description-type breakdown).  This is a
CPU evaluator in the reference as well; it is not part of the device hot path.  (`np.bool` of the reference is `bool`.)"""
from collections import OrderedDict

import numpy as np

TASK_TYPES = OrderedDict([("VCMR", "Video Corpus Moment Retrieval"), ("SVMR", "Single Video Moment Retrieval"),
                          ("VR", "regular Video Retrieval")])
DESC_TYPE2IDX = {"v": 0, "t": 1, "vt": 2}


def get_rounded_percentage(float_number, n_floats=2):
    return round(float_number * 100, n_floats)


def compute_temporal_iou_batch(preds, gt):
    inter = np.maximum(0, np.minimum(preds[..., 1], gt[..., 1]) - np.maximum(preds[..., 0], gt[..., 0]))
    union = np.maximum(preds[..., 1], gt[..., 1]) - np.minimum(preds[..., 0], gt[..., 0])
    return np.divide(inter, union, out=np.zeros_like(inter), where=union != 0)


_GT_CACHE = {}


def _ground_truth_arrays(ground_truth, keys, gt_by_id, video2idx, use_desc_type):
    """(gt video idx f32, description type, gt spans (n, n_ts, 2) f32, spans per query) for `keys`.  The three tasks of one
    eval_retrieval call -- and the calls before / after NMS -- share one ground truth: built once per (list, key set)."""
    ck = (id(ground_truth), len(ground_truth), id(video2idx), bool(use_desc_type), tuple(keys))
    hit = _GT_CACHE.get(ck)
    if hit is not None and hit[0] is ground_truth:      # (the whole key tuple is in ck: a different middle key misses)
        return hit[1]
    n_desc = len(keys)
    gt_vid = np.zeros(n_desc, dtype=np.float32)
    desc_types = np.zeros(n_desc, dtype=np.int64)
    n_ts = max(len(gt_by_id[k]["ts"]) if len(gt_by_id[k]["ts"]) >= 4 else 1 for k in keys)
    gt_ts = np.zeros((n_desc, n_ts, 2), dtype=np.float32)
    n_gt = np.ones(n_desc, dtype=np.int64)
    for i, k in enumerate(keys):
        g = gt_by_id[k]
        gt_vid[i] = video2idx[g["vid_name"]]
        if use_desc_type:
            desc_types[i] = DESC_TYPE2IDX[g["type"]]
        if len(g["ts"]) >= 4:                                   # didemo: list of [st, ed]
            ts = np.asarray(g["ts"], dtype=np.float32)
            gt_ts[i, :len(ts)] = ts
            n_gt[i] = len(ts)
        else:
            gt_ts[i, 0] = np.asarray(g["ts"], dtype=np.float32)
    out = (gt_vid, desc_types, gt_ts, n_gt)
    _GT_CACHE.clear()                       # one ground truth at a time (an evaluation run)
    _GT_CACHE[ck] = (ground_truth, out)
    return out


def eval_by_task_type(moment_predictions, video2idx, ground_truth, iou_thds=(0.5, 0.7), recall_topks=(1, 5, 10, 100),
                      task_type="SVMR", max_pred_per_query=100, match_number=True, verbose=True, use_desc_type=True):
    assert task_type in TASK_TYPES
    from .results import MomentResults
    arrays = isinstance(moment_predictions, MomentResults)     # the engine's (Nq, n) columns: no per-query list walking
    if arrays:
        pred_by_id = {d: i for i, d in enumerate(moment_predictions.desc_ids)}
    else:
        pred_by_id = {e["desc_id"]: e for e in moment_predictions}
    gt_by_id = {e["desc_id"]: e for e in ground_truth}
    if match_number:
        assert set(gt_by_id.keys()) == set(pred_by_id.keys()), "desc_ids in predictions and ground_truth must match"
    keys = [k for k in gt_by_id if match_number or k in pred_by_id]
    n_desc = len(keys)
    if arrays:
        rows = np.array([pred_by_id[k] for k in keys], dtype=np.int64)
        cnt = np.minimum(moment_predictions.count[rows], max_pred_per_query)
        n_pred = int(cnt.max())
        P = np.zeros((n_desc, n_pred, 3), dtype=np.float32)   # [vid, st, ed], zero padded like pad_sequences_1d_np
        valid = np.arange(n_pred)[None, :] < cnt[:, None]
        for c, col in enumerate((moment_predictions.vid, moment_predictions.st, moment_predictions.ed)):
            P[..., c] = np.where(valid, col[rows, :n_pred], 0)
    else:
        n_pred = max(min(len(pred_by_id[k]["predictions"]), max_pred_per_query) for k in keys)
        P = np.zeros((n_desc, n_pred, 3), dtype=np.float32)       # [vid, st, ed], zero padded like pad_sequences_1d_np
        valid = np.zeros((n_desc, n_pred), dtype=bool)
    gt_vid, desc_types, gt_ts, n_gt = _ground_truth_arrays(ground_truth, keys, gt_by_id, video2idx, use_desc_type)
    if not arrays:
        for i, k in enumerate(keys):
            pr = [e[:3] for e in pred_by_id[k]["predictions"]][:max_pred_per_query]
            if len(pr):
                P[i, :len(pr)] = np.asarray(pr, dtype=np.float32)
                valid[i, :len(pr)] = True
    vid_match = (P[..., 0] == gt_vid[:, None]) & valid                      # (n_desc, n_pred)
    iou = compute_temporal_iou_batch(P[:, :, None, 1:3], gt_ts[:, None, :, :]) * vid_match[..., None]   # (n_desc,n_pred,n_ts)
    ts_valid = np.arange(gt_ts.shape[1])[None, :] < n_gt[:, None]            # (n_desc, n_ts)
    multi = n_gt >= 4
    corrects = []
    for thd in iou_thds:
        hit = (iou >= thd) & ts_valid[:, None, :]
        c = np.where(multi[:, None], hit.sum(-1) >= 2, hit[..., 0]) & valid
        corrects.append(c)

    metrics, metrics_by_type = OrderedDict(), OrderedDict()

    def first_k_hit(c, k):                      # VCMR / VR: any positive among the first k predictions
        return c[:, :k].any(1)

    rank_cache = []

    def first_k_hit_matched(c, k):              # SVMR: among the first k predictions OF THE GT VIDEO
        if not rank_cache:                      # 1-based rank of each matched prediction: one cumsum for all thresholds / ks
            rank_cache.append(np.cumsum(vid_match, axis=1, dtype=np.int32))
        return (c & vid_match & (rank_cache[0] <= k)).any(1)

    if task_type == "VCMR":
        for c, thd in zip(corrects, iou_thds):
            for k in recall_topks:
                metrics["{}-r{}".format(thd, k)] = get_rounded_percentage(np.mean(first_k_hit(c, k)))
        if use_desc_type:
            for dt, di in DESC_TYPE2IDX.items():
                tc = desc_types == di
                for c, thd in zip(corrects, iou_thds):
                    for k in recall_topks:
                        metrics_by_type["{}-{}-r{}".format(dt, thd, k)] = get_rounded_percentage(
                            1.0 * np.sum(first_k_hit(c, k) & tc) / np.sum(tc))
    elif task_type == "SVMR":
        for c, thd in zip(corrects, iou_thds):
            for k in recall_topks:
                metrics["{}-r{}".format(thd, k)] = get_rounded_percentage(np.mean(first_k_hit_matched(c, k)))
        if use_desc_type:
            for dt, di in DESC_TYPE2IDX.items():
                tc = desc_types == di
                for c, thd in zip(corrects, iou_thds):
                    for k in recall_topks:
                        metrics_by_type["{}-{}-r{}".format(dt, thd, k)] = get_rounded_percentage(
                            1.0 * np.sum(first_k_hit_matched(c, k) & tc) / np.sum(tc))
    else:   # VR
        for k in recall_topks:
            metrics["r{}".format(k)] = get_rounded_percentage(np.mean(first_k_hit(vid_match, k)))
        if use_desc_type:
            for dt, di in DESC_TYPE2IDX.items():
                tc = desc_types == di
                for k in recall_topks:
                    metrics_by_type["{}-r{}".format(dt, k)] = get_rounded_percentage(
                        1.0 * np.sum(first_k_hit(vid_match, k) & tc) / np.sum(tc))
    if use_desc_type:
        metrics_by_type["desc_type_ratio"] = "v {} t {} vt {}".format(
            *[get_rounded_percentage(1.0 * np.sum(desc_types == DESC_TYPE2IDX[k]) / len(desc_types)) for k in ["v", "t", "vt"]])
    return metrics, metrics_by_type


def eval_retrieval(submission, ground_truth, iou_thds=(0.5, 0.7), verbose=True, match_number=True, use_desc_type=True):
    """standalone_eval/eval.py:255-276."""
    video2idx = submission["video2idx"]
    tasks = [k for k in TASK_TYPES if k in submission]
    raw = {}
    for t in tasks:
        m, mt = eval_by_task_type(submission[t], video2idx, ground_truth, iou_thds=iou_thds,
                                  recall_topks=(1, 5, 10, 100), task_type=t, max_pred_per_query=100,
                                  match_number=match_number, verbose=verbose, use_desc_type=use_desc_type)
        raw[t], raw[t + "_by_type"] = m, mt
    out = OrderedDict((t, raw[t]) for t in tasks)
    if use_desc_type:
        for t in tasks:
            out[t + "_by_type"] = raw[t + "_by_type"]
    return out
