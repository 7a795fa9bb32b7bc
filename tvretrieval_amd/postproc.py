"""Post-processing of retrieval results ("next" row 8f-1): temporal NMS + submission trimming.

Mirrors the reference helpers that xml/inference.py imports
(baselines/clip_alignment_with_language/inference.py:189-265,503-515; utils/temporal_nms.py:25-74) with the same names
and list-of-dict formats; the O(n^2) greedy suppression runs in C++ on the host (libxmlhip.so, xml_nms_*_host).  Whole
result sets -- the engine's array-shaped results.MomentResults or the reference's list of dicts -- go through ONE call of
the *_batched_host entries (queries spread over host threads) instead of one ctypes call per query."""
import ctypes

import numpy as np

from . import _lib


def _dbl(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _run_vcmr(preds, nms_thd, max_before_nms, max_after_nms):
    n = len(preds)
    if n == 0:
        return []
    arr = np.asarray([p[:4] for p in preds], dtype=np.float64)
    vid = np.ascontiguousarray(arr[:, 0].astype(np.int64))
    st, ed, sc = _dbl(arr[:, 1]), _dbl(arr[:, 2]), _dbl(arr[:, 3])
    out = np.empty(n, dtype=np.int32)
    n_out = ctypes.c_int32(0)
    _lib.check(_lib.load().xml_nms_vcmr_host(vid.ctypes.data, st.ctypes.data, ed.ctypes.data, sc.ctypes.data, n,
                                             float(nms_thd), int(max_before_nms), int(max_after_nms), out.ctypes.data,
                                             ctypes.addressof(n_out)), "xml_nms_vcmr_host")
    return [preds[i] for i in out[:n_out.value]]


def temporal_non_maximum_suppression(predictions, nms_threshold, max_after_nms=100):
    """predictions: list of [st, ed, score] -> kept predictions, best first (utils/temporal_nms.py:25-74)."""
    n = len(predictions)
    if n <= 1:
        return predictions
    arr = np.asarray(predictions, dtype=np.float64)
    st, ed, sc = _dbl(arr[:, 0]), _dbl(arr[:, 1]), _dbl(arr[:, 2])
    out = np.empty(n, dtype=np.int32)
    n_out = ctypes.c_int32(0)
    # svmr entry = plain NMS over the first max_before predictions; its internal per-list cap is 100 like the reference
    _lib.check(_lib.load().xml_nms_svmr_host(st.ctypes.data, ed.ctypes.data, sc.ctypes.data, n, float(nms_threshold), n,
                                             int(max_after_nms), out.ctypes.data, ctypes.addressof(n_out)),
               "xml_nms_svmr_host")
    return [list(predictions[i]) for i in out[:n_out.value]]


def filter_vcmr_by_nms(all_video_predictions, nms_threshold=0.6, max_before_nms=1000, max_after_nms=100,
                       score_col_idx=3):
    assert score_col_idx == 3
    return _run_vcmr(all_video_predictions, nms_threshold, max_before_nms, max_after_nms)


def nms_batched(res, task, nms_thd, max_before_nms=1000, max_after_nms=100, n_threads=0):
    """Temporal NMS of a whole result set in one call: res = results.MomentResults -> (index (Nq, max_after) int32 into
    each row, count (Nq,) int32) from xml_nms_{vcmr,svmr}_batched_host (host C++, queries spread over threads).
    task "VCMR" = filter_vcmr_by_nms per query, "SVMR" = post_processing_svmr_nms per query."""
    nq = len(res)
    idx = np.zeros((nq, max(int(max_after_nms), 1)), dtype=np.int32)
    cnt = np.zeros(nq, dtype=np.int32)
    if nq == 0:
        return idx, cnt
    lib = _lib.load()
    if task == "VCMR":
        _lib.check(lib.xml_nms_vcmr_batched_host(res.vid.ctypes.data, res.st.ctypes.data, res.ed.ctypes.data,
                                                 res.score.ctypes.data, res.count.ctypes.data, nq, res.width,
                                                 float(nms_thd), int(max_before_nms), int(max_after_nms), idx.ctypes.data,
                                                 idx.shape[1], cnt.ctypes.data, int(n_threads)),
                   "xml_nms_vcmr_batched_host")
    else:
        _lib.check(lib.xml_nms_svmr_batched_host(res.st.ctypes.data, res.ed.ctypes.data, res.score.ctypes.data,
                                                 res.count.ctypes.data, nq, res.width, float(nms_thd), int(max_before_nms),
                                                 int(max_after_nms), idx.ctypes.data, idx.shape[1], cnt.ctypes.data,
                                                 int(n_threads)), "xml_nms_svmr_batched_host")
    return idx, cnt


def _post_nms(task_res, task, nms_thd, max_before_nms, max_after_nms):
    from .results import MomentResults
    if isinstance(task_res, MomentResults):         # the engine's arrays: stays arrays
        idx, cnt = nms_batched(task_res, task, nms_thd, max_before_nms, max_after_nms)
        return task_res.take(idx, cnt)
    # the reference's list of dicts: one conversion to columns, one batched call, the kept rows picked from the ORIGINAL
    # lists (mutating the entries like the reference does)
    res = MomentResults.from_list(task_res)
    idx, cnt = nms_batched(res, task, nms_thd, max_before_nms, max_after_nms)
    out = []
    for e, row, c in zip(task_res, idx.tolist(), cnt.tolist()):
        preds = e["predictions"]
        if task == "VCMR":
            e["predictions"] = [preds[i] for i in row[:c]]
        else:
            video_id = preds[0][0]
            e["predictions"] = [[video_id, ] + list(preds[i][1:]) for i in row[:c]]
        out.append(e)
    return out


def post_processing_vcmr_nms(vcmr_res, nms_thd=0.6, max_before_nms=1000, max_after_nms=100):
    """clip_alignment_with_language/inference.py:228-244; vcmr_res: the reference's list of dicts or a MomentResults."""
    return _post_nms(vcmr_res, "VCMR", nms_thd, max_before_nms, max_after_nms)


def post_processing_svmr_nms(svmr_res, nms_thd=0.6, max_before_nms=1000, max_after_nms=100):
    """clip_alignment_with_language/inference.py:247-265; svmr_res: the reference's list of dicts or a MomentResults."""
    return _post_nms(svmr_res, "SVMR", nms_thd, max_before_nms, max_after_nms)


def get_submission_top_n(submission, top_n=100):
    """baselines/clip_alignment_with_language/inference.py:503-515 (truncates the entries of `submission` in place)."""
    from .results import MomentResults
    res = dict(video2idx=submission["video2idx"])
    for k in submission:
        if k != "video2idx":
            if isinstance(submission[k], MomentResults):
                submission[k].truncate(top_n)
            else:
                for e in submission[k]:
                    e["predictions"] = e["predictions"][:top_n]
            res[k] = submission[k]
    return res
