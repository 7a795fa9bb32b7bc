"""Post-processing of retrieval results ("next" row 8f-1): temporal NMS + submission trimming.

Mirrors the reference helpers that xml/inference.py imports
(baselines/clip_alignment_with_language/inference.py:189-265,503-515; utils/temporal_nms.py:25-74) with the same names
and list-of-dict formats; the O(n^2) greedy suppression runs in C++ on the host (libxmlhip.so, xml_nms_*_host)."""
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


def post_processing_vcmr_nms(vcmr_res, nms_thd=0.6, max_before_nms=1000, max_after_nms=100):
    out = []
    for e in vcmr_res:
        e["predictions"] = filter_vcmr_by_nms(e["predictions"], nms_threshold=nms_thd, max_before_nms=max_before_nms,
                                              max_after_nms=max_after_nms)
        out.append(e)
    return out


def post_processing_svmr_nms(svmr_res, nms_thd=0.6, max_before_nms=1000, max_after_nms=100):
    out = []
    for e in svmr_res:
        preds = [d[1:] for d in e["predictions"][:max_before_nms]]
        kept = temporal_non_maximum_suppression(preds, nms_threshold=nms_thd)[:max_after_nms]
        video_id = e["predictions"][0][0]
        e["predictions"] = [[video_id, ] + list(d) for d in kept]
        out.append(e)
    return out


def get_submission_top_n(submission, top_n=100):
    """baselines/clip_alignment_with_language/inference.py:503-515."""
    res = dict(video2idx=submission["video2idx"])
    for k in submission:
        if k != "video2idx":
            for e in submission[k]:
                e["predictions"] = e["predictions"][:top_n]
            res[k] = submission[k]
    return res
