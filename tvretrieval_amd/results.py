"""Array-shaped retrieval results: what compute_query2ctx_info produces before anything turns it into Python lists.

The reference builds, per query, `predictions = [[video_idx, st, ed, score], ...]` in Python loops
(xml/inference.py:391-445, :229-239) and every later stage -- get_submission_top_n, temporal NMS, the evaluator --
walks those lists again.  Here one task's results are four (Nq, n) columns + a per-row count, filled from the
16-byte records of the device epilogue K10 (xml_moments_decode) in one copy; truncation, NMS (xml_nms_*_batched_host)
and the evaluator work on the columns, and the reference's nested lists are materialised only when a caller asks for
them (`to_list()`, built in C by csrc/pylists.c).

Column types are the Python types the reference's lists hold, widened exactly: vid int64 (Python int), st / ed / score
float64 (Python float).  VCMR seconds were computed in float32 on the device with numpy's arithmetic and are widened;
SVMR seconds are float64 products like the reference's `_sorted_triples[:, :2] * clip_length`.
"""
import ctypes
import gc
import os

import numpy as np

MOMENT_DTYPE = np.dtype([("vid", "<i4"), ("st", "<f4"), ("ed", "<f4"), ("score", "<f4")])   # xml_moment
_HERE = os.path.dirname(os.path.abspath(__file__))
_PYLIB_PATH = os.path.join(_HERE, "csrc", "libxmlpy.so")
_pylib = None


def _load_pylib():
    global _pylib
    if _pylib is None:
        if not os.path.isfile(_PYLIB_PATH):
            raise RuntimeError("libxmlpy.so not found at %s -- run tvretrieval_amd/csrc/build.sh" % _PYLIB_PATH)
        lib = ctypes.PyDLL(_PYLIB_PATH)
        lib.xmlpy_prediction_rows.restype = ctypes.py_object
        lib.xmlpy_prediction_rows.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        lib.xmlpy_rows_to_arrays.restype = ctypes.c_int
        lib.xmlpy_rows_to_arrays.argtypes = [ctypes.py_object] + [ctypes.c_void_p] * 5 + [ctypes.c_int64]
        _pylib = lib
    return _pylib


class MomentResults(object):
    """One task's result set.  Behaves like the reference's list of dicts where that is cheap (len, indexing, iteration:
    one query materialised at a time) and exposes the columns for everything that is not."""

    def __init__(self, desc_ids, descs, vid, st, ed, score, count, int_spans=False):
        nq = len(desc_ids)
        # video-retrieval lists: the reference's entries are [video_idx, 0, 0, score] with INTEGER zeros (xml/inference.py:409)
        self.int_spans = bool(int_spans)
        self.desc_ids = list(desc_ids)
        self.descs = list(descs)
        self.vid = np.ascontiguousarray(vid, dtype=np.int64).reshape(nq, -1)
        self.st = np.ascontiguousarray(st, dtype=np.float64).reshape(nq, -1)
        self.ed = np.ascontiguousarray(ed, dtype=np.float64).reshape(nq, -1)
        self.score = np.ascontiguousarray(score, dtype=np.float64).reshape(nq, -1)
        self.count = np.ascontiguousarray(count, dtype=np.int32).reshape(nq)
        assert self.vid.shape == self.st.shape == self.ed.shape == self.score.shape
        assert len(self.descs) == nq

    # ---- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def from_records(cls, desc_ids, descs, rec, count, scale=None, int_spans=False):
        """rec: (Nq, n) array of MOMENT_DTYPE (K10's output), count (Nq,).  scale: float64 factor applied to st / ed after
        widening (SVMR: the records hold clip units, the reference multiplies by clip_length in float64)."""
        rec = np.asarray(rec)
        st = rec["st"].astype(np.float64)
        ed = rec["ed"].astype(np.float64)
        if scale is not None:
            st *= float(scale)
            ed *= float(scale)
        return cls(desc_ids, descs, rec["vid"], st, ed, rec["score"], count, int_spans=int_spans)

    @classmethod
    def from_list(cls, res, width=None):
        """The reference's format -> columns.  res: list of dict(desc_id, desc, predictions=[[vid, st, ed, score], ...])."""
        nq = len(res)
        n = max([len(e["predictions"]) for e in res] + [1]) if width is None else int(width)
        vid = np.zeros((nq, n), np.int64)
        st, ed, sc = (np.zeros((nq, n), np.float64) for _ in range(3))
        cnt = np.zeros(nq, np.int32)
        rows = [e["predictions"] if isinstance(e["predictions"], list) else list(e["predictions"]) for e in res]
        if _load_pylib().xmlpy_rows_to_arrays(rows, vid.ctypes.data, st.ctypes.data, ed.ctypes.data, sc.ctypes.data,
                                              cnt.ctypes.data, n) != 0:
            raise ValueError("malformed predictions")      # (ctypes.PyDLL re-raises the Python exception set in C first)
        first = next((r[0] for r in rows if len(r)), None)
        return cls([e["desc_id"] for e in res], [e.get("desc", "") for e in res], vid, st, ed, sc, cnt,
                   int_spans=first is not None and type(first[1]) is int and type(first[2]) is int)

    @classmethod
    def concat(cls, parts):
        """Row-wise concatenation of result sets of possibly different widths (zero-padded to the widest)."""
        n = max(p.width for p in parts)

        def col(name, dtype):
            out = np.zeros((sum(len(p) for p in parts), n), dtype)
            r = 0
            for p in parts:
                out[r:r + len(p), :p.width] = getattr(p, name)
                r += len(p)
            return out
        return cls(sum((p.desc_ids for p in parts), []), sum((p.descs for p in parts), []), col("vid", np.int64),
                   col("st", np.float64), col("ed", np.float64), col("score", np.float64),
                   np.concatenate([p.count for p in parts]), int_spans=all(p.int_spans for p in parts))

    # ---- the cheap part of the list protocol ------------------------------------------------------------------------
    @property
    def width(self):
        return self.vid.shape[1]

    def __len__(self):
        return len(self.desc_ids)

    def predictions(self, i):
        n = int(self.count[i])
        st, ed = self.st[i, :n], self.ed[i, :n]
        if self.int_spans:
            st, ed = st.astype(np.int64), ed.astype(np.int64)
        return [list(t) for t in zip(self.vid[i, :n].tolist(), st.tolist(), ed.tolist(), self.score[i, :n].tolist())]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return dict(desc_id=self.desc_ids[i], desc=self.descs[i], predictions=self.predictions(i))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # ---- array-side operations ----------------------------------------------------------------------------------------
    def truncate(self, top_n):
        """get_submission_top_n on one task, in place like the reference's (clip_alignment_with_language/inference.py:503-515)."""
        top_n = int(top_n)
        if top_n < self.width:
            self.vid = np.ascontiguousarray(self.vid[:, :top_n])
            self.st = np.ascontiguousarray(self.st[:, :top_n])
            self.ed = np.ascontiguousarray(self.ed[:, :top_n])
            self.score = np.ascontiguousarray(self.score[:, :top_n])
        np.minimum(self.count, top_n, out=self.count)
        return self

    def copy(self):
        return MomentResults(self.desc_ids, self.descs, self.vid.copy(), self.st.copy(), self.ed.copy(), self.score.copy(),
                             self.count.copy(), int_spans=self.int_spans)

    def take(self, index, count):
        """Rows re-ordered / filtered by per-row index lists: index (Nq, m) int32 into each row, count (Nq,) valid entries."""
        index = np.asarray(index, dtype=np.int64)
        keep = np.arange(index.shape[1])[None, :] < np.asarray(count)[:, None]
        idx = np.where(keep, index, 0)

        def g(a):
            return np.where(keep, np.take_along_axis(a, idx, axis=1), 0)
        return MomentResults(self.desc_ids, self.descs, g(self.vid), g(self.st), g(self.ed), g(self.score), count,
                             int_spans=self.int_spans)

    def to_list(self):
        """The reference's format: [dict(desc_id, desc, predictions=[[video_idx, st, ed, score], ...]), ...]."""
        was = gc.isenabled()
        gc.disable()        # millions of small lists that all survive: the generational collector would only re-walk them
        try:
            rows = _load_pylib().xmlpy_prediction_rows(self.vid.ctypes.data, self.st.ctypes.data, self.ed.ctypes.data,
                                                       self.score.ctypes.data, self.count.ctypes.data, len(self), self.width,
                                                       int(self.int_spans))
            return [dict(desc_id=d, desc=t, predictions=r) for d, t, r in zip(self.desc_ids, self.descs, rows)]
        finally:
            if was:
                gc.enable()


def as_results(task_res, width=None):
    return task_res if isinstance(task_res, MomentResults) else MomentResults.from_list(task_res, width)


def to_lists(res):
    """A result dict whose tasks may be MomentResults -> the reference's plain dict of lists."""
    return {k: (v.to_list() if isinstance(v, MomentResults) else v) for k, v in res.items()}
