/* libxmlpy.so -- the reference's result format built in C against the CPython API.
 *
 * compute_query2ctx_info returns, per query, predictions = [[video_idx (int), st (float), ed (float), score (float)], ...]
 * (xml/inference.py:402-439, :229-239): 200-1000 four-element lists per query, 2-11 M Python objects for TVR val.  The
 * engine keeps results as (Nq, n) arrays (tvretrieval_amd/results.py); this file turns them into that nested list in one
 * call when a caller asks for the reference's format (JSON submission files).  Loaded with ctypes.PyDLL: every entry runs
 * with the GIL held and returns a new reference.  Host-only, no device code; not part of libxmlhip.so's C ABI because it
 * speaks PyObject*.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* rows[q] = [[vid, st, ed, score] for the first count[q] entries of row q]; arrays are (nq, ld) C-contiguous.
 * int_spans != 0: st / ed as Python ints -- the reference's VR entries are [video_idx, 0, 0, score] with integer zeros
 * (xml/inference.py:409). */
PyObject* xmlpy_prediction_rows(const int64_t* vid, const double* st, const double* ed, const double* score,
                                const int32_t* count, int64_t nq, int64_t ld, int int_spans) {
  PyObject* rows = PyList_New((Py_ssize_t)nq);
  if (!rows) return NULL;
  for (int64_t q = 0; q < nq; ++q) {
    int64_t n = count ? count[q] : ld;
    if (n < 0) n = 0;
    if (n > ld) n = ld;
    PyObject* row = PyList_New((Py_ssize_t)n);
    if (!row) goto fail;
    PyList_SET_ITEM(rows, (Py_ssize_t)q, row);
    const int64_t o = q * ld;
    for (int64_t i = 0; i < n; ++i) {
      PyObject* e = PyList_New(4);
      if (!e) goto fail;
      PyList_SET_ITEM(row, (Py_ssize_t)i, e);
      PyObject* a = PyLong_FromLongLong((long long)vid[o + i]);
      PyObject* b = int_spans ? PyLong_FromLongLong((long long)st[o + i]) : PyFloat_FromDouble(st[o + i]);
      PyObject* c = int_spans ? PyLong_FromLongLong((long long)ed[o + i]) : PyFloat_FromDouble(ed[o + i]);
      PyObject* d = PyFloat_FromDouble(score[o + i]);
      if (!a || !b || !c || !d) {
        Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(d);
        goto fail;
      }
      PyList_SET_ITEM(e, 0, a);
      PyList_SET_ITEM(e, 1, b);
      PyList_SET_ITEM(e, 2, c);
      PyList_SET_ITEM(e, 3, d);
    }
  }
  return rows;
fail:
  /* unfilled slots of `rows` / `row` / `e` are NULL, which list_dealloc accepts */
  Py_DECREF(rows);
  return NULL;
}

/* the reverse: predictions of every query -> (nq, ld) arrays + count; entries beyond ld are dropped.
 * returns 0, or -1 with a Python exception set (malformed rows). */
int xmlpy_rows_to_arrays(PyObject* rows, int64_t* vid, double* st, double* ed, double* score, int32_t* count, int64_t ld) {
  if (!PyList_Check(rows)) { PyErr_SetString(PyExc_TypeError, "rows must be a list"); return -1; }
  const Py_ssize_t nq = PyList_GET_SIZE(rows);
  for (Py_ssize_t q = 0; q < nq; ++q) {
    PyObject* row = PyList_GET_ITEM(rows, q);
    if (!PyList_Check(row)) { PyErr_SetString(PyExc_TypeError, "predictions must be a list"); return -1; }
    Py_ssize_t n = PyList_GET_SIZE(row);
    if (n > ld) n = (Py_ssize_t)ld;
    count[q] = (int32_t)n;
    const int64_t o = (int64_t)q * ld;
    for (Py_ssize_t i = 0; i < n; ++i) {
      PyObject* e = PySequence_Fast(PyList_GET_ITEM(row, i), "a prediction must be a sequence");
      if (!e) return -1;
      if (PySequence_Fast_GET_SIZE(e) < 4) {
        Py_DECREF(e);
        PyErr_SetString(PyExc_ValueError, "a prediction needs [video_idx, st, ed, score]");
        return -1;
      }
      PyObject** it = PySequence_Fast_ITEMS(e);
      vid[o + i] = PyFloat_Check(it[0]) ? (int64_t)PyFloat_AS_DOUBLE(it[0]) : (int64_t)PyLong_AsLongLong(it[0]);
      st[o + i] = PyFloat_AsDouble(it[1]);
      ed[o + i] = PyFloat_AsDouble(it[2]);
      score[o + i] = PyFloat_AsDouble(it[3]);
      Py_DECREF(e);
      if (PyErr_Occurred()) return -1;
    }
  }
  return 0;
}
