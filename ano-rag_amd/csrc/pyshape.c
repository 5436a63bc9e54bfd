/* _pyshape — result shaping of VectorIndex.search in C (CPython API).
 *
 * The reference shapes faiss's (scores, indices) arrays into a list of dicts per query with a Python loop
 * (vector_store/vector_index.py:226-259): {"index", "score", "rank", "similarity"} per hit, ids of -1
 * skipped, rank = position in the row, similarity = score (cosine) or 1 / (1 + score) (L2).  At batch 64 x
 * top-100 that loop costs about as long as the whole 10 M-row scan on the device, so the drop-in builds the
 * same objects here: same keys, same Python types (int / float), same values bit for bit.
 *
 *   shape_hits(indices: buffer int64 [nq*k], scores: buffer float32 [nq*k], nq, k, cosine) -> list[list[dict]]
 *
 * No arithmetic of the hot path lives here; this is the host-side boundary of the Python class surface.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *k_index, *k_score, *k_rank, *k_similarity;

static PyObject *shape_hits(PyObject *self, PyObject *args) {
  Py_buffer bi, bs;
  Py_ssize_t nq, k;
  int cosine;
  if (!PyArg_ParseTuple(args, "y*y*nnp", &bi, &bs, &nq, &k, &cosine)) return NULL;
  PyObject *out = NULL;
  if (nq < 0 || k < 0 || bi.len < (Py_ssize_t)(nq * k * 8) || bs.len < (Py_ssize_t)(nq * k * 4)) {
    PyErr_SetString(PyExc_ValueError, "shape_hits: buffers shorter than nq * k entries");
    goto done;
  }
  {
    const int64_t *I = (const int64_t *)bi.buf;
    const float *S = (const float *)bs.buf;
    out = PyList_New(nq);
    if (!out) goto done;
    for (Py_ssize_t q = 0; q < nq; ++q) {
      PyObject *hits = PyList_New(0);
      if (!hits) goto fail;
      PyList_SET_ITEM(out, q, hits);
      for (Py_ssize_t r = 0; r < k; ++r) {
        const int64_t id = I[q * k + r];
        if (id == -1) continue; /* faiss padding (vector_index.py:234) */
        const double sc = (double)S[q * k + r];
        PyObject *d = _PyDict_NewPresized(4);
        PyObject *vi = PyLong_FromLongLong((long long)id);
        PyObject *vs = PyFloat_FromDouble(sc);
        PyObject *vr = PyLong_FromSsize_t(r);
        PyObject *vm = cosine ? vs : PyFloat_FromDouble(1.0 / (1.0 + sc));
        int bad = !d || !vi || !vs || !vr || !vm;
        if (!bad) {
          bad = PyDict_SetItem(d, k_index, vi) || PyDict_SetItem(d, k_score, vs) || PyDict_SetItem(d, k_rank, vr) ||
                PyDict_SetItem(d, k_similarity, vm);
          if (!bad) bad = PyList_Append(hits, d);
        }
        Py_XDECREF(vi);
        Py_XDECREF(vs);
        Py_XDECREF(vr);
        if (!cosine) Py_XDECREF(vm);
        Py_XDECREF(d);
        if (bad) goto fail;
      }
    }
  }
  goto done;
fail:
  Py_CLEAR(out);
done:
  PyBuffer_Release(&bi);
  PyBuffer_Release(&bs);
  return out;
}

static PyMethodDef methods[] = {
    {"shape_hits", shape_hits, METH_VARARGS, "list of per-query hit dicts from (indices int64, scores float32)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyshape", "VectorIndex.search result shaping", -1, methods};

PyMODINIT_FUNC PyInit__pyshape(void) {
  k_index = PyUnicode_InternFromString("index");
  k_score = PyUnicode_InternFromString("score");
  k_rank = PyUnicode_InternFromString("rank");
  k_similarity = PyUnicode_InternFromString("similarity");
  if (!k_index || !k_score || !k_rank || !k_similarity) return NULL;
  return PyModule_Create(&moddef);
}
