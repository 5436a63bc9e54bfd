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
 *   shape_fused(...) -> the result dicts of HybridSearcher.fuse for a batch (see below)
 *
 * No arithmetic of the hot path lives here; this is the host-side boundary of the Python class surface.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *k_index, *k_score, *k_rank, *k_similarity;
static PyObject *k_note_id, *k_scores, *k_final, *k_tags, *k_source, *k_is_bridge, *k_src[4], *v_graph, *v_semantic;

static PyObject *shape_hits(PyObject *self, PyObject *args) {
  Py_buffer bi, bs;
  Py_ssize_t nq, k;
  int cosine;
  /* the cyclic collector is paused while the (acyclic) result objects are built: tens of thousands of new dicts would
   * trigger generation after generation of traversals — half the shaping time */
  int gc_was_on = 0;
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
    gc_was_on = PyGC_Disable();
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
          /* int / float values only: the collector would untrack this dict at its first pass (dicts of atomic values
           * are never part of a cycle) — after traversing it.  6 400 hits per batch are 6 400 fewer objects for the
           * generation-0 pass that the very next allocation triggers (an insertion of a container re-tracks a dict). */
          if (!bad) PyObject_GC_UnTrack(d);
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
  if (gc_was_on) PyGC_Enable();
  PyBuffer_Release(&bi);
  PyBuffer_Release(&bs);
  return out;
}

/* shape_fused(ids int64 [nq*pool], finals float64 [nq*pool], src float64 [nq*pool*4] (NaN = absent), counts int32 [nq],
 *             nq, pool, note_ids: sequence or None) -> list[list[dict]]
 * The result dicts of HybridSearcher.fuse (retrieval/hybrid_search.py:85-103): {"note_id", "scores": {dense, bm25,
 * graph, path: float | None}, "final_similarity", "tags": {"source": "graph" | "semantic", "is_bridge": bool}}. */
static PyObject *shape_fused(PyObject *self, PyObject *args) {
  Py_buffer bi, bf, bs, bc;
  Py_ssize_t nq, pool;
  PyObject *names;
  int gc_was_on = 0;
  if (!PyArg_ParseTuple(args, "y*y*y*y*nnO", &bi, &bf, &bs, &bc, &nq, &pool, &names)) return NULL;
  PyObject *out = NULL;
  if (nq < 0 || pool < 0 || bi.len < (Py_ssize_t)(nq * pool * 8) || bf.len < (Py_ssize_t)(nq * pool * 8) ||
      bs.len < (Py_ssize_t)(nq * pool * 32) || bc.len < (Py_ssize_t)(nq * 4)) {
    PyErr_SetString(PyExc_ValueError, "shape_fused: buffers shorter than nq * pool entries");
    goto done;
  }
  {
    const int64_t *I = (const int64_t *)bi.buf;
    const double *F = (const double *)bf.buf, *S = (const double *)bs.buf;
    const int32_t *N = (const int32_t *)bc.buf;
    out = PyList_New(nq);
    if (!out) goto done;
    gc_was_on = PyGC_Disable();
    for (Py_ssize_t q = 0; q < nq; ++q) {
      Py_ssize_t cnt = N[q] < 0 ? 0 : (N[q] > pool ? pool : N[q]);
      PyObject *res = PyList_New(cnt);
      if (!res) goto fail;
      PyList_SET_ITEM(out, q, res);
      for (Py_ssize_t j = 0; j < cnt; ++j) {
        const double *sc = S + (q * pool + j) * 4;
        PyObject *d = _PyDict_NewPresized(4), *scores = _PyDict_NewPresized(4), *tags = _PyDict_NewPresized(2);
        PyObject *id = NULL, *fin = PyFloat_FromDouble(F[q * pool + j]);
        if (names == Py_None) id = PyLong_FromLongLong((long long)I[q * pool + j]);
        else id = PySequence_GetItem(names, (Py_ssize_t)I[q * pool + j]);
        int bad = !d || !scores || !tags || !id || !fin;
        for (int s = 0; s < 4 && !bad; ++s) {
          if (sc[s] != sc[s]) bad = PyDict_SetItem(scores, k_src[s], Py_None);
          else {
            PyObject *v = PyFloat_FromDouble(sc[s]);
            bad = !v || PyDict_SetItem(scores, k_src[s], v);
            Py_XDECREF(v);
          }
        }
        if (!bad)
          bad = PyDict_SetItem(tags, k_source, sc[2] == sc[2] ? v_graph : v_semantic) ||
                PyDict_SetItem(tags, k_is_bridge, sc[3] == sc[3] ? Py_True : Py_False) ||
                PyDict_SetItem(d, k_note_id, id) || PyDict_SetItem(d, k_scores, scores) ||
                PyDict_SetItem(d, k_final, fin) || PyDict_SetItem(d, k_tags, tags);
        /* `scores` and `tags` hold floats / None / str / bool only: untracked here, as the collector itself would after
         * its first traversal (two thirds of the 48 000 containers of a 200-query batch; the collections that the next
         * allocations trigger were 4 ms of a 9-ms fuse_bm25 call — cProfile booked them on whoever allocated next) */
        if (!bad) {
          PyObject_GC_UnTrack(scores);
          PyObject_GC_UnTrack(tags);
        }
        Py_XDECREF(id);
        Py_XDECREF(fin);
        Py_XDECREF(scores);
        Py_XDECREF(tags);
        if (bad) {
          Py_XDECREF(d);
          goto fail;
        }
        PyList_SET_ITEM(res, j, d);
      }
    }
  }
  goto done;
fail:
  Py_CLEAR(out);
done:
  if (gc_was_on) PyGC_Enable();
  PyBuffer_Release(&bi);
  PyBuffer_Release(&bf);
  PyBuffer_Release(&bs);
  PyBuffer_Release(&bc);
  return out;
}

static PyMethodDef methods[] = {
    {"shape_fused", shape_fused, METH_VARARGS, "list of per-query fused result dicts from the anr_fuse_dense outputs"},
    {"shape_hits", shape_hits, METH_VARARGS, "list of per-query hit dicts from (indices int64, scores float32)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyshape", "VectorIndex.search result shaping", -1, methods};

PyMODINIT_FUNC PyInit__pyshape(void) {
  k_index = PyUnicode_InternFromString("index");
  k_score = PyUnicode_InternFromString("score");
  k_rank = PyUnicode_InternFromString("rank");
  k_similarity = PyUnicode_InternFromString("similarity");
  if (!k_index || !k_score || !k_rank || !k_similarity) return NULL;
  k_note_id = PyUnicode_InternFromString("note_id");
  k_scores = PyUnicode_InternFromString("scores");
  k_final = PyUnicode_InternFromString("final_similarity");
  k_tags = PyUnicode_InternFromString("tags");
  k_source = PyUnicode_InternFromString("source");
  k_is_bridge = PyUnicode_InternFromString("is_bridge");
  k_src[0] = PyUnicode_InternFromString("dense");
  k_src[1] = PyUnicode_InternFromString("bm25");
  k_src[2] = PyUnicode_InternFromString("graph");
  k_src[3] = PyUnicode_InternFromString("path");
  v_graph = PyUnicode_InternFromString("graph");
  v_semantic = PyUnicode_InternFromString("semantic");
  if (!k_note_id || !k_scores || !k_final || !k_tags || !k_source || !k_is_bridge || !k_src[0] || !k_src[1] || !k_src[2] ||
      !k_src[3] || !v_graph || !v_semantic)
    return NULL;
  return PyModule_Create(&moddef);
}
