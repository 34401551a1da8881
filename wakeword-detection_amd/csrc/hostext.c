/* _wwhostext - the per-clip part of staging a test split, outside the interpreter's loop.
 *
 * wwhip/evaluate.py stages a rank's share of a test split by handing (address, count) runs to the library's uploader
 * (ww_uploader_submit).  The clips are separate NumPy arrays in a Python list; looking at each one from Python - is it
 * contiguous int16, where does it start, how long is it - costs ~1.6 us per clip, i.e. most of the "slicing" phase of a rank
 * of eight (2 x 316 clips per rank at hey-snips size, round 5: 1.30 ms of a 3.99 ms job).  Here it is one call per chunk:
 * the buffer protocol per clip (~50 ns), nothing NumPy-specific, no GPU, no copy of any sample.
 *
 *   scan_pcm16(seq, idx, addr, nsamp, seen) -> number of clips taken
 *     seq   a list / tuple of objects (the clips)
 *     idx   int64 buffer: the positions in seq to look at
 *     addr, nsamp  writable int64 buffers, one entry per element of seq: for every idx[i] whose object exports a
 *                  C-contiguous one-dimensional buffer of 2-byte signed integers ("h") they receive the address of its first
 *                  sample and its length in samples
 *     seen  writable uint8 buffer, one entry per element of seq: set to 1 for those clips; the others (another dtype, a
 *           strided view, not a buffer at all) are left untouched for the caller's per-clip path
 *   The objects stay referenced by seq; the addresses are valid as long as the caller keeps seq (and does not resize the
 *   arrays), exactly like the addresses evaluate.py took one by one before.
 *
 * Reference: utils/evaluate_models.py:45-61 (every file is read and fed sample by sample through the Filter's ring) - this is
 * the bookkeeping of the replacement's staging, not arithmetic on samples. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static int get_i64(PyObject *o, Py_buffer *v, int writable) {
  if (PyObject_GetBuffer(o, v, (writable ? PyBUF_WRITABLE : 0) | PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) != 0) return -1;
  if (v->itemsize != 8 || !v->format || (v->format[0] != 'l' && v->format[0] != 'q')) {
    PyBuffer_Release(v);
    PyErr_SetString(PyExc_TypeError, "expected a contiguous int64 buffer");
    return -1;
  }
  return 0;
}

static PyObject *scan_pcm16(PyObject *self, PyObject *args) {
  PyObject *seq, *o_idx, *o_addr, *o_n, *o_seen;
  if (!PyArg_ParseTuple(args, "OOOOO", &seq, &o_idx, &o_addr, &o_n, &o_seen)) return NULL;
  PyObject *fast = PySequence_Fast(seq, "scan_pcm16: a list or tuple of clips is needed");
  if (!fast) return NULL;
  Py_buffer idx, addr, ns, seen;
  if (get_i64(o_idx, &idx, 0) != 0) { Py_DECREF(fast); return NULL; }
  if (get_i64(o_addr, &addr, 1) != 0) { PyBuffer_Release(&idx); Py_DECREF(fast); return NULL; }
  if (get_i64(o_n, &ns, 1) != 0) { PyBuffer_Release(&idx); PyBuffer_Release(&addr); Py_DECREF(fast); return NULL; }
  if (PyObject_GetBuffer(o_seen, &seen, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) {
    PyBuffer_Release(&idx); PyBuffer_Release(&addr); PyBuffer_Release(&ns); Py_DECREF(fast);
    return NULL;
  }
  const Py_ssize_t n_seq = PySequence_Fast_GET_SIZE(fast), n_idx = idx.len / 8;
  PyObject *ret = NULL;
  if (addr.len / 8 != n_seq || ns.len / 8 != n_seq || seen.len != n_seq || seen.itemsize != 1) {
    PyErr_SetString(PyExc_ValueError, "scan_pcm16: addr / nsamp / seen need one entry per clip");
    goto done;
  }
  {
    const int64_t *ix = (const int64_t *)idx.buf;
    int64_t *pa = (int64_t *)addr.buf, *pn = (int64_t *)ns.buf;
    unsigned char *ps = (unsigned char *)seen.buf;
    Py_ssize_t taken = 0;
    for (Py_ssize_t i = 0; i < n_idx; ++i) {
      const int64_t k = ix[i];
      if (k < 0 || k >= n_seq) {
        PyErr_SetString(PyExc_IndexError, "scan_pcm16: clip index out of range");
        goto done;
      }
      PyObject *clip = PySequence_Fast_GET_ITEM(fast, k);
      Py_buffer v;
      if (!PyObject_CheckBuffer(clip) || PyObject_GetBuffer(clip, &v, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) != 0) {
        PyErr_Clear();  /* not a (contiguous) buffer: the caller's per-clip path converts or refuses it */
        continue;
      }
      if (v.ndim == 1 && v.itemsize == 2 && v.format && v.format[0] == 'h' && v.format[1] == 0) {
        pa[k] = (int64_t)(intptr_t)v.buf;
        pn[k] = (int64_t)(v.len / 2);
        ps[k] = 1;
        ++taken;
      }
      PyBuffer_Release(&v);
    }
    ret = PyLong_FromSsize_t(taken);
  }
done:
  PyBuffer_Release(&idx); PyBuffer_Release(&addr); PyBuffer_Release(&ns); PyBuffer_Release(&seen);
  Py_DECREF(fast);
  return ret;
}

/* take(seq, idx) -> (list, lengths): the items seq[idx[i]] as a new list and len() of each as an int64 bytes-like (a
 * bytearray of 8 n bytes; np.frombuffer(..., np.int64) views it) - what the evaluators' "prepare" phase needs of EVERY clip of a test
 * split on EVERY rank (the plan is a function of all lengths): two list comprehensions and two passes of len() in the interpreter
 * before (0.22 ms per 2 x 2,529 clips), one call here. */
static PyObject *take(PyObject *self, PyObject *args) {
  PyObject *seq, *o_idx;
  if (!PyArg_ParseTuple(args, "OO", &seq, &o_idx)) return NULL;
  PyObject *fast = PySequence_Fast(seq, "take: a list or tuple is needed");
  if (!fast) return NULL;
  Py_buffer idx;
  if (get_i64(o_idx, &idx, 0) != 0) { Py_DECREF(fast); return NULL; }
  const Py_ssize_t n_seq = PySequence_Fast_GET_SIZE(fast), n = idx.len / 8;
  const int64_t *ix = (const int64_t *)idx.buf;
  PyObject *out = PyList_New(n), *lens = PyByteArray_FromStringAndSize(NULL, n * 8), *ret = NULL;
  if (!out || !lens) goto done;
  {
    int64_t *pl = (int64_t *)PyByteArray_AS_STRING(lens);
    for (Py_ssize_t i = 0; i < n; ++i) {
      const int64_t k = ix[i];
      if (k < 0 || k >= n_seq) {
        PyErr_SetString(PyExc_IndexError, "take: index out of range");
        goto done;
      }
      PyObject *item = PySequence_Fast_GET_ITEM(fast, k);
      const Py_ssize_t len = PyObject_Length(item);
      if (len < 0) goto done;  /* (the item's own error: an object without len()) */
      Py_INCREF(item);
      PyList_SET_ITEM(out, i, item);
      pl[i] = (int64_t)len;
    }
    ret = PyTuple_Pack(2, out, lens);
  }
done:
  PyBuffer_Release(&idx);
  Py_DECREF(fast);
  if (!ret && out) {  /* slots not filled yet are NULL: fill them so that the list can be released */
    for (Py_ssize_t i = 0; i < n; ++i)
      if (!PyList_GET_ITEM(out, i)) { Py_INCREF(Py_None); PyList_SET_ITEM(out, i, Py_None); }
  }
  Py_XDECREF(out);
  Py_XDECREF(lens);
  return ret;
}

static PyMethodDef methods[] = {
    {"take", take, METH_VARARGS, "(seq[idx[i]] as a list, their len() as int64 bytes) - see csrc/hostext.c"},
    {"scan_pcm16", scan_pcm16, METH_VARARGS, "addresses and lengths of the contiguous int16 clips seq[idx[i]] (see csrc/hostext.c)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_wwhostext", "host-side staging helpers of wwhip.evaluate", -1, methods};
PyMODINIT_FUNC PyInit__wwhostext(void) { return PyModule_Create(&moddef); }
