/*
 * _c_utils.c -- native half of the `c_utils` drop-in (CPython C API + numpy C API, built in-tree by
 * hgs_runtime.build(): gcc -shared, no Cython at run time).
 *
 * filter_strand_list_segments(strands_list) -> int64 [pairs, 2, 2]
 *   replaces c_utils.filter_strand_list_segments of the reference (c_utils/c_utils.pyx:83-127): `strands_list` is a 1-D
 *   numpy OBJECT array (or any sequence) of [n_j, 2] int64 arrays; the result holds every pair of consecutive rows of
 *   every strand, strands in order (pairs = sum over strands of max(n_j - 1, 0)).  Same two passes as the reference
 *   (count, then fill); rows are read through their strides, so views and Fortran-ordered strands work too.
 *   Errors as the reference raises them: None -> TypeError; a strand of >= 2 rows that is not a 2-D int64 array ->
 *   ValueError (the reference's typed memoryview refuses it with "Buffer dtype mismatch" / "Buffer has wrong number of
 *   dimensions").
 */
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <numpy/arrayobject.h>

static Py_ssize_t strand_rows(PyObject* item) {
  if (PyArray_Check(item)) {
    PyArrayObject* a = (PyArrayObject*)item;
    if (PyArray_NDIM(a) < 1) {
      PyErr_SetString(PyExc_IndexError, "tuple index out of range");   /* what `.shape[0]` of a 0-d array raises */
      return -1;
    }
    return (Py_ssize_t)PyArray_DIM(a, 0);
  }
  /* anything else with a .shape (the reference only asks for shape[0] in its first pass) */
  PyObject* shape = PyObject_GetAttrString(item, "shape");
  if (!shape) return -1;
  PyObject* first = PySequence_GetItem(shape, 0);
  Py_DECREF(shape);
  if (!first) return -1;
  const Py_ssize_t n = PyLong_AsSsize_t(first);
  Py_DECREF(first);
  return (n == -1 && PyErr_Occurred()) ? -1 : n;
}

static PyObject* filter_strand_list_segments(PyObject* self, PyObject* strands_list) {
  (void)self;
  if (strands_list == Py_None) {
    PyErr_SetString(PyExc_TypeError, "Argument 'strands_list' must not be None");
    return NULL;
  }
  PyObject* seq = PySequence_Fast(strands_list, "strands_list must be a sequence of [n, 2] int64 arrays");
  if (!seq) return NULL;
  const Py_ssize_t n_strands = PySequence_Fast_GET_SIZE(seq);
  PyObject** items = PySequence_Fast_ITEMS(seq);
  npy_intp total = 0;
  for (Py_ssize_t j = 0; j < n_strands; j++) { /* first pass, c_utils.pyx:105-110 */
    const Py_ssize_t n = strand_rows(items[j]);
    if (n < 0 && PyErr_Occurred()) { Py_DECREF(seq); return NULL; }
    if (n >= 2) total += n - 1;
  }
  npy_intp dims[3] = {total, 2, 2};
  PyArrayObject* out = (PyArrayObject*)PyArray_EMPTY(3, dims, NPY_INT64, 0);
  if (!out) { Py_DECREF(seq); return NULL; }
  npy_int64* o = (npy_int64*)PyArray_DATA(out);
  for (Py_ssize_t j = 0; j < n_strands; j++) { /* second pass, :117-127 */
    PyObject* item = items[j];
    const Py_ssize_t n = strand_rows(item);
    if (n < 2) continue;
    if (!PyArray_Check(item) || PyArray_NDIM((PyArrayObject*)item) != 2 || PyArray_TYPE((PyArrayObject*)item) != NPY_INT64 ||
        PyArray_DIM((PyArrayObject*)item, 1) < 2) {
      PyErr_Format(PyExc_ValueError, "strand %zd: expected a 2-D int64 array with 2 columns", j);
      Py_DECREF(out);
      Py_DECREF(seq);
      return NULL;
    }
    PyArrayObject* a = (PyArrayObject*)item;
    const char* base = (const char*)PyArray_DATA(a);
    const npy_intp s0 = PyArray_STRIDE(a, 0), s1 = PyArray_STRIDE(a, 1);
    npy_int64 p0 = *(const npy_int64*)base, p1 = *(const npy_int64*)(base + s1);
    for (Py_ssize_t i = 1; i < n; i++) {
      const npy_int64 q0 = *(const npy_int64*)(base + i * s0), q1 = *(const npy_int64*)(base + i * s0 + s1);
      o[0] = p0; o[1] = p1; o[2] = q0; o[3] = q1;
      o += 4;
      p0 = q0; p1 = q1;
    }
  }
  Py_DECREF(seq);
  return (PyObject*)out;
}

/* filter_strand_segments_flat(offsets[S+1], rows[total, 2]) -> int64 [pairs, 2, 2]: the same pairs from the flat form
 * this package keeps its strands in (strand j = rows[offsets[j] : offsets[j+1]]). */
static PyObject* filter_strand_segments_flat(PyObject* self, PyObject* args) {
  (void)self;
  PyObject *off_o, *rows_o;
  if (!PyArg_ParseTuple(args, "OO", &off_o, &rows_o)) return NULL;
  PyArrayObject* off = (PyArrayObject*)PyArray_FROM_OTF(off_o, NPY_INT64, NPY_ARRAY_IN_ARRAY);
  if (!off) return NULL;
  PyArrayObject* rows = (PyArrayObject*)PyArray_FROM_OTF(rows_o, NPY_INT64, NPY_ARRAY_IN_ARRAY);
  if (!rows) { Py_DECREF(off); return NULL; }
  PyArrayObject* out = NULL;
  if (PyArray_NDIM(off) != 1 || PyArray_DIM(off, 0) < 1 || PyArray_NDIM(rows) != 2 || PyArray_DIM(rows, 1) != 2) {
    PyErr_SetString(PyExc_ValueError, "expected offsets [S+1] and rows [total, 2]");
    goto done;
  }
  {
    const npy_intp S = PyArray_DIM(off, 0) - 1, nrows = PyArray_DIM(rows, 0);
    const npy_int64* o = (const npy_int64*)PyArray_DATA(off);
    const npy_int64* r = (const npy_int64*)PyArray_DATA(rows);
    npy_intp total = 0;
    for (npy_intp j = 0; j < S; j++) {
      if (o[j] < 0 || o[j + 1] < o[j] || o[j + 1] > nrows) {
        PyErr_SetString(PyExc_ValueError, "offsets must be non-decreasing and within rows");
        goto done;
      }
      if (o[j + 1] - o[j] >= 2) total += o[j + 1] - o[j] - 1;
    }
    npy_intp dims[3] = {total, 2, 2};
    out = (PyArrayObject*)PyArray_EMPTY(3, dims, NPY_INT64, 0);
    if (!out) goto done;
    npy_int64* w = (npy_int64*)PyArray_DATA(out);
    for (npy_intp j = 0; j < S; j++)
      for (npy_int64 i = o[j]; i + 1 < o[j + 1]; i++) {
        w[0] = r[2 * i]; w[1] = r[2 * i + 1]; w[2] = r[2 * i + 2]; w[3] = r[2 * i + 3];
        w += 4;
      }
  }
done:
  Py_DECREF(off);
  Py_DECREF(rows);
  return (PyObject*)out;
}

static PyMethodDef methods[] = {
    {"filter_strand_segments_flat", filter_strand_segments_flat, METH_VARARGS,
     "Pairs of consecutive rows of every strand from (offsets [S+1], rows [total, 2]) -> int64 [pairs, 2, 2]."},
    {"filter_strand_list_segments", filter_strand_list_segments, METH_O,
     "Pairs of consecutive rows of every strand: object array of [n_j, 2] int64 arrays -> int64 [pairs, 2, 2]."},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_c_utils", "native half of the c_utils drop-in", -1, methods,
                                       NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__c_utils(void) {
  import_array();
  return PyModule_Create(&moduledef);
}
