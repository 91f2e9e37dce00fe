/* _kctfast -- CPython call glue for the one entry point a per-record loop hammers:
 *
 *     for rec in records: table.consume(rec)          (the reference's documented loop, README.md:96-98)
 *
 * ctypes spends ~0.6 us per call on argument conversion; pyo3, which the reference binds with (lib.rs:545-546), spends
 * ~0.1 us.  This module is the pyo3-sized shim for Python callers of the C ABI: it is handed the ADDRESS of
 * kct_consume (include/kct.h) by oxli_amd/table.py and forwards (handle, str | bytes, skip_bad) to it.  No k-mer
 * arithmetic lives here; without it table.py uses ctypes, with identical results.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef int (*kct_consume_fn)(void *t, const char *seq, size_t len, int skip_bad, uint64_t *n_out);
typedef int (*kct_will_defer_fn)(const void *t, size_t len, int skip_bad);
static kct_consume_fn g_consume = NULL;
static kct_will_defer_fn g_will_defer = NULL;

/* bind(address of kct_consume, address of kct_consume_will_defer) */
static PyObject *fast_bind(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 2) { PyErr_SetString(PyExc_TypeError, "bind(consume_address, will_defer_address)"); return NULL; }
    const unsigned long long a = PyLong_AsUnsignedLongLong(args[0]), b = PyLong_AsUnsignedLongLong(args[1]);
    if ((a == (unsigned long long)-1 || b == (unsigned long long)-1) && PyErr_Occurred()) return NULL;
    g_consume = (kct_consume_fn)(uintptr_t)a;
    g_will_defer = (kct_will_defer_fn)(uintptr_t)b;
    Py_RETURN_NONE;
}

/* consume(handle: int, seq: str | bytes, skip_bad) -> n (int) on KCT_OK, (status, n) otherwise, None for other seq types */
static PyObject *fast_consume(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "consume(handle, seq, skip_bad)"); return NULL; }
    if (!g_consume) { PyErr_SetString(PyExc_RuntimeError, "_kctfast is not bound to libkct_hip.so"); return NULL; }
    const unsigned long long h = PyLong_AsUnsignedLongLong(args[0]);
    if (h == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
    const char *p;
    Py_ssize_t len;
    if (PyUnicode_Check(args[1])) {
        p = PyUnicode_AsUTF8AndSize(args[1], &len);  /* the UTF-8 bytes of the str, as the reference sees them (lib.rs:548, 577) */
        if (!p) return NULL;
    } else if (PyBytes_Check(args[1])) {
        if (PyBytes_AsStringAndSize(args[1], (char **)&p, &len) < 0) return NULL;
    } else Py_RETURN_NONE;
    const int skip = PyObject_IsTrue(args[2]);
    if (skip < 0) return NULL;
    uint64_t n = 0;
    int st;
    /* A call that only appends to deferred mode's buffer (~60 ns) keeps the GIL: releasing and re-taking it would cost as much
     * again.  A call that will run a device pass -- the buffer is full, error mode, deferred mode off -- releases it, as ctypes
     * does: other Python threads (torch.distributed's watchdog, a feeder of ANOTHER table) must not stall behind the pass.  A
     * thread that calls into THIS table meanwhile is turned away by the library (KCT_ERR_BUSY -> RuntimeError("Already
     * borrowed"), as pyo3 does for a `&mut self` method): a table takes one caller at a time.  The str / bytes object stays
     * alive through args[1]; its buffer is immutable. */
    if (g_will_defer && g_will_defer((const void *)(uintptr_t)h, (size_t)len, skip)) st = g_consume((void *)(uintptr_t)h, p, (size_t)len, skip, &n);
    else {
        Py_BEGIN_ALLOW_THREADS
        st = g_consume((void *)(uintptr_t)h, p, (size_t)len, skip, &n);
        Py_END_ALLOW_THREADS
    }
    if (st == 0) return PyLong_FromUnsignedLongLong(n);
    return Py_BuildValue("(iK)", st, (unsigned long long)n);
}

/* csr(seqs: list | tuple of str | bytes) -> (data: bytes, offsets: bytes of len(seqs) + 1 native u64), or None if an item is
 * of another type.  The concatenation consume_batch() hands to kct_consume_batch, built without a Python-level pass per
 * record (a list comprehension of .encode() calls, a cumsum of a list of lengths and a join were 0.25 s per million reads). */
static PyObject *fast_csr(PyObject *self, PyObject *arg) {
    (void)self;
    PyObject *fastseq = PySequence_Fast(arg, "csr(seqs): a list or tuple");
    if (!fastseq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fastseq);
    PyObject **items = PySequence_Fast_ITEMS(fastseq);
    PyObject *offs = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)((n + 1) * sizeof(uint64_t)));
    if (!offs) { Py_DECREF(fastseq); return NULL; }
    uint64_t *o = (uint64_t *)PyBytes_AS_STRING(offs);
    uint64_t total = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_ssize_t len;
        if (PyUnicode_Check(items[i])) {
            if (!PyUnicode_AsUTF8AndSize(items[i], &len)) { Py_DECREF(offs); Py_DECREF(fastseq); return NULL; }
        } else if (PyBytes_Check(items[i])) len = PyBytes_GET_SIZE(items[i]);
        else { Py_DECREF(offs); Py_DECREF(fastseq); Py_RETURN_NONE; }
        o[i] = total;
        total += (uint64_t)len;
    }
    o[n] = total;
    PyObject *data = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (!data) { Py_DECREF(offs); Py_DECREF(fastseq); return NULL; }
    char *d = PyBytes_AS_STRING(data);
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_ssize_t len;
        const char *p = PyUnicode_Check(items[i]) ? PyUnicode_AsUTF8AndSize(items[i], &len) : (len = PyBytes_GET_SIZE(items[i]), PyBytes_AS_STRING(items[i]));
        memcpy(d + o[i], p, (size_t)len);
    }
    Py_DECREF(fastseq);
    PyObject *r = PyTuple_Pack(2, data, offs);
    Py_DECREF(data);
    Py_DECREF(offs);
    return r;
}

static PyMethodDef methods[] = {
    {"csr", fast_csr, METH_O, "csr(seqs) -> (data, offsets)"},
    {"bind", (PyCFunction)(void (*)(void))fast_bind, METH_FASTCALL, "bind(address of kct_consume, address of kct_consume_will_defer)"},
    {"consume", (PyCFunction)(void (*)(void))fast_consume, METH_FASTCALL, "consume(handle, seq, skip_bad)"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_kctfast", "call glue for kct_consume", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__kctfast(void) { return PyModule_Create(&moduledef); }
