/* fastboard.c — CPython accelerator of the single-board `Board` façade (qtttgym_amd/board.py).
 *
 * Not a compute path: the rules run in libqttt_hip.so (qttt_board_op_host, include/qttt.h).  What lives here is the host
 * bookkeeping AROUND that call — writing a Board's attributes (.moves, .board, .qstructs; board.py:4-6 of the
 * reference) into the 64-byte pinned record and taking them back, in place, with the reference's aliasing behaviour —
 * which costs 5 us per call written in Python (profiles/r04/facade_latency.json: make_move 15.2 us for a 10.2 us device
 * round trip) and well under 1 us here.  board.py keeps the same logic in Python (`_Staging.pack`, `Board._adopt`):
 * this module must agree with it attribute for attribute (tests/test_facade_gpu.py runs the golden episodes through
 * both), and declines (returns -100) whenever an attribute is not the plain list / tuple / set it expects, so that the
 * Python path handles — and reports — the unusual case.
 *
 *   gcc -O2 -shared -fPIC -I/usr/include/python3.10 qtttgym_amd/csrc/fastboard.c -o qtttgym_amd/_fastboard.so
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef int (*board_op_host_fn)(const void *, void *, int64_t, void *);

static board_op_host_fn g_op_host = NULL;
static uint8_t *g_in = NULL, *g_out = NULL;
static PyObject *s_moves, *s_board, *s_qstructs, *s_win, *s_update;

#define DECLINE (-100)

/* small non-negative int from a Python int; -1 if it is not one (no exception left behind) */
static long small_int(PyObject *o) {
    if (!PyLong_Check(o)) return -1;
    long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
    return v;
}

/* a set of squares -> its 9-bit mask (public iteration API only: PyObject_GetIter / PyIter_Next).  0 ok / DECLINE */
static int set_mask(PyObject *s, unsigned *mask_out) {
    unsigned mask = 0;
    if (PySet_GET_SIZE(s) > 9) return DECLINE;        /* not a set of squares: the Python path reports it */
    PyObject *it = PyObject_GetIter(s);
    if (!it) { PyErr_Clear(); return DECLINE; }
    PyObject *key;
    while ((key = PyIter_Next(it)) != NULL) {
        long x = small_int(key);
        Py_DECREF(key);
        if (x < 0) { Py_DECREF(it); return DECLINE; }  /* (a negative square: the Python path raises what Python raises) */
        if (x < 16) mask |= 1u << x;
    }
    Py_DECREF(it);
    if (PyErr_Occurred()) { PyErr_Clear(); return DECLINE; }
    *mask_out = mask;
    return 0;
}

/* board's attributes + the move -> the first 41 bytes of the in record (board.py: _Staging.pack).  0 ok / DECLINE */
static int pack(PyObject *board, int op, int lo, int hi, int bit, int drop_last, PyObject **moves_o, PyObject **board_o,
                PyObject **qs_o) {
    PyObject *moves = PyObject_GetAttr(board, s_moves), *bd = PyObject_GetAttr(board, s_board),
             *qs = PyObject_GetAttr(board, s_qstructs);
    int rc = DECLINE;
    uint8_t rec[41];
    if (!moves || !bd || !qs) { PyErr_Clear(); goto done; }
    if (!PyList_CheckExact(moves) || !PyList_CheckExact(bd) || !PyList_CheckExact(qs)) goto done;
    if (PyList_GET_SIZE(bd) < 9) goto done;
    Py_ssize_t n = PyList_GET_SIZE(moves) - (drop_last ? 1 : 0);
    if (n < 0) goto done;
    if (n > 9) n = 9;
    memset(rec, 0xFF, 18);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *m = PyList_GET_ITEM(moves, i);
        PyObject *a, *b;
        if (PyTuple_CheckExact(m) && PyTuple_GET_SIZE(m) >= 2) { a = PyTuple_GET_ITEM(m, 0); b = PyTuple_GET_ITEM(m, 1); }
        else if (PyList_CheckExact(m) && PyList_GET_SIZE(m) >= 2) { a = PyList_GET_ITEM(m, 0); b = PyList_GET_ITEM(m, 1); }
        else goto done;
        if (!PyLong_Check(a) || !PyLong_Check(b)) goto done;
        long va = PyLong_AsLong(a), vb = PyLong_AsLong(b);
        if ((va == -1 || vb == -1) && PyErr_Occurred()) { PyErr_Clear(); goto done; }
        rec[2 * i] = (uint8_t)(va & 255);
        rec[2 * i + 1] = (uint8_t)(vb & 255);
    }
    rec[18] = (uint8_t)n;
    for (int v = 0; v < 9; ++v) {
        PyObject *x = PyList_GET_ITEM(bd, v);
        if (!PyLong_Check(x)) goto done;
        long xv = PyLong_AsLong(x);
        if (xv == -1 && PyErr_Occurred()) { PyErr_Clear(); goto done; }
        rec[19 + v] = (uint8_t)(xv & 255);
    }
    Py_ssize_t nq = PyList_GET_SIZE(qs);
    if (nq > 4) nq = 4;
    rec[28] = (uint8_t)nq;
    rec[29] = (uint8_t)op;
    memset(rec + 30, 0, 8);
    for (Py_ssize_t k = 0; k < nq; ++k) {
        PyObject *s = PyList_GET_ITEM(qs, k);
        if (!PyAnySet_CheckExact(s)) goto done;
        unsigned mask = 0;
        if (set_mask(s, &mask) != 0) goto done;
        rec[30 + 2 * k] = (uint8_t)(mask & 255);
        rec[31 + 2 * k] = (uint8_t)(mask >> 8);
    }
    rec[38] = (uint8_t)(lo & 255);
    rec[39] = (uint8_t)(hi & 255);
    rec[40] = (uint8_t)(bit & 255);
    memcpy(g_in, rec, 41);
    rc = 0;
done:
    if (rc == 0) { *moves_o = moves; *board_o = bd; *qs_o = qs; }
    else { Py_XDECREF(moves); Py_XDECREF(bd); Py_XDECREF(qs); }
    return rc;
}

static PyObject *set_of_mask(unsigned mask) {
    PyObject *s = PySet_New(NULL);
    if (!s) return NULL;
    for (int v = 0; v < 9; ++v)
        if (mask >> v & 1u) {
            PyObject *x = PyLong_FromLong(v);
            if (!x || PySet_Add(s, x) < 0) { Py_XDECREF(x); Py_DECREF(s); return NULL; }
            Py_DECREF(x);
        }
    return s;
}

static long i8(uint8_t x) { return x > 127 ? (long)x - 256 : (long)x; }

/* the out record -> the attributes, IN PLACE (board.py: Board._adopt; the reference's own aliasing: board.py:19,25
 * append to .moves, :53-54 write into .board, :56-69 pop / assign / append on .qstructs).  0 ok / -1 with an exception */
static int adopt(PyObject *board, PyObject *moves, PyObject *bd, PyObject *qs, const uint8_t *r) {
    int n = r[18] < 9 ? r[18] : 9;
    PyObject *nm = PyList_New(n);
    if (!nm) return -1;
    for (int i = 0; i < n; ++i) {
        PyObject *t = Py_BuildValue("(iii)", (int)r[2 * i], (int)r[2 * i + 1], i);
        if (!t) { Py_DECREF(nm); return -1; }
        PyList_SET_ITEM(nm, i, t);
    }
    int rc = PyList_SetSlice(moves, 0, PyList_GET_SIZE(moves), nm);
    Py_DECREF(nm);
    if (rc < 0) return -1;
    PyObject *nb = PyList_New(9);
    if (!nb) return -1;
    for (int v = 0; v < 9; ++v) {
        PyObject *x = PyLong_FromLong(i8(r[19 + v]));
        if (!x) { Py_DECREF(nb); return -1; }
        PyList_SET_ITEM(nb, v, x);
    }
    rc = PyList_SetSlice(bd, 0, PyList_GET_SIZE(bd), nb);
    Py_DECREF(nb);
    if (rc < 0) return -1;
    /* qstructs: new sets from the masks, then the OBJECTS of the old list are kept where the reference keeps them */
    int nq = r[28] < 4 ? r[28] : 4;
    PyObject *nw = PyList_New(nq);
    if (!nw) return -1;
    for (int k = 0; k < nq; ++k) {
        PyObject *s = set_of_mask(((unsigned)r[30 + 2 * k] | ((unsigned)r[31 + 2 * k] << 8)) & 511u);
        if (!s) { Py_DECREF(nw); return -1; }
        PyList_SET_ITEM(nw, k, s);
    }
    Py_ssize_t n_old = PyList_GET_SIZE(qs);
    if (n_old > 0) {
        PyObject *spare = PyList_GetSlice(qs, 0, n_old);
        if (!spare) { Py_DECREF(nw); return -1; }
        for (int k = 0; k < nq; ++k) {                    /* an untouched component stays the object it was */
            PyObject *s = PyList_GET_ITEM(nw, k);
            for (Py_ssize_t j = 0; j < PyList_GET_SIZE(spare); ++j) {
                PyObject *t = PyList_GET_ITEM(spare, j);
                int eq = PyObject_RichCompareBool(t, s, Py_EQ);
                if (eq < 0) { Py_DECREF(spare); Py_DECREF(nw); return -1; }
                if (eq) {
                    Py_INCREF(t);
                    if (PyList_SetItem(nw, k, t) < 0 || PySequence_DelItem(spare, j) < 0) {   /* SetItem steals t, drops s */
                        Py_DECREF(spare); Py_DECREF(nw); return -1;
                    }
                    break;
                }
            }
        }
        if (nq == n_old) {                                /* "add to sets" grows its set in place (board.py:68-69) */
            for (int k = 0; k < nq; ++k) {
                PyObject *s = PyList_GET_ITEM(nw, k);
                int is_old = 0;
                for (Py_ssize_t j = 0; j < n_old && !is_old; ++j) is_old = PyList_GET_ITEM(qs, j) == s;
                if (is_old) continue;
                Py_ssize_t hit = -1, hits = 0;
                for (Py_ssize_t j = 0; j < PyList_GET_SIZE(spare); ++j) {
                    int lt = PyObject_RichCompareBool(PyList_GET_ITEM(spare, j), s, Py_LT);
                    if (lt < 0) { Py_DECREF(spare); Py_DECREF(nw); return -1; }
                    if (lt) { hit = j; ++hits; }
                }
                if (hits == 1) {
                    PyObject *t = PyList_GET_ITEM(spare, hit);
                    PyObject *u = PyObject_CallMethodObjArgs(t, s_update, s, NULL);
                    if (!u) { Py_DECREF(spare); Py_DECREF(nw); return -1; }
                    Py_DECREF(u);
                    Py_INCREF(t);
                    if (PyList_SetItem(nw, k, t) < 0 || PySequence_DelItem(spare, hit) < 0) {
                        Py_DECREF(spare); Py_DECREF(nw); return -1;
                    }
                }
            }
        }
        Py_DECREF(spare);
    }
    rc = PyList_SetSlice(qs, 0, PyList_GET_SIZE(qs), nw);
    Py_DECREF(nw);
    if (rc < 0) return -1;
    /* check_win of the new state comes back in the same record: remembered, keyed by the board it belongs to */
    PyObject *key = PyList_AsTuple(bd);
    if (!key) return -1;
    PyObject *w = Py_BuildValue("(Nll)", key, i8(r[49]), i8(r[50]));
    if (!w) return -1;
    rc = PyObject_SetAttr(board, s_win, w);
    Py_DECREF(w);
    return rc;
}

/* init(address of qttt_board_op_host, address of the pinned in record, address of the pinned out record) */
static PyObject *fb_init(PyObject *self, PyObject *args) {
    unsigned long long fn, pin, pout;
    if (!PyArg_ParseTuple(args, "KKK", &fn, &pin, &pout)) return NULL;
    g_op_host = (board_op_host_fn)(uintptr_t)fn;
    g_in = (uint8_t *)(uintptr_t)pin;
    g_out = (uint8_t *)(uintptr_t)pout;
    Py_RETURN_NONE;
}

/* board_op(board, op, lo, hi, bit, drop_last_move, stream) -> 0 done | -100 declined (use the Python path) | rc of the
 * library (> 0 hipError_t, < 0 argument error).  The CALLER serialises the one staging record (board.py holds
 * _Staging.lock around this function); the GIL is released for the device round trip alone, like the ctypes call of the
 * Python path — normally ~5 us of polling, but a stream synchronise when the poll gives up behind a long kernel. */
static PyObject *fb_board_op(PyObject *self, PyObject *args) {
    PyObject *board;
    int op, lo, hi, bit, drop_last;
    unsigned long long stream;
    if (!PyArg_ParseTuple(args, "OiiiipK", &board, &op, &lo, &hi, &bit, &drop_last, &stream)) return NULL;
    if (!g_op_host || !g_in || !g_out) { PyErr_SetString(PyExc_RuntimeError, "_fastboard.init() was not called"); return NULL; }
    PyObject *moves, *bd, *qs;
    if (pack(board, op, lo, hi, bit, drop_last, &moves, &bd, &qs) != 0) return PyLong_FromLong(DECLINE);
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = g_op_host(g_in, g_out, 1, (void *)(uintptr_t)stream);
    Py_END_ALLOW_THREADS
    if (rc == 0) {
        uint8_t r[64];
        memcpy(r, g_out, 64);
        if (adopt(board, moves, bd, qs, r) < 0) { Py_DECREF(moves); Py_DECREF(bd); Py_DECREF(qs); return NULL; }
    }
    Py_DECREF(moves); Py_DECREF(bd); Py_DECREF(qs);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"init", fb_init, METH_VARARGS, "init(fn_address, in_record_address, out_record_address)"},
    {"board_op", fb_board_op, METH_VARARGS, "board_op(board, op, lo, hi, bit, drop_last_move, stream) -> rc"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fastboard", "CPython accelerator of qtttgym_amd.board.Board's device round trip", -1, methods};

PyMODINIT_FUNC PyInit__fastboard(void) {
    s_moves = PyUnicode_InternFromString("moves");
    s_board = PyUnicode_InternFromString("board");
    s_qstructs = PyUnicode_InternFromString("qstructs");
    s_win = PyUnicode_InternFromString("_win");
    s_update = PyUnicode_InternFromString("update");
    if (!s_moves || !s_board || !s_qstructs || !s_win || !s_update) return NULL;
    return PyModule_Create(&module);
}
