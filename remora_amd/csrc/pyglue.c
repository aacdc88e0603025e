/* _pyglue.so - the CPython side of the batch staging (loaded with ctypes.PyDLL, i.e. called with the GIL held).
 *
 * DeviceReads gathers a batch of RemoraRead objects into one pinned buffer with rmr_pack_reads (native threads); what was
 * left in Python was the walk over the reads that collects, per read, three array addresses, their sizes and two floats -
 * 5-8 us a read in the interpreter (a dozen attribute look-ups, ascontiguousarray and __array_interface__ calls), 10-16 ms
 * for the 2048 reads of a batch and the longest bar of the 16-bit reads pipeline (profiles/r05_reads_timeline.md).  Here the
 * same walk through the C API: attribute look-ups by interned name and the buffer protocol, about 0.3 us a read.
 *
 * Not part of the C ABI of libremora_hip.so (include/remora_hip.h): it exists only because the host is Python. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static int int_format(const char *f, Py_ssize_t itemsize, const char *allowed) {
    if (!f) return 0;
    if (*f == '<' || *f == '=' || *f == '@' || *f == '|') ++f;
    if (!*f || f[1]) return 0;
    if (!strchr(allowed, *f)) return 0;
    return itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8;
}

/* Returns the number of reads collected (= n) on success; -2 when a read's arrays are not in the layout the native gather
 * takes as it is (int16 C-contiguous dacs, int64 C-contiguous mapping of n_bases + 1 entries, integer C-contiguous bases):
 * the caller then walks the batch in Python, which converts or refuses with the reference's messages; -1 with a Python
 * exception set (not a sequence, missing attribute).
 * `keep` (a list): every attribute object whose buffer address is handed back is appended to it - the CALLER holds that list
 * until rmr_pack_reads has copied from the addresses, so an array that a property computed for this call, or that another
 * thread replaces on the read meanwhile, stays alive (the buffer itself is released here: numpy arrays do not move). */
int64_t rmr_py_collect_reads(PyObject *reads, int64_t n, void **p_dacs, int64_t *sig_n, void **p_maps, void **p_seqs,
                             int64_t *seq_n, int32_t *seq_itemsize, double *shift, double *scale, PyObject *keep) {
    if (!keep || !PyList_Check(keep)) {
        PyErr_SetString(PyExc_TypeError, "rmr_py_collect_reads: keep must be a list");
        return -1;
    }
    static PyObject *s_dacs, *s_map, *s_seq, *s_shift, *s_scale;
    if (!s_dacs) {
        s_dacs = PyUnicode_InternFromString("dacs");
        s_map = PyUnicode_InternFromString("seq_to_sig_map");
        s_seq = PyUnicode_InternFromString("int_seq");
        s_shift = PyUnicode_InternFromString("shift");
        s_scale = PyUnicode_InternFromString("scale");
    }
    PyObject *fast = PySequence_Fast(reads, "reads must be a sequence");
    if (!fast) return -1;
    if (PySequence_Fast_GET_SIZE(fast) != n) {
        Py_DECREF(fast);
        PyErr_SetString(PyExc_ValueError, "rmr_py_collect_reads: length mismatch");
        return -1;
    }
    int64_t rc = n;
    for (int64_t i = 0; i < n && rc == n; ++i) {
        PyObject *r = PySequence_Fast_GET_ITEM(fast, i);
        PyObject *names[3] = {s_dacs, s_map, s_seq};
        const char *allowed[3] = {"h", "lq", "bBhHiIlLqQ"};
        void *ptr[3];
        Py_ssize_t count[3], isz[3];
        for (int k = 0; k < 3 && rc == n; ++k) {
            PyObject *a = PyObject_GetAttr(r, names[k]);
            if (!a) { rc = -1; break; }
            Py_buffer v;
            if (PyObject_GetBuffer(a, &v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
                PyErr_Clear();
                rc = -2;
            } else {
                if (!int_format(v.format, v.itemsize, allowed[k]) || (k == 0 && v.itemsize != 2) || (k == 1 && v.itemsize != 8)) rc = -2;
                ptr[k] = v.buf;
                isz[k] = v.itemsize;
                count[k] = v.itemsize ? v.len / v.itemsize : 0;
                PyBuffer_Release(&v);
                if (rc == n && PyList_Append(keep, a) != 0) rc = -1; /* ownership of the array travels with its address */
            }
            Py_DECREF(a);
        }
        if (rc != n) break;
        if (count[1] != count[2] + 1) { rc = -2; break; }
        p_dacs[i] = ptr[0]; sig_n[i] = count[0];
        p_maps[i] = ptr[1];
        p_seqs[i] = ptr[2]; seq_n[i] = count[2]; seq_itemsize[i] = (int32_t)isz[2];
        PyObject *f = PyObject_GetAttr(r, s_shift);
        if (!f) { rc = -1; break; }
        shift[i] = PyFloat_AsDouble(f);
        Py_DECREF(f);
        f = PyObject_GetAttr(r, s_scale);
        if (!f) { rc = -1; break; }
        scale[i] = PyFloat_AsDouble(f);
        Py_DECREF(f);
        if (PyErr_Occurred()) { rc = -1; break; }
    }
    Py_DECREF(fast);
    return rc;
}
