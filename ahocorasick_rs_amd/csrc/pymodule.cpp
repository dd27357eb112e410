// pymodule.cpp -- C++ CPython extension `ahocorasick_rs_amd.ahocorasick_rs`.
//
// Host-side mirror of the reference's PyO3 module (/root/reference/src/lib.rs):
// same classes, signatures, defaults, keyword names and exceptions, sitting on
// the C ABI of include/acx.h instead of the `aho-corasick` crate.  The
// reference's host is Rust; Rust is not available in this image, so the shim is
// C++ (raw CPython C API).  Reference lines are cited at each entry point.
//
//   #[pymodule] fn ahocorasick_rs          src/lib.rs:438-445
//   class MatchKind / Implementation       src/lib.rs:92-128
//   class AhoCorasick                      src/lib.rs:29-33, 131-273
//   class BytesAhoCorasick                 src/lib.rs:360-435
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "acx.h"

namespace {

// ---------------------------------------------------------------------------
// enums: MatchKind, Implementation  (pyclass(eq) simple enums)
// ---------------------------------------------------------------------------
struct EnumObject {
    PyObject_HEAD
    int value;
    const char *qualname; // e.g. "MatchKind.Standard"
};

PyTypeObject *MatchKindType = nullptr;
PyTypeObject *ImplementationType = nullptr;

PyObject *enum_repr(PyObject *self) {
    return PyUnicode_FromString(reinterpret_cast<EnumObject *>(self)->qualname);
}
Py_hash_t enum_hash(PyObject *self) { return reinterpret_cast<EnumObject *>(self)->value + 1; }
PyObject *enum_int(PyObject *self) {
    return PyLong_FromLong(reinterpret_cast<EnumObject *>(self)->value);
}
PyObject *enum_richcompare(PyObject *a, PyObject *b, int op) {
    if (op != Py_EQ && op != Py_NE) Py_RETURN_NOTIMPLEMENTED;
    int eq;
    // pyclass(eq) without eq_int (src/lib.rs:93, 113): only members of the same enum compare
    // equal; `MatchKind.Standard == 0` is False
    if (Py_TYPE(a) != Py_TYPE(b)) Py_RETURN_NOTIMPLEMENTED;
    eq = reinterpret_cast<EnumObject *>(a)->value == reinterpret_cast<EnumObject *>(b)->value;
    if ((op == Py_EQ) == (eq != 0)) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

PyType_Slot enum_slots[] = {
    {Py_tp_repr, reinterpret_cast<void *>(enum_repr)},
    {Py_tp_hash, reinterpret_cast<void *>(enum_hash)},
    {Py_tp_richcompare, reinterpret_cast<void *>(enum_richcompare)},
    {Py_nb_int, reinterpret_cast<void *>(enum_int)},
    {Py_nb_index, reinterpret_cast<void *>(enum_int)},
    {0, nullptr},
};

// `full` must outlive the type (CPython keeps the pointer as tp_name)
PyTypeObject *make_enum(PyObject *module, const char *name, const char *full,
                        const char *const *variants, const char *const *qualnames, int n) {
    PyType_Spec spec = {full, sizeof(EnumObject), 0,
                        Py_TPFLAGS_DEFAULT | Py_TPFLAGS_DISALLOW_INSTANTIATION, enum_slots};
    PyTypeObject *tp = reinterpret_cast<PyTypeObject *>(PyType_FromSpec(&spec));
    if (!tp) return nullptr;
    for (int i = 0; i < n; i++) {
        EnumObject *o = PyObject_New(EnumObject, tp);
        if (!o) return nullptr;
        o->value = i;
        o->qualname = qualnames[i];
        if (PyObject_SetAttrString(reinterpret_cast<PyObject *>(tp), variants[i],
                                   reinterpret_cast<PyObject *>(o)) < 0)
            return nullptr;
        Py_DECREF(o);
    }
    // no public constructor
    if (PyModule_AddObject(module, name, reinterpret_cast<PyObject *>(tp)) < 0) return nullptr;
    Py_INCREF(tp);
    return tp;
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
// status code -> Python exception (SURVEY.md §8b "Errors")
PyObject *raise_acx(int rc) {
    const char *msg = acx_last_error();
    switch (rc) {
    case ACX_EINVAL: case ACX_EEMPTY: case ACX_EOVERLAP: case ACX_ETOOBIG:
        PyErr_SetString(PyExc_ValueError, msg); break; // src/lib.rs:36-39, 215, 406
    case ACX_ENOMEM: PyErr_SetString(PyExc_MemoryError, msg); break;
    default: PyErr_SetString(PyExc_RuntimeError, msg); break; // no device / HIP failure
    }
    return nullptr;
}

// `overlapping: bool` / `store_patterns: Option<bool>` (src/lib.rs:135, 229, 253, 422): PyO3
// extracts a real bool and raises TypeError for anything else
bool parse_bool(PyObject *o, const char *name, int *out) {
    if (!PyBool_Check(o)) {
        PyErr_Format(PyExc_TypeError, "argument '%s': '%.100s' object cannot be converted to 'PyBool'", name,
                     Py_TYPE(o)->tp_name);
        return false;
    }
    *out = o == Py_True;
    return true;
}

bool parse_matchkind(PyObject *o, int *out) {
    if (!o) { *out = ACX_MATCH_STANDARD; return true; }
    if (Py_TYPE(o) != MatchKindType) {
        PyErr_Format(PyExc_TypeError, "argument 'matchkind': '%.100s' object cannot be converted to 'MatchKind'",
                     Py_TYPE(o)->tp_name);
        return false;
    }
    *out = reinterpret_cast<EnumObject *>(o)->value;
    return true;
}

bool parse_implementation(PyObject *o, int *out) {
    if (!o || o == Py_None) { *out = ACX_IMPL_AUTO; return true; }
    if (Py_TYPE(o) != ImplementationType) {
        PyErr_Format(PyExc_TypeError,
                     "argument 'implementation': '%.100s' object cannot be converted to 'Implementation'",
                     Py_TYPE(o)->tp_name);
        return false;
    }
    *out = reinterpret_cast<EnumObject *>(o)->value;
    return true;
}

// PyBufferBytes::try_from, src/lib.rs:282-302
bool get_bytes_view(PyObject *obj, Py_buffer *view) {
    if (PyObject_GetBuffer(obj, view, PyBUF_FULL_RO) < 0) return false; // TypeError for non-buffers
    if (view->ndim > 1) {
        PyBuffer_Release(view);
        PyErr_SetString(PyExc_TypeError, "Only one-dimensional sequences are supported");
        return false;
    }
    // PyBuffer::<u8>::get (src/lib.rs:286) accepts unsigned one-byte items only: format "B"
    // (absent = "B"), optionally behind a byte-order character; 'b', 'c', '?' are rejected
    const char *fmt = view->format;
    if (fmt && (*fmt == '@' || *fmt == '=' || *fmt == '<' || *fmt == '>' || *fmt == '!')) fmt++;
    if (view->itemsize != 1 || (fmt && !(fmt[0] == 'B' && fmt[1] == 0))) {
        PyBuffer_Release(view);
        PyErr_SetString(PyExc_BufferError, "buffer contents are not compatible with u8");
        return false;
    }
    if (!PyBuffer_IsContiguous(view, 'C')) {
        PyBuffer_Release(view);
        PyErr_SetString(PyExc_TypeError, "Must be a contiguous sequence of bytes");
        return false;
    }
    return true;
}

PyObject *matches_to_list(const acx_match_t *m, uint64_t n) {
    PyObject *list = PyList_New((Py_ssize_t)n);
    if (!list) return nullptr;
    for (uint64_t i = 0; i < n; i++) {
        PyObject *t = PyTuple_New(3);
        if (!t) { Py_DECREF(list); return nullptr; }
        PyObject *a = PyLong_FromUnsignedLongLong(m[i].pattern);
        PyObject *b = PyLong_FromUnsignedLongLong(m[i].start);
        PyObject *c = PyLong_FromUnsignedLongLong(m[i].end);
        if (!a || !b || !c) { Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_DECREF(t); Py_DECREF(list); return nullptr; }
        PyTuple_SET_ITEM(t, 0, a); PyTuple_SET_ITEM(t, 1, b); PyTuple_SET_ITEM(t, 2, c);
        PyList_SET_ITEM(list, (Py_ssize_t)i, t);
    }
    return list;
}

// run acx_find with the GIL released (py.detach, src/lib.rs:238, 261, 433)
int find_nogil(acx_automaton_t *a, const uint8_t *hay, uint64_t len, int overlapping,
               int codepoints, acx_match_t **out, uint64_t *n) {
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = acx_find(a, hay, len, overlapping, codepoints, out, n);
    Py_END_ALLOW_THREADS
    return rc;
}

bool utf8_is_ascii(const char *s, Py_ssize_t n, PyObject *str) {
    (void)s; (void)n;
    return PyUnicode_IS_ASCII(str);
}

PyObject *info_dict(acx_automaton_t *a) {
    acx_info_t i;
    if (acx_automaton_info(a, &i) != ACX_OK) return raise_acx(ACX_EINVAL);
    return Py_BuildValue("{s:K,s:K,s:I,s:I,s:I,s:I,s:K,s:I,s:s,s:i,s:i,s:I}", "n_patterns",
                         (unsigned long long)i.n_patterns, "n_states", (unsigned long long)i.n_states,
                         "n_classes", i.n_classes, "stride", i.stride, "min_pattern_len",
                         i.min_pattern_len, "max_pattern_len", i.max_pattern_len, "table_bytes",
                         (unsigned long long)i.table_bytes, "lds_hot_rows", i.lds_hot_rows, "kernel",
                         i.kernel == ACX_KERNEL_PREFILTER ? "prefilter" : "dfa_walk", "match_kind",
                         i.match_kind, "device", i.device, "filter_q", i.filter_q);
}

// Replicas of an object's automaton on other devices (find_matches_as_indexes_batch(devices=[...])):
// built on first use, owned by the Python object.
typedef std::map<int, acx_automaton_t *> Replicas;

void free_replicas(Replicas *r) {
    if (!r) return;
    for (auto &kv : *r) acx_free_automaton(kv.second);
    delete r;
}

// devices (None, or a sequence of device ordinals) -> the handles the batch is sharded over
bool batch_handles(acx_automaton_t *a, PyObject *devices, Replicas **replicas, std::vector<acx_automaton_t *> *out) {
    out->clear();
    if (!devices || devices == Py_None) { out->push_back(a); return true; }
    PyObject *seq = PySequence_Fast(devices, "devices must be a sequence of device ordinals");
    if (!seq) return false;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    if (n == 0) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "devices must name at least one device");
        return false;
    }
    for (Py_ssize_t i = 0; i < n; i++) {
        const long d = PyLong_AsLong(PySequence_Fast_GET_ITEM(seq, i));
        if (d == -1 && PyErr_Occurred()) { Py_DECREF(seq); return false; }
        if (d == acx_automaton_device(a)) { out->push_back(a); continue; }
        if (!*replicas) *replicas = new Replicas();
        auto it = (*replicas)->find((int)d);
        if (it == (*replicas)->end()) {
            acx_automaton_t *r = nullptr;
            const int rc = acx_replicate(a, (int)d, &r);
            if (rc != ACX_OK) { Py_DECREF(seq); raise_acx(rc); return false; }
            it = (*replicas)->emplace((int)d, r).first;
        }
        out->push_back(it->second);
    }
    Py_DECREF(seq);
    return true;
}

// Batched search shared by both classes: `items` are str (utf8 = true) or
// buffers.  Returns list[list[tuple]].  devices: None = the object's own device; a sequence of
// ordinals = the batch is cut into that many contiguous ranges of haystacks, one host thread per
// device (acx_find_batch_multi).
PyObject *find_batch_impl(acx_automaton_t *a, PyObject *haystacks, int overlapping, bool utf8,
                          PyObject *devices, Replicas **replicas) {
    std::vector<acx_automaton_t *> handles;
    if (!batch_handles(a, devices, replicas, &handles)) return nullptr;
    PyObject *seq = PySequence_Fast(haystacks, "haystacks must be a sequence");
    if (!seq) return nullptr;
    Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    std::vector<uint64_t> off((size_t)n + 1, 0);
    std::vector<uint8_t> blob;
    bool all_ascii = true;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *it = PySequence_Fast_GET_ITEM(seq, i);
        if (utf8) {
            if (!PyUnicode_Check(it)) {
                PyErr_Format(PyExc_TypeError, "'%.100s' object cannot be converted to 'PyString'",
                             Py_TYPE(it)->tp_name);
                Py_DECREF(seq); return nullptr;
            }
            Py_ssize_t len; const char *s = PyUnicode_AsUTF8AndSize(it, &len);
            if (!s) { Py_DECREF(seq); return nullptr; }
            all_ascii = all_ascii && PyUnicode_IS_ASCII(it);
            blob.insert(blob.end(), s, s + len);
        } else {
            Py_buffer v;
            if (!get_bytes_view(it, &v)) { Py_DECREF(seq); return nullptr; }
            blob.insert(blob.end(), (const uint8_t *)v.buf, (const uint8_t *)v.buf + v.len);
            PyBuffer_Release(&v);
        }
        off[(size_t)i + 1] = blob.size();
    }
    std::vector<uint64_t> counts((size_t)n, 0);
    acx_match_t *m = nullptr; uint64_t total = 0;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = acx_find_batch_multi(handles.data(), (int)handles.size(), blob.data(), off.data(), (uint64_t)n, overlapping,
                              (utf8 && !all_ascii) ? 1 : 0, &m, &total, counts.data());
    Py_END_ALLOW_THREADS
    Py_DECREF(seq);
    if (rc != ACX_OK) return raise_acx(rc);
    PyObject *outer = PyList_New(n);
    uint64_t pos = 0;
    for (Py_ssize_t i = 0; outer && i < n; i++) {
        PyObject *inner = matches_to_list(m + pos, counts[(size_t)i]);
        if (!inner) { Py_CLEAR(outer); break; }
        PyList_SET_ITEM(outer, i, inner);
        pos += counts[(size_t)i];
    }
    acx_free_matches(m);
    return outer;
}

// ---------------------------------------------------------------------------
// AhoCorasick (str)
// ---------------------------------------------------------------------------
struct AcObject {
    PyObject_HEAD
    acx_automaton_t *ac;
    PyObject *patterns; // list[str] or NULL   (src/lib.rs:30-33 `patterns: Option<Vec<Py<PyString>>>`)
    Replicas *replicas; // automata on other devices (batch calls with devices=[...]), or NULL
};

void ac_dealloc(PyObject *self) {
    AcObject *o = reinterpret_cast<AcObject *>(self);
    if (o->ac) acx_free_automaton(o->ac);
    free_replicas(o->replicas);
    Py_XDECREF(o->patterns);
    PyTypeObject *tp = Py_TYPE(self);
    tp->tp_free(self);
    Py_DECREF(tp);
}

// src/lib.rs:134-224
PyObject *ac_new(PyTypeObject *type, PyObject *args, PyObject *kwargs) {
    static const char *kw[] = {"patterns", "matchkind", "store_patterns", "implementation", nullptr};
    PyObject *patterns = nullptr, *mk_o = nullptr, *store_o = Py_None, *impl_o = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|OOO:AhoCorasick", const_cast<char **>(kw),
                                     &patterns, &mk_o, &store_o, &impl_o))
        return nullptr;
    int mk, impl;
    if (!parse_matchkind(mk_o, &mk) || !parse_implementation(impl_o, &impl)) return nullptr;
    int store = -1; // None -> heuristic
    if (store_o != Py_None && !parse_bool(store_o, "store_patterns", &store)) return nullptr;
    PyObject *iter = PyObject_GetIter(patterns); // non-iterable -> TypeError (tests/test_ac.py:79-80)
    if (!iter) return nullptr;
    PyObject *kept = PyList_New(0);
    if (!kept) { Py_DECREF(iter); return nullptr; }
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(1, 0);
    uint64_t total_chars = 0;
    bool heuristic_store = true; // src/lib.rs:164-178: store while the running total <= 4096
    PyObject *item;
    bool failed = false;
    while ((item = PyIter_Next(iter))) {
        if (!PyUnicode_Check(item)) { // cast_into::<PyString>, src/lib.rs:147-150
            PyErr_Format(PyExc_TypeError, "'%.100s' object cannot be converted to 'PyString'",
                         Py_TYPE(item)->tp_name);
            Py_DECREF(item); failed = true; break;
        }
        Py_ssize_t len;
        const char *s = PyUnicode_AsUTF8AndSize(item, &len);
        if (!s) {
            // src/lib.rs:200: `extract::<PyBackedStr>().ok()` -> the reference silently stops
            // consuming patterns at a str that has no UTF-8 form (lone surrogates).
            PyErr_Clear();
            Py_DECREF(item);
            break;
        }
        if (len == 0) { // src/lib.rs:204-208
            PyErr_SetString(PyExc_ValueError, "You passed in an empty string as a pattern");
            Py_DECREF(item); failed = true; break;
        }
        blob.insert(blob.end(), s, s + len);
        off.push_back(blob.size());
        bool keep = store == 1;
        if (store == -1 && heuristic_store) {
            total_chars += (uint64_t)PyUnicode_GET_LENGTH(item);
            keep = true; // the reference pushes the pattern before testing the total
            if (total_chars > 4096) heuristic_store = false;
        }
        if (keep && PyList_Append(kept, item) < 0) { Py_DECREF(item); failed = true; break; }
        Py_DECREF(item);
    }
    Py_DECREF(iter);
    if (failed || PyErr_Occurred()) { Py_DECREF(kept); return nullptr; } // iterator errors propagate
    bool do_store = store == 1 || (store == -1 && heuristic_store);
    acx_automaton_t *ac = nullptr;
    int rc;
    blob.push_back(0);
    Py_BEGIN_ALLOW_THREADS // the reference yields the GIL while building, src/lib.rs:198
    rc = acx_build(blob.data(), off.data(), off.size() - 1, mk, impl, &ac);
    Py_END_ALLOW_THREADS
    if (rc != ACX_OK) { Py_DECREF(kept); return raise_acx(rc); }
    AcObject *self = reinterpret_cast<AcObject *>(type->tp_alloc(type, 0));
    if (!self) { acx_free_automaton(ac); Py_DECREF(kept); return nullptr; }
    self->ac = ac;
    self->replicas = nullptr;
    if (do_store) self->patterns = kept;
    else { self->patterns = nullptr; Py_DECREF(kept); }
    return reinterpret_cast<PyObject *>(self);
}

bool parse_find_args(PyObject *args, PyObject *kwargs, const char *fmt, PyObject **hay,
                     int *overlapping) {
    static const char *kw[] = {"haystack", "overlapping", nullptr};
    *overlapping = 0;
    PyObject *ov = nullptr;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, fmt, const_cast<char **>(kw), hay, &ov)) return false;
    return !ov || parse_bool(ov, "overlapping", overlapping);
}

bool str_view(PyObject *hay, const char **s, Py_ssize_t *len) {
    if (!PyUnicode_Check(hay)) {
        PyErr_Format(PyExc_TypeError, "argument 'haystack': '%.100s' object cannot be converted to 'PyString'",
                     Py_TYPE(hay)->tp_name);
        return false;
    }
    *s = PyUnicode_AsUTF8AndSize(hay, len);
    return *s != nullptr;
}

// src/lib.rs:229-249: code-point offsets
PyObject *ac_find_indexes(PyObject *self_, PyObject *args, PyObject *kwargs) {
    AcObject *self = reinterpret_cast<AcObject *>(self_);
    PyObject *hay; int overlapping;
    if (!parse_find_args(args, kwargs, "O|O:find_matches_as_indexes", &hay, &overlapping)) return nullptr;
    const char *s; Py_ssize_t len;
    if (!str_view(hay, &s, &len)) return nullptr;
    // ASCII haystack: byte offset == code-point index, skip the device fix-up
    int codepoints = utf8_is_ascii(s, len, hay) ? 0 : 1;
    acx_match_t *m = nullptr; uint64_t n = 0;
    int rc = find_nogil(self->ac, reinterpret_cast<const uint8_t *>(s), (uint64_t)len, overlapping,
                        codepoints, &m, &n);
    if (rc != ACX_OK) return raise_acx(rc);
    PyObject *list = matches_to_list(m, n);
    acx_free_matches(m);
    return list;
}

// src/lib.rs:253-272
PyObject *ac_find_strings(PyObject *self_, PyObject *args, PyObject *kwargs) {
    AcObject *self = reinterpret_cast<AcObject *>(self_);
    PyObject *hay; int overlapping;
    if (!parse_find_args(args, kwargs, "O|O:find_matches_as_strings", &hay, &overlapping)) return nullptr;
    const char *s; Py_ssize_t len;
    if (!str_view(hay, &s, &len)) return nullptr;
    acx_match_t *m = nullptr; uint64_t n = 0;
    int rc = find_nogil(self->ac, reinterpret_cast<const uint8_t *>(s), (uint64_t)len, overlapping,
                        0 /* byte offsets */, &m, &n);
    if (rc != ACX_OK) return raise_acx(rc);
    PyObject *list = PyList_New((Py_ssize_t)n);
    for (uint64_t i = 0; list && i < n; i++) {
        PyObject *item;
        if (self->patterns) { // clone_ref of the stored pattern, src/lib.rs:263-266
            item = PyList_GET_ITEM(self->patterns, (Py_ssize_t)m[i].pattern);
            Py_INCREF(item);
        } else {              // slice of the haystack by byte offsets, src/lib.rs:267-270
            item = PyUnicode_DecodeUTF8(s + m[i].start, (Py_ssize_t)(m[i].end - m[i].start), "strict");
            if (!item) { Py_CLEAR(list); break; }
        }
        PyList_SET_ITEM(list, (Py_ssize_t)i, item);
    }
    acx_free_matches(m);
    return list;
}

PyObject *ac_find_batch(PyObject *self_, PyObject *args, PyObject *kwargs) {
    static const char *kw[] = {"haystacks", "overlapping", "devices", nullptr};
    PyObject *hs, *ov = nullptr, *devs = nullptr; int overlapping = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|OO:find_matches_as_indexes_batch",
                                     const_cast<char **>(kw), &hs, &ov, &devs))
        return nullptr;
    if (ov && !parse_bool(ov, "overlapping", &overlapping)) return nullptr;
    AcObject *self = reinterpret_cast<AcObject *>(self_);
    return find_batch_impl(self->ac, hs, overlapping, true, devs, &self->replicas);
}

PyObject *ac_info(PyObject *self_, PyObject *) {
    return info_dict(reinterpret_cast<AcObject *>(self_)->ac);
}

PyMethodDef ac_methods[] = {
    {"find_matches_as_indexes", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(ac_find_indexes)),
     METH_VARARGS | METH_KEYWORDS,
     "Return matches as tuple of (index_into_patterns, start_index_in_haystack, "
     "end_index_in_haystack). If ``overlapping`` is ``False`` (the default), don't include "
     "overlapping results."},
    {"find_matches_as_strings", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(ac_find_strings)),
     METH_VARARGS | METH_KEYWORDS,
     "Return matches as list of patterns (i.e. strings). If ``overlapping`` is ``False`` (the "
     "default), don't include overlapping results."},
    {"find_matches_as_indexes_batch", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(ac_find_batch)),
     METH_VARARGS | METH_KEYWORDS,
     "[extension] one device pass over many haystacks; equals "
     "[self.find_matches_as_indexes(h, overlapping) for h in haystacks].  devices=[ordinals]: the batch "
     "is cut into contiguous ranges of haystacks, one per device (replicas of the automaton are built on "
     "first use), scanned side by side from one host thread each."},
    {"_info", ac_info, METH_NOARGS, "[extension] automaton / device facts as a dict."},
    {nullptr, nullptr, 0, nullptr},
};

PyType_Slot ac_slots[] = {
    {Py_tp_new, reinterpret_cast<void *>(ac_new)},
    {Py_tp_dealloc, reinterpret_cast<void *>(ac_dealloc)},
    {Py_tp_methods, ac_methods},
    {Py_tp_doc, const_cast<char *>(
        "Search for multiple pattern strings against a single haystack string.\n\n"
        "AhoCorasick(patterns, matchkind=MatchKind.Standard, store_patterns=None, implementation=None)")},
    {0, nullptr},
};

// ---------------------------------------------------------------------------
// BytesAhoCorasick
// ---------------------------------------------------------------------------
struct BacObject {
    PyObject_HEAD
    acx_automaton_t *ac;
    Replicas *replicas;
};

void bac_dealloc(PyObject *self) {
    BacObject *o = reinterpret_cast<BacObject *>(self);
    if (o->ac) acx_free_automaton(o->ac);
    free_replicas(o->replicas);
    PyTypeObject *tp = Py_TYPE(self);
    tp->tp_free(self);
    Py_DECREF(tp);
}

// src/lib.rs:369-413
PyObject *bac_new(PyTypeObject *type, PyObject *args, PyObject *kwargs) {
    static const char *kw[] = {"patterns", "matchkind", "implementation", nullptr};
    PyObject *patterns = nullptr, *mk_o = nullptr, *impl_o = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|OO:BytesAhoCorasick", const_cast<char **>(kw),
                                     &patterns, &mk_o, &impl_o))
        return nullptr;
    int mk, impl;
    if (!parse_matchkind(mk_o, &mk) || !parse_implementation(impl_o, &impl)) return nullptr;
    PyObject *iter = PyObject_GetIter(patterns);
    if (!iter) return nullptr;
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(1, 0);
    PyObject *item;
    bool failed = false;
    while ((item = PyIter_Next(iter))) {
        Py_buffer v;
        if (!get_bytes_view(item, &v)) { Py_DECREF(item); failed = true; break; }
        if (v.len == 0) { // src/lib.rs:386-389
            PyBuffer_Release(&v); Py_DECREF(item);
            PyErr_SetString(PyExc_ValueError, "You passed in an empty pattern");
            failed = true; break;
        }
        blob.insert(blob.end(), (const uint8_t *)v.buf, (const uint8_t *)v.buf + v.len);
        off.push_back(blob.size());
        PyBuffer_Release(&v); // no reference to the pattern objects is kept, src/lib.rs:350-351
        Py_DECREF(item);
    }
    Py_DECREF(iter);
    if (failed || PyErr_Occurred()) return nullptr;
    acx_automaton_t *ac = nullptr;
    int rc;
    blob.push_back(0);
    Py_BEGIN_ALLOW_THREADS
    rc = acx_build(blob.data(), off.data(), off.size() - 1, mk, impl, &ac);
    Py_END_ALLOW_THREADS
    if (rc != ACX_OK) return raise_acx(rc);
    BacObject *self = reinterpret_cast<BacObject *>(type->tp_alloc(type, 0));
    if (!self) { acx_free_automaton(ac); return nullptr; }
    self->ac = ac;
    self->replicas = nullptr;
    return reinterpret_cast<PyObject *>(self);
}

// ---- device-resident haystacks (extends the buffer adapter of src/lib.rs:276-340): an object that
// is no host buffer but exports __dlpack__ (a torch / cupy tensor in HBM) is searched where it lies
// -- no H2D copy.  DLPack structs (dlpack.h, ABI v0: the capsule "dltensor"):
struct DLDeviceC { int32_t device_type; int32_t device_id; };
struct DLDataTypeC { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensorC { void *data; DLDeviceC device; int32_t ndim; DLDataTypeC dtype; int64_t *shape; int64_t *strides; uint64_t byte_offset; };
struct DLManagedTensorC { DLTensorC dl_tensor; void *manager_ctx; void (*deleter)(DLManagedTensorC *); };
constexpr int32_t kDLCPU = 1, kDLCUDA = 2, kDLCUDAHost = 3, kDLROCM = 10, kDLROCMHost = 11;

// haystack -> (pointer, length, on_device).  Returns the capsule to release afterwards (new reference).
PyObject *dlpack_view(PyObject *hay, int want_device, const uint8_t **ptr, uint64_t *len, bool *on_device) {
    PyObject *cap = PyObject_CallMethod(hay, "__dlpack__", nullptr);
    if (!cap) return nullptr;
    DLManagedTensorC *mt = PyCapsule_IsValid(cap, "dltensor")
                               ? static_cast<DLManagedTensorC *>(PyCapsule_GetPointer(cap, "dltensor")) : nullptr;
    if (!mt) {
        Py_DECREF(cap);
        PyErr_SetString(PyExc_TypeError, "__dlpack__ did not return a 'dltensor' capsule");
        return nullptr;
    }
    const DLTensorC &t = mt->dl_tensor;
    const char *err = nullptr;
    uint64_t n = 1;
    for (int32_t d = 0; d < t.ndim; d++) n *= (uint64_t)t.shape[d];
    if (t.ndim > 1) err = "Only one-dimensional sequences are supported";                  // src/lib.rs:288-292
    else if (t.dtype.code != 1 || t.dtype.bits != 8 || t.dtype.lanes != 1) err = "buffer contents are not compatible with u8";
    else if (t.ndim == 1 && t.strides && t.shape[0] > 1 && t.strides[0] != 1) err = "Must be a contiguous sequence of bytes"; // :293-297
    const bool dev = t.device.device_type == kDLROCM || t.device.device_type == kDLCUDA;
    const bool host = t.device.device_type == kDLCPU || t.device.device_type == kDLROCMHost ||
                      t.device.device_type == kDLCUDAHost;
    if (!err && !dev && !host) err = "unsupported DLPack device type";
    if (!err && dev && t.device.device_id != want_device) {
        PyErr_Format(PyExc_ValueError, "haystack lives on device %d, the automaton on device %d",
                     (int)t.device.device_id, want_device);
        Py_DECREF(cap);
        return nullptr;
    }
    if (err) {
        PyErr_SetString(err[0] == 'b' ? PyExc_BufferError : PyExc_TypeError, err);
        Py_DECREF(cap);
        return nullptr;
    }
    *ptr = static_cast<const uint8_t *>(t.data) + t.byte_offset;
    *len = n;
    *on_device = dev;
    return cap;
}

void dlpack_release(PyObject *cap) { // we consumed the capsule: rename it and run the producer's deleter
    if (!cap) return;
    if (PyCapsule_IsValid(cap, "dltensor")) {
        DLManagedTensorC *mt = static_cast<DLManagedTensorC *>(PyCapsule_GetPointer(cap, "dltensor"));
        PyCapsule_SetName(cap, "used_dltensor");
        if (mt && mt->deleter) mt->deleter(mt);
    }
    Py_DECREF(cap);
}

// device-resident search -> list of tuples (the records come back with ONE D2H copy of the result)
PyObject *find_on_device(acx_automaton_t *a, const uint8_t *d_hay, uint64_t len, int overlapping) {
    acx_result_t *r = nullptr;
    std::vector<acx_match_t> m;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    // the producer's kernels may still be writing the tensor on its own stream: the library's
    // streams are non-blocking ones, so order the search behind everything queued on the device
    // (on the AUTOMATON's device -- the tensor's: dlpack_view checked that -- not the thread's current one)
    rc = acx_device_synchronize_on(acx_automaton_device(a));
    if (rc == ACX_OK) rc = acx_find_device(a, d_hay, len, nullptr, 0, 0, overlapping, 0, &r);
    if (rc == ACX_OK) {
        m.resize((size_t)acx_result_count(r));
        if (!m.empty()) rc = acx_result_copy(r, m.data());
    }
    if (r) acx_free_result(r);
    Py_END_ALLOW_THREADS
    if (rc != ACX_OK) return raise_acx(rc);
    return matches_to_list(m.data(), (uint64_t)m.size());
}

// src/lib.rs:422-434: byte offsets, no fix-up
PyObject *bac_find_indexes(PyObject *self_, PyObject *args, PyObject *kwargs) {
    BacObject *self = reinterpret_cast<BacObject *>(self_);
    PyObject *hay; int overlapping;
    if (!parse_find_args(args, kwargs, "O|O:find_matches_as_indexes", &hay, &overlapping)) return nullptr;
    if (!PyObject_CheckBuffer(hay) && PyObject_HasAttrString(hay, "__dlpack__")) {
        const uint8_t *p = nullptr; uint64_t len = 0; bool on_device = false;
        PyObject *cap = dlpack_view(hay, acx_automaton_device(self->ac), &p, &len, &on_device);
        if (!cap) return nullptr;
        PyObject *res;
        if (on_device) {
            res = find_on_device(self->ac, p, len, overlapping);
        } else { // (host memory behind DLPack: the ordinary entry point)
            acx_match_t *m = nullptr; uint64_t n = 0;
            const int rc = find_nogil(self->ac, p, len, overlapping, 0, &m, &n);
            res = rc != ACX_OK ? raise_acx(rc) : matches_to_list(m, n);
            if (rc == ACX_OK) acx_free_matches(m);
        }
        dlpack_release(cap);
        return res;
    }
    Py_buffer v;
    if (!get_bytes_view(hay, &v)) return nullptr;
    acx_match_t *m = nullptr; uint64_t n = 0;
    int rc = find_nogil(self->ac, (const uint8_t *)v.buf, (uint64_t)v.len, overlapping, 0, &m, &n);
    PyBuffer_Release(&v);
    if (rc != ACX_OK) return raise_acx(rc);
    PyObject *list = matches_to_list(m, n);
    acx_free_matches(m);
    return list;
}

PyObject *bac_find_batch(PyObject *self_, PyObject *args, PyObject *kwargs) {
    static const char *kw[] = {"haystacks", "overlapping", "devices", nullptr};
    PyObject *hs, *ov = nullptr, *devs = nullptr; int overlapping = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "O|OO:find_matches_as_indexes_batch",
                                     const_cast<char **>(kw), &hs, &ov, &devs))
        return nullptr;
    if (ov && !parse_bool(ov, "overlapping", &overlapping)) return nullptr;
    BacObject *self = reinterpret_cast<BacObject *>(self_);
    return find_batch_impl(self->ac, hs, overlapping, false, devs, &self->replicas);
}

PyObject *bac_info(PyObject *self_, PyObject *) {
    return info_dict(reinterpret_cast<BacObject *>(self_)->ac);
}

PyMethodDef bac_methods[] = {
    {"find_matches_as_indexes", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(bac_find_indexes)),
     METH_VARARGS | METH_KEYWORDS,
     "Return matches as tuple of (index_into_patterns, start_index_in_haystack, "
     "end_index_in_haystack). If ``overlapping`` is ``False`` (the default), don't include "
     "overlapping results."},
    {"find_matches_as_indexes_batch", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(bac_find_batch)),
     METH_VARARGS | METH_KEYWORDS,
     "[extension] one device pass over many haystacks; equals "
     "[self.find_matches_as_indexes(h, overlapping) for h in haystacks].  devices=[ordinals]: the batch "
     "is cut into contiguous ranges of haystacks, one per device (replicas of the automaton are built on "
     "first use), scanned side by side from one host thread each."},
    {"_info", bac_info, METH_NOARGS, "[extension] automaton / device facts as a dict."},
    {nullptr, nullptr, 0, nullptr},
};

PyType_Slot bac_slots[] = {
    {Py_tp_new, reinterpret_cast<void *>(bac_new)},
    {Py_tp_dealloc, reinterpret_cast<void *>(bac_dealloc)},
    {Py_tp_methods, bac_methods},
    {Py_tp_doc, const_cast<char *>(
        "Search for multiple pattern bytes against a single bytes haystack.\n\n"
        "BytesAhoCorasick(patterns, matchkind=MatchKind.Standard, implementation=None)")},
    {0, nullptr},
};

PyModuleDef moduledef = {
    PyModuleDef_HEAD_INIT, "ahocorasick_rs",
    "MI355X-native Aho-Corasick matcher behind the ahocorasick_rs API (HIP kernels via libacx_hip.so).",
    -1, nullptr, nullptr, nullptr, nullptr, nullptr,
};

} // namespace

extern "C" __attribute__((visibility("default"))) PyObject *PyInit_ahocorasick_rs(void) {
    PyObject *m = PyModule_Create(&moduledef);
    if (!m) return nullptr;
    static const char *mk_v[] = {"Standard", "LeftmostFirst", "LeftmostLongest"};
    static const char *mk_q[] = {"MatchKind.Standard", "MatchKind.LeftmostFirst",
                                 "MatchKind.LeftmostLongest"};
    static const char *im_v[] = {"NoncontiguousNFA", "ContiguousNFA", "DFA"};
    static const char *im_q[] = {"Implementation.NoncontiguousNFA", "Implementation.ContiguousNFA",
                                 "Implementation.DFA"};
    MatchKindType = make_enum(m, "MatchKind", "ahocorasick_rs.MatchKind", mk_v, mk_q, 3);
    ImplementationType = make_enum(m, "Implementation", "ahocorasick_rs.Implementation", im_v, im_q, 3);
    if (!MatchKindType || !ImplementationType) { Py_DECREF(m); return nullptr; }
    PyType_Spec ac_spec = {"ahocorasick_rs.AhoCorasick", sizeof(AcObject), 0, Py_TPFLAGS_DEFAULT, ac_slots};
    PyType_Spec bac_spec = {"ahocorasick_rs.BytesAhoCorasick", sizeof(BacObject), 0, Py_TPFLAGS_DEFAULT,
                            bac_slots};
    PyObject *ac_t = PyType_FromSpec(&ac_spec);
    PyObject *bac_t = PyType_FromSpec(&bac_spec);
    if (!ac_t || !bac_t || PyModule_AddObject(m, "AhoCorasick", ac_t) < 0 ||
        PyModule_AddObject(m, "BytesAhoCorasick", bac_t) < 0) {
        Py_XDECREF(ac_t); Py_XDECREF(bac_t); Py_DECREF(m);
        return nullptr;
    }
    return m;
}
