/* _hostcopy: the host-side copies of the batched drivers that numpy only does on one core.
 *
 *   concat_rows(arrays, out, threads=8)
 *       `arrays`: a sequence of C-contiguous float64 buffers (the caller's [T_r, D] templates); `out`: a writable
 *       C-contiguous buffer of exactly their total size.  The pieces are copied back to back into `out` -- what
 *       np.concatenate(arrays, out=out) does -- by `threads` threads with the GIL released.  62 MB in 2 000 pieces
 *       (train_words on ten words x 200 templates): 5.7 ms with numpy on one core; sublists handed to Python threads are
 *       SLOWER (a GIL hand-over per piece, profiles/r05c_host_copy_probe.txt).
 *       Raises TypeError for anything that is not a C-contiguous float64 buffer -- of `out`'s row width, when `out` is a
 *       matrix -- (the caller then takes numpy's route, with its casts and its errors) and ValueError when the sizes do not
 *       add up.
 *
 * Plain buffer protocol: no numpy headers.  Built by speech-recognition_amd/build.py with the system compiler.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <string.h>
#include <stdlib.h>

typedef struct {
    const char* src;
    char* dst;
    size_t bytes;
} piece_t;

typedef struct {
    const piece_t* pieces;
    Py_ssize_t first, last;
} job_t;

static void* copy_job(void* arg) {
    const job_t* j = (const job_t*)arg;
    for (Py_ssize_t i = j->first; i < j->last; ++i)
        if (j->pieces[i].bytes) memcpy(j->pieces[i].dst, j->pieces[i].src, j->pieces[i].bytes);
    return NULL;
}

static int is_float64(const Py_buffer* v) {
    if (v->itemsize != 8) return 0;
    if (!v->format) return 0;
    const char* f = v->format;
    if (*f == '@' || *f == '=' || *f == '<') ++f;              /* native / little-endian doubles */
    return f[0] == 'd' && f[1] == '\0';
}

static PyObject* concat_rows(PyObject* self, PyObject* args, PyObject* kwargs) {
    static char* names[] = {"arrays", "out", "threads", NULL};
    PyObject *seq_in, *out_obj;
    int threads = 8;
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "OO|i", names, &seq_in, &out_obj, &threads)) return NULL;
    PyObject* seq = PySequence_Fast(seq_in, "concat_rows: `arrays` must be a sequence");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer out;
    if (PyObject_GetBuffer(out_obj, &out, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
        Py_DECREF(seq);
        return NULL;
    }
    Py_buffer* views = (Py_buffer*)calloc(n > 0 ? (size_t)n : 1, sizeof(Py_buffer));
    piece_t* pieces = (piece_t*)calloc(n > 0 ? (size_t)n : 1, sizeof(piece_t));
    Py_ssize_t got = 0;
    PyObject* result = NULL;
    if (!views || !pieces) { PyErr_NoMemory(); goto done; }
    if (!is_float64(&out)) { PyErr_SetString(PyExc_TypeError, "concat_rows: `out` must be a float64 buffer"); goto done; }
    {
        size_t at = 0;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* item = PySequence_Fast_GET_ITEM(seq, i);
            if (PyObject_GetBuffer(item, &views[i], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
                PyErr_Clear();
                PyErr_Format(PyExc_TypeError, "concat_rows: item %zd is not a C-contiguous buffer", i);
                goto done;
            }
            got = i + 1;
            if (!is_float64(&views[i])) { PyErr_Format(PyExc_TypeError, "concat_rows: item %zd is not float64", i); goto done; }
            /* a matrix as `out`: every piece a matrix of the same width (sizes that merely ADD UP would pass the byte count) */
            if (out.ndim == 2 && out.shape && (views[i].ndim != 2 || !views[i].shape || views[i].shape[1] != out.shape[1])) {
                PyErr_Format(PyExc_TypeError, "concat_rows: item %zd is not a [rows, %zd] array", i, out.shape[1]);
                goto done;
            }
            if ((size_t)views[i].len > (size_t)out.len - at) { PyErr_SetString(PyExc_ValueError, "concat_rows: the pieces are larger than `out`"); goto done; }
            pieces[i].src = (const char*)views[i].buf;
            pieces[i].dst = (char*)out.buf + at;
            pieces[i].bytes = (size_t)views[i].len;
            at += (size_t)views[i].len;
        }
        if (at != (size_t)out.len) { PyErr_SetString(PyExc_ValueError, "concat_rows: the pieces do not fill `out`"); goto done; }
    }
    {
        if (threads < 1) threads = 1;
        if (threads > 64) threads = 64;
        if ((size_t)out.len < ((size_t)1 << 20) || n < 2) threads = 1;
        if ((Py_ssize_t)threads > n) threads = (int)(n > 0 ? n : 1);
        /* pieces to threads by BYTES: thread t takes the pieces whose first byte lies in its share of the output */
        job_t jobs[64];
        pthread_t tid[64];
        Py_ssize_t p = 0;
        for (int t = 0; t < threads; ++t) {
            const size_t end = (size_t)out.len / (size_t)threads * (size_t)(t + 1);
            jobs[t].pieces = pieces;
            jobs[t].first = p;
            while (p < n && (t == threads - 1 || (size_t)(pieces[p].dst - (char*)out.buf) < end)) ++p;
            jobs[t].last = p;
        }
        int started = 0, failed = 0;
        Py_BEGIN_ALLOW_THREADS
        for (int t = 1; t < threads; ++t) {
            if (pthread_create(&tid[t], NULL, copy_job, &jobs[t]) != 0) { failed = t; break; }
            started = t;
        }
        copy_job(&jobs[0]);
        for (int t = failed ? failed : threads; failed && t < threads; ++t) copy_job(&jobs[t]);   /* (no thread to be had: this one copies) */
        for (int t = 1; t <= started; ++t) pthread_join(tid[t], NULL);
        Py_END_ALLOW_THREADS
    }
    result = Py_None;
    Py_INCREF(result);
done:
    for (Py_ssize_t i = 0; i < got; ++i) PyBuffer_Release(&views[i]);
    free(views);
    free(pieces);
    PyBuffer_Release(&out);
    Py_DECREF(seq);
    return result;
}

static PyMethodDef methods[] = {
    {"concat_rows", (PyCFunction)(void (*)(void))concat_rows, METH_VARARGS | METH_KEYWORDS,
     "concat_rows(arrays, out, threads=8): C-contiguous float64 buffers copied back to back into `out` by several threads"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostcopy", "threaded host copies of the batched drivers", -1, methods};

PyMODINIT_FUNC PyInit__hostcopy(void) { return PyModule_Create(&module); }
