/* A CPython extension that makes the per-step call of the single-state protocol (PPO.select_action as
 * train.py:176 calls it, 10^5 times per second) without ctypes: argument marshalling through ctypes costs
 * more than half a microsecond per call, next to the 3.5 us the device needs for the step itself.
 *
 * The module knows nothing about HIP and links against nothing but Python: it is handed the ADDRESS of
 * ppo_act_session_step_bound (include/ppo_hip.h), the session, and the page-locked state / action buffers
 * that were bound to the session, and per step it
 *   1. takes the observation through the buffer protocol (a contiguous vector of float64 - what gym hands
 *      over - or float32), converts it into the page-locked state buffer,
 *   2. calls ppo_act_session_step_bound(session, row, rng_offset),
 *   3. returns the discrete action as a Python int (None for a continuous policy: the caller copies the
 *      action vector out of its page-locked buffer).
 * An observation it cannot take (other type, shape, layout) makes it return NotImplemented; the caller
 * then goes through the ctypes binding for that step.  An error code of the library is raised as
 * RuntimeError(code).  The GIL is held during the call: the call is a few microseconds of spinning on host
 * memory by design.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef int (*step_bound_fn)(void* session, int64_t row, uint64_t rng_offset);

typedef struct {
    PyObject_HEAD
    step_bound_fn fn;
    void* session;
    float* host_state;        /* page-locked, state_dim floats */
    const int32_t* action;    /* page-locked, word 0 = the discrete action */
    Py_ssize_t state_dim;
    int continuous;
} Stepper;

static PyObject* Stepper_step(PyObject* self_, PyObject* const* args, Py_ssize_t nargs) {
    Stepper* self = (Stepper*)self_;
    if (nargs != 3) {
        PyErr_SetString(PyExc_TypeError, "step(state, row, rng_offset)");
        return NULL;
    }
    Py_buffer view;
    if (PyObject_GetBuffer(args[0], &view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) < 0) {
        PyErr_Clear();
        Py_RETURN_NOTIMPLEMENTED;
    }
    const char* f = view.format ? view.format : "B";
    if (*f == '@' || *f == '=' || *f == '<') ++f;
    int ok = view.ndim == 1 && view.shape[0] == self->state_dim && f[1] == '\0';
    if (ok && f[0] == 'd' && view.itemsize == 8) {
        const double* src = (const double*)view.buf;
        for (Py_ssize_t i = 0; i < self->state_dim; ++i) self->host_state[i] = (float)src[i];
    } else if (ok && f[0] == 'f' && view.itemsize == 4) {
        memcpy(self->host_state, view.buf, sizeof(float) * (size_t)self->state_dim);
    } else {
        ok = 0;
    }
    PyBuffer_Release(&view);
    if (!ok) Py_RETURN_NOTIMPLEMENTED;
    const long long row = PyLong_AsLongLong(args[1]);
    if (row == -1 && PyErr_Occurred()) return NULL;
    const unsigned long long off = PyLong_AsUnsignedLongLong(args[2]);
    if (off == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
    const int rc = self->fn(self->session, (int64_t)row, (uint64_t)off);
    if (rc != 0) {
        PyObject* code = PyLong_FromLong(rc);
        if (code) {
            PyErr_SetObject(PyExc_RuntimeError, code);
            Py_DECREF(code);
        }
        return NULL;
    }
    if (self->continuous) Py_RETURN_NONE;
    return PyLong_FromLong((long)self->action[0]);
}

static int Stepper_init(PyObject* self_, PyObject* args, PyObject* kwds) {
    Stepper* self = (Stepper*)self_;
    unsigned long long fn, session, host_state, action;
    Py_ssize_t state_dim;
    int continuous;
    (void)kwds;
    if (!PyArg_ParseTuple(args, "KKKKnp", &fn, &session, &host_state, &action, &state_dim, &continuous)) return -1;
    if (!fn || !session || !host_state || !action || state_dim < 1) {
        PyErr_SetString(PyExc_ValueError, "Stepper(fn_address, session_address, host_state_address, "
                                          "action_address, state_dim, continuous): null address");
        return -1;
    }
    self->fn = (step_bound_fn)(uintptr_t)fn;
    self->session = (void*)(uintptr_t)session;
    self->host_state = (float*)(uintptr_t)host_state;
    self->action = (const int32_t*)(uintptr_t)action;
    self->state_dim = state_dim;
    self->continuous = continuous;
    return 0;
}

static PyMethodDef Stepper_methods[] = {
    {"step", (PyCFunction)(void (*)(void))Stepper_step, METH_FASTCALL,
     "step(state, row, rng_offset) -> int | None | NotImplemented"},
    {NULL, NULL, 0, NULL}};

static PyTypeObject StepperType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "ppo_amd._fastcall.Stepper",
    .tp_basicsize = sizeof(Stepper),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "the per-step call of a single-state act session, bound to one session",
    .tp_methods = Stepper_methods,
    .tp_init = Stepper_init,
    .tp_new = PyType_GenericNew,
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastcall",
                                       "ctypes-free per-step call of the single-state protocol", -1,
                                       NULL, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fastcall(void) {
    if (PyType_Ready(&StepperType) < 0) return NULL;
    PyObject* m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    Py_INCREF(&StepperType);
    if (PyModule_AddObject(m, "Stepper", (PyObject*)&StepperType) < 0) {
        Py_DECREF(&StepperType);
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
