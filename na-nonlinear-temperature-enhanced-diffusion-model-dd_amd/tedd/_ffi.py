"""ctypes binding of the C ABI declared in include/tedd.h (libtedd_hip.so).

The library is loaded from `<package root>/lib/libtedd_hip.so`.  There is no CPU fallback:
if the HIP library is missing or cannot be loaded, every entry into the device path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIBRARY = os.path.join(_PKG_ROOT, "lib", "libtedd_hip.so")

CONST_NAMES = ("R0", "Ea", "K1", "K2", "K3", "K4", "DT", "Dl_max", "phi_l", "gamma_T",
               "Kd", "Sd", "Dd_max", "phi_d", "phi_T", "r_sp", "T_ref")

MMS_STREAM, MMS_POL, MMS_EXPSIN, MMS_PEAKS, MMS_NFS, MMS_PROGRAM = -1, 0, 1, 2, 3, 4
REACT_CSTRIPLE, REACT_REGHCSTRIPLE, REACT_HCSTRIPLE = 0, 1, 2

_dptr = C.POINTER(C.c_double)
_i32ptr = C.POINTER(C.c_int32)


class TeddConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32),
        ("row_lo", C.c_int32), ("row_hi", C.c_int32),
        ("ghost", C.c_int32), ("model_kind", C.c_int32),
        ("reaction_kind", C.c_int32), ("mms_case", C.c_int32),
        ("device", C.c_int32), ("reserved", C.c_int32),
        ("eta", C.c_double),
        ("consts", C.c_double * 17),
        ("mms_params", C.c_double * 8),
    ]


class TeddStepStats(C.Structure):
    _fields_ = [
        ("sweeps", C.c_int32 * 3), ("cs_iterations", C.c_int32),
        ("last_update", C.c_double * 3), ("increment", C.c_double * 3),
        ("warm_started", C.c_int32 * 3), ("reserved", C.c_int32),
        ("first_update", C.c_double * 3), ("worst_ratio", C.c_double * 3),
    ]


# int (*tedd_reduce_fn)(void* user, double* dev_max, double* dev_min, int32_t n)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
# int (*tedd_ghost_fn)(void* user)
GHOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


# name -> (restype, argtypes); every symbol include/tedd.h declares
PROTOTYPES = {
    "tedd_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(TeddConfig), _dptr, _dptr]),
    "tedd_destroy": (None, [C.c_void_p]),
    "tedd_last_error": (C.c_char_p, [C.c_void_p]),
    "tedd_set_model": (C.c_int, [C.c_void_p, _dptr, C.c_int32, C.c_double, _dptr]),
    "tedd_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "tedd_set_mms_program": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "tedd_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tedd_sync": (C.c_int, [C.c_void_p]),
    "tedd_local_rows": (C.c_int32, [C.c_void_p]),
    "tedd_row_offset": (C.c_int32, [C.c_void_p]),
    "tedd_pitch": (C.c_int32, [C.c_void_p]),
    "tedd_placement": (C.c_char_p, [C.c_void_p]),
    "tedd_set_state": (C.c_int, [C.c_void_p] + [_dptr] * 5),
    "tedd_get_state": (C.c_int, [C.c_void_p] + [_dptr] * 5),
    "tedd_state_device_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "tedd_ghosts_refreshed": (C.c_int, [C.c_void_p]),
    "tedd_record_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "tedd_prefetch_forcing": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "tedd_set_state_exact": (C.c_int, [C.c_void_p, C.c_double]),
    "tedd_set_forcing": (C.c_int, [C.c_void_p, C.c_int32] + [_dptr] * 5),
    "tedd_step_pc": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                               C.c_double, _i32ptr, C.c_double, C.c_int32]),
    "tedd_step_feuler": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "tedd_advance": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_double, _i32ptr, C.c_double, C.c_int32, _dptr]),
    "tedd_predict_cp_cs": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _dptr, _dptr]),
    "tedd_correct_cp_cs": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _dptr, _dptr, _dptr, C.c_int32, C.c_double,
                                     _dptr, _dptr]),
    "tedd_newton_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, _dptr, _dptr, _dptr, C.c_int32,
                                   C.c_double, C.c_int32, _dptr, _dptr]),
    "tedd_get_residuals": (C.c_int, [C.c_void_p] + [_dptr] * 3),
    "tedd_eval_field": (C.c_int, [C.c_void_p, C.c_double] + [_dptr] * 5),
    "tedd_eval_exact": (C.c_int, [C.c_void_p, C.c_double] + [_dptr] * 5),
    "tedd_eval_mms": (C.c_int, [C.c_void_p, C.c_double, C.c_int32] + [_dptr] * 5),
    "tedd_eval_forcing": (C.c_int, [C.c_void_p, C.c_double] + [_dptr] * 5),
    "tedd_eval_jacobian": (C.c_int, [C.c_void_p, C.c_int32] + [_dptr] * 5),
    "tedd_error_norms": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(_dptr), _dptr]),
    "tedd_error_norms_noalloc": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "tedd_get_stats": (C.c_int, [C.c_void_p, C.POINTER(TeddStepStats)]),
    "tedd_check_converged": (C.c_int, [C.c_void_p]),
    "tedd_set_reduce_hook": (C.c_int, [C.c_void_p, REDUCE_FN, C.c_void_p]),
    "tedd_set_ghost_hook": (C.c_int, [C.c_void_p, GHOST_FN, C.c_void_p]),
    "tedd_ghosts_pending": (C.c_int, [C.c_void_p]),
    "tedd_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "tedd_profile_get": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), _dptr]),
}

_lock = threading.Lock()
_lib = None
_lib_path = DEFAULT_LIBRARY


def bind(lib):
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def use_library(path):
    """Load the C-ABI library from an explicit path (tests use this for the host twin)."""
    global _lib, _lib_path
    with _lock:
        _lib = bind(C.CDLL(path))
        _lib_path = path
    return _lib


def library():
    global _lib
    if _lib is None:
        if not os.path.exists(_lib_path):
            raise RuntimeError(
                f"tedd: the HIP library {_lib_path} is missing; build it with "
                "`python __graft_entry__.py build` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        use_library(_lib_path)
    return _lib


def library_path():
    return _lib_path


def _raise(rc, msg):
    msg = msg or f"tedd error {rc}"
    if rc == -1:
        raise ValueError(msg)
    if rc == -2:
        raise AssertionError(msg)
    raise RuntimeError(msg)


def _arr(a):
    return a.ctypes.data_as(_dptr) if a is not None else None


class Context:
    """One device context (one slab of the mesh).  Thin, allocation-explicit wrapper."""

    def __init__(self, x, y, consts, *, model_kind=2, reaction_kind=REACT_REGHCSTRIPLE, eta=0.0,
                 mms_case=MMS_STREAM, mms_params=(), row_lo=0, row_hi=None, ghost=0, device=-1):
        self._lib = library()
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.N, self.M = len(x) - 1, len(y) - 1
        cfg = TeddConfig()
        cfg.N, cfg.M = self.N, self.M
        cfg.row_lo = row_lo
        cfg.row_hi = self.N + 1 if row_hi is None else row_hi
        cfg.ghost = ghost
        cfg.model_kind = model_kind
        cfg.reaction_kind = reaction_kind
        cfg.mms_case = mms_case
        cfg.device = device
        cfg.eta = float(eta)
        for k, v in enumerate(consts):
            cfg.consts[k] = float(v)
        params = list(mms_params) + [0.0] * (8 - len(mms_params))
        for k in range(8):
            cfg.mms_params[k] = float(params[k])
        h = C.c_void_p()
        rc = self._lib.tedd_create(C.byref(h), C.byref(cfg), _arr(x), _arr(y))
        if rc != 0:
            _raise(rc, (self._lib.tedd_last_error(None) or b"").decode())
        self._h = h
        self.mms_case = mms_case
        self.reaction_kind = reaction_kind
        self.nrows = int(self._lib.tedd_local_rows(h))
        self.row_offset = int(self._lib.tedd_row_offset(h))
        self.pitch = int(self._lib.tedd_pitch(h))
        self.placement = self._lib.tedd_placement(h).decode()
        self.shape = (self.nrows, self.M + 1)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.tedd_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            err, self._hook_error = getattr(self, "_hook_error", None), None
            if err is not None:
                raise err
            _raise(rc, (self._lib.tedd_last_error(self._h) or b"").decode())
        return rc

    def _in(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != self.shape:
            raise AssertionError(f"array shape {a.shape} != local mesh shape {self.shape}")
        return a

    def _out(self, n):
        return [np.empty(self.shape, dtype=np.float64) for _ in range(n)]

    # -- configuration
    def set_model(self, consts, model_kind, eta, mms_params=()):
        c = (C.c_double * 17)(*[float(v) for v in consts])
        params = list(mms_params) + [0.0] * (8 - len(mms_params))
        p = (C.c_double * 8)(*[float(v) for v in params])
        self._check(self._lib.tedd_set_model(self._h, c, model_kind, float(eta), p))

    def prefetch_forcing(self, t, keep_t):
        self._check(self._lib.tedd_prefetch_forcing(self._h, float(t), float(keep_t)))

    def set_mms_program(self, which, code, consts, n_regs):
        """code: uint32 words, consts: float64 (tedd_set_mms_program; which 0 node program, 1 cell program)."""
        code = np.ascontiguousarray(code, dtype=np.uint32)
        consts = np.ascontiguousarray(consts, dtype=np.float64)
        self._check(self._lib.tedd_set_mms_program(
            self._h, int(which), code.ctypes.data, int(code.size),
            consts.ctypes.data if consts.size else None, int(consts.size), int(n_regs)))

    def set_option(self, name, value):
        self._check(self._lib.tedd_set_option(self._h, name.encode(), float(value)))

    def set_stream(self, stream_handle):
        self._check(self._lib.tedd_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        self._check(self._lib.tedd_sync(self._h))

    # -- state
    def set_state(self, cp, T, cl, cd, cs):
        arrs = [self._in(a) for a in (cp, T, cl, cd, cs)]
        self._check(self._lib.tedd_set_state(self._h, *[_arr(a) for a in arrs]))

    def get_state(self):
        out = self._out(5)
        self._check(self._lib.tedd_get_state(self._h, *[_arr(a) for a in out]))
        return out

    def set_state_exact(self, t):
        self._check(self._lib.tedd_set_state_exact(self._h, float(t)))

    def state_device_ptrs(self):
        p = (C.c_void_p * 5)()
        self._check(self._lib.tedd_state_device_ptrs(self._h, p))
        return [int(v) for v in p]

    def record_device_ptr(self):
        """(address, n): the convergence record of the step in flight in the context's memory (tedd_record_device_ptr)."""
        p, n = C.c_void_p(), C.c_int32()
        self._check(self._lib.tedd_record_device_ptr(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def ghosts_refreshed(self):
        self._check(self._lib.tedd_ghosts_refreshed(self._h))

    def set_forcing(self, which, fcp, fT, fcl, fcd, fcs):
        arrs = [self._in(a) for a in (fcp, fT, fcl, fcd, fcs)]
        self._check(self._lib.tedd_set_forcing(self._h, which, *[_arr(a) for a in arrs]))

    # -- stepping
    @staticmethod
    def _sweeps(sweeps):
        if sweeps is None:
            return None
        return (C.c_int32 * 3)(*[int(s) for s in sweeps])

    def step_pc(self, t0, dt, *, num_pc_steps=1, num_newton_steps=1, cs_newton_iterations=5,
                cs_consec_xs_rtol=1e-6, jacobi_sweeps=None, jacobi_tol=1e-13, jacobi_max=100000):
        self._check(self._lib.tedd_step_pc(self._h, float(t0), float(dt), num_pc_steps, num_newton_steps,
                                           cs_newton_iterations, float(cs_consec_xs_rtol),
                                           self._sweeps(jacobi_sweeps), float(jacobi_tol), jacobi_max))

    def step_feuler(self, t0, dt):
        self._check(self._lib.tedd_step_feuler(self._h, float(t0), float(dt)))

    def advance(self, t0, dt, nsteps, *, integrator=1, num_pc_steps=1, num_newton_steps=1,
                cs_newton_iterations=5, cs_consec_xs_rtol=1e-6, jacobi_sweeps=None, jacobi_tol=1e-13,
                jacobi_max=100000, want_errors=True):
        series = np.empty((nsteps + 1, 8), dtype=np.float64) if want_errors else None
        self._check(self._lib.tedd_advance(self._h, float(t0), float(dt), nsteps, integrator, num_pc_steps,
                                           num_newton_steps, cs_newton_iterations, float(cs_consec_xs_rtol),
                                           self._sweeps(jacobi_sweeps), float(jacobi_tol), jacobi_max,
                                           _arr(series)))
        return series

    # -- the pieces of a predictor-corrector step (the set state is at_t / at_t0)
    def predict_cp_cs(self, t, dt):
        cp1, cs1 = self._out(2)
        self._check(self._lib.tedd_predict_cp_cs(self._h, float(t), float(dt), _arr(cp1), _arr(cs1)))
        return cp1, cs1

    def correct_cp_cs(self, t0, dt, T1, cl1, cd1, *, cs_newton_iterations=5, cs_consec_xs_rtol=1e-6):
        ins = [self._in(a) for a in (T1, cl1, cd1)]
        cp1, cs1 = self._out(2)
        self._check(self._lib.tedd_correct_cp_cs(self._h, float(t0), float(dt), *[_arr(a) for a in ins],
                                                 int(cs_newton_iterations), float(cs_consec_xs_rtol),
                                                 _arr(cp1), _arr(cs1)))
        return cp1, cs1

    def newton_step(self, var, t0, dt, Y, T1=None, cl1=None, *, jacobi_sweeps=0, jacobi_tol=1e-13,
                    jacobi_max=100000):
        """var: 1 T, 2 cl, 3 cd.  Returns (updated unknown, residual)."""
        ins = [self._in(a) if a is not None else None for a in (Y, T1, cl1)]
        out, res = self._out(2)
        self._check(self._lib.tedd_newton_step(self._h, int(var), float(t0), float(dt), *[_arr(a) for a in ins],
                                               int(jacobi_sweeps), float(jacobi_tol), int(jacobi_max),
                                               _arr(out), _arr(res)))
        return out, res

    def get_residuals(self):
        out = self._out(3)
        self._check(self._lib.tedd_get_residuals(self._h, *[_arr(a) for a in out]))
        return out

    def get_stats(self):
        s = TeddStepStats()
        self._check(self._lib.tedd_get_stats(self._h, C.byref(s)))
        return {"sweeps": list(s.sweeps), "cs_iterations": int(s.cs_iterations),
                "last_update": list(s.last_update), "increment": list(s.increment),
                "warm_started": [bool(w) for w in s.warm_started], "first_update": list(s.first_update),
                "worst_ratio": list(s.worst_ratio)}

    def check_converged(self):
        """RuntimeError if a Jacobi solve with a library-chosen sweep count missed its tolerance by more than a
        factor 10 in some step since the last check (tedd_check_converged)."""
        self._check(self._lib.tedd_check_converged(self._h))

    def set_reduce_hook(self, fn):
        """fn(dev_max_address, dev_min_address, n) -> None, or None to remove the hook (tedd_set_reduce_hook).
        An exception raised by fn fails the step and is re-raised by the call that was running."""
        if fn is None:
            self._reduce_cb = None
            self._check(self._lib.tedd_set_reduce_hook(self._h, C.cast(None, REDUCE_FN), None))
            return

        def trampoline(_user, pmax, pmin, n):
            try:
                fn(int(pmax), int(pmin), int(n))
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not propagate through the C frames
                self._hook_error = e
                return 1

        self._reduce_cb = REDUCE_FN(trampoline)   # keep it alive as long as the library may call it
        self._check(self._lib.tedd_set_reduce_hook(self._h, self._reduce_cb, None))

    def set_ghost_hook(self, fn):
        """fn() -> None completes the transfer of the ghost rows that ghosts_pending() declared in flight, or None to
        remove the hook (tedd_set_ghost_hook).  An exception raised by fn fails the step and is re-raised."""
        if fn is None:
            self._ghost_cb = None
            self._check(self._lib.tedd_set_ghost_hook(self._h, C.cast(None, GHOST_FN), None))
            return

        def trampoline(_user):
            try:
                fn()
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not propagate through the C frames
                self._hook_error = e
                return 1

        self._ghost_cb = GHOST_FN(trampoline)
        self._check(self._lib.tedd_set_ghost_hook(self._h, self._ghost_cb, None))

    def ghosts_pending(self):
        """The driver posted the transfer of the current state's ghost rows and did not wait (tedd_ghosts_pending)."""
        self._check(self._lib.tedd_ghosts_pending(self._h))

    # -- parity hooks
    def eval_field(self, t):
        out = self._out(5)
        self._check(self._lib.tedd_eval_field(self._h, float(t), *[_arr(a) for a in out]))
        return out

    def eval_exact(self, t):
        out = self._out(5)
        self._check(self._lib.tedd_eval_exact(self._h, float(t), *[_arr(a) for a in out]))
        return out

    def eval_mms(self, t, what):
        out = self._out(5)
        self._check(self._lib.tedd_eval_mms(self._h, float(t), int(what), *[_arr(a) for a in out]))
        return out

    def eval_forcing(self, t):
        out = self._out(5)
        self._check(self._lib.tedd_eval_forcing(self._h, float(t), *[_arr(a) for a in out]))
        return out

    def eval_jacobian(self, op):
        out = self._out(5)
        self._check(self._lib.tedd_eval_jacobian(self._h, int(op), *[_arr(a) for a in out]))
        return out

    def error_norms(self, t, exact=None):
        out = np.empty(8, dtype=np.float64)
        if exact is not None:
            arrs = [self._in(a) for a in exact]
            ptrs = (_dptr * 5)(*[_arr(a) for a in arrs])
            self._check(self._lib.tedd_error_norms(self._h, float(t), ptrs, _arr(out)))
        else:
            self._check(self._lib.tedd_error_norms(self._h, float(t), None, _arr(out)))
        return out

    def error_norms_noalloc(self, t, out_device_ptr):
        """The 8 error sums into the caller's device memory (an address), without allocating or synchronising."""
        self._check(self._lib.tedd_error_norms_noalloc(self._h, float(t), C.c_void_p(int(out_device_ptr))))

    # -- timing record
    def profile_enable(self, on=True):
        self._check(self._lib.tedd_profile_enable(self._h, 1 if on else 0))

    def profile_get(self):
        cap = 32
        names = (C.c_char_p * cap)()
        n = (C.c_int64 * cap)()
        ms = (C.c_double * cap)()
        k = self._check(self._lib.tedd_profile_get(self._h, cap, names, n, ms))
        return {names[i].decode(): {"launches": int(n[i]), "total_ms": float(ms[i])} for i in range(k)}
