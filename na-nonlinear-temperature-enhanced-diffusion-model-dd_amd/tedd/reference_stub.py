"""The binding a maintainer of the reference adds to run its predictor-corrector integrator
(P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple, src/prob1base.py:3596-3702) through the C ABI
of include/tedd.h.  ctypes and NumPy only: nothing of this package is imported, and every object the class
touches (grid, model, field, StateVars) is the reference's own.

    import reference_stub
    Integrator = reference_stub.make_integrator_class(".../lib/libtedd_hip.so")
    trial = mms_trial_utils.MMSTrial(..., integrator_cls=Integrator,
                                     integrator_params={"regularization_factor": 50.0})

The class is handed to the reference's MMSTrial (src/mms_trial_utils.py:247-249) in place of its integrator
and needs no other change on the reference's side.
"""
import ctypes as C
from collections.abc import Mapping

import numpy as np

CONST_FIELDS = ("R0", "Ea", "K1", "K2", "K3", "K4", "DT", "Dl_max", "phi_l", "gamma_T", "Kd", "Sd", "Dd_max",
                "phi_d", "phi_T", "r_sp", "T_ref")   # ModelConsts._fields, src/prob1base.py:28-45
MMS_STREAM, MMS_POL, MMS_EXPSIN = -1, 0, 1     # include/tedd.h: tedd_mms_case
REACT_REGHCSTRIPLE = 1
ERR_INVALID, ERR_ASSERT, ERR_RUNTIME, ERR_NOT_CONVERGED = -1, -2, -3, -4   # include/tedd.h: tedd_status
dp = C.POINTER(C.c_double)


class TeddConfig(C.Structure):            # include/tedd.h: tedd_config
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("row_lo", C.c_int32), ("row_hi", C.c_int32),
                ("ghost", C.c_int32), ("model_kind", C.c_int32), ("reaction_kind", C.c_int32),
                ("mms_case", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("eta", C.c_double), ("consts", C.c_double * 17), ("mms_params", C.c_double * 8)]


class TeddStats(C.Structure):             # include/tedd.h: tedd_step_stats
    _fields_ = [("sweeps", C.c_int32 * 3), ("cs_iterations", C.c_int32), ("last_update", C.c_double * 3),
                ("increment", C.c_double * 3), ("warm_started", C.c_int32 * 3), ("reserved", C.c_int32),
                ("first_update", C.c_double * 3), ("worst_ratio", C.c_double * 3)]


def load(path):
    lib = C.CDLL(path)
    lib.tedd_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(TeddConfig), dp, dp]
    lib.tedd_destroy.argtypes = [C.c_void_p]
    lib.tedd_destroy.restype = None
    lib.tedd_last_error.argtypes = [C.c_void_p]
    lib.tedd_last_error.restype = C.c_char_p
    lib.tedd_set_model.argtypes = [C.c_void_p, dp, C.c_int32, C.c_double, dp]
    lib.tedd_set_state.argtypes = [C.c_void_p] + [dp] * 5
    lib.tedd_get_state.argtypes = [C.c_void_p] + [dp] * 5
    lib.tedd_set_forcing.argtypes = [C.c_void_p, C.c_int32] + [dp] * 5
    lib.tedd_step_pc.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_double, C.POINTER(C.c_int32), C.c_double, C.c_int32]
    lib.tedd_check_converged.argtypes = [C.c_void_p]
    lib.tedd_get_stats.argtypes = [C.c_void_p, C.POINTER(TeddStats)]
    lib.tedd_get_residuals.argtypes = [C.c_void_p] + [dp] * 3
    return lib


def _ptr(a):
    return a.ctypes.data_as(dp)


class LastResidual(Mapping):
    """integrator.last_residual (src/prob1base.py:2942-2943, 3043, 3078, 3113): keys "T", "cl", "cd", full-size
    arrays 2 v1 - dt F_v(S', t1) - Y_v0 of the last step.  The library evaluates them when first asked
    (tedd_get_residuals), which must be before the next step() of the same integrator."""

    def __init__(self, fetch):
        self._fetch, self._got = fetch, None

    def __getitem__(self, key):
        if self._got is None:
            self._got = self._fetch()
        return self._got[key]

    def __iter__(self):
        return iter(("T", "cl", "cd") if self._fetch else ())

    def __len__(self):
        return 3 if self._fetch else 0


def make_integrator_class(library_path, *, mms_case=MMS_STREAM, device=-1):
    """-> a class with the constructor, attributes, step() and last_residual of the reference's RegHCsTriple
    predictor-corrector.  mms_case: MMS_STREAM hands the library the forcing arrays of the reference's own field
    at t0 and t0+dt before every step (any MMS case); MMS_POL / MMS_EXPSIN let the library evaluate the forcing
    of that case itself (the field must have been built on that case)."""
    lib = load(library_path)

    class HipRegHCsTripleIntegrator:
        jacobi_tol, jacobi_max = 1e-13, 100000     # of the Jacobi sweeps that stand in for spsolve (:2103, :2130)

        def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1, regularization_factor,
                     num_newton_iterations=5, consec_xs_rtol=1e-6):
            # attribute names as in the reference (:2934-2943, :3627-3629): its tests read and write them
            f = self.semi_discrete_field = semi_discrete_field
            self._model, self._grid = f.model, f.grid
            self.num_pc_steps, self.num_newton_steps = num_pc_steps, num_newton_steps
            self._regularization_factor = regularization_factor
            self._num_newton_iterations, self._consec_xs_rtol = num_newton_iterations, consec_xs_rtol
            self.last_residual = LastResidual(None)
            self.streamed = mms_case == MMS_STREAM
            g = self._grid
            cfg = TeddConfig(N=g.N, M=g.M, row_lo=0, row_hi=g.N + 1, ghost=0, reaction_kind=REACT_REGHCSTRIPLE,
                             mms_case=mms_case, device=device)
            self._pushed = self._model_now()
            cfg.model_kind, cfg.eta = self._pushed[1], self._pushed[2]
            cfg.consts[:] = self._pushed[0]
            self.x = np.ascontiguousarray(g.x, dtype=np.float64)
            self.y = np.ascontiguousarray(g.y, dtype=np.float64)
            self.h = C.c_void_p()
            if lib.tedd_create(C.byref(self.h), C.byref(cfg), _ptr(self.x), _ptr(self.y)) != 0:
                raise RuntimeError(lib.tedd_last_error(None).decode())

        def __del__(self):
            if getattr(self, "h", None):
                lib.tedd_destroy(self.h)
                self.h = None

        def _check(self, rc):
            if rc != 0:
                msg = lib.tedd_last_error(self.h).decode()
                raise {ERR_INVALID: ValueError, ERR_ASSERT: AssertionError}.get(rc, RuntimeError)(msg)

        def _model_now(self):
            m = self._model
            return (tuple(float(getattr(m, k)) for k in CONST_FIELDS),
                    2 if type(m).__name__ == "DefaultModel02" else 1, float(self._regularization_factor))

        def _refresh_model(self):
            """The reference reads self._model and self._regularization_factor anew in every step and its tests
            change them in place (tests/test_time_integrator_hcs_triple.py:377-387): so does every step here."""
            now = self._model_now()
            if now != self._pushed:
                if not self.streamed:   # the library's own forcing belongs to a case BUILT on the constants it has
                    raise NotImplementedError("model changed in place: only with mms_case=MMS_STREAM")
                consts = np.array(now[0])
                self._check(lib.tedd_set_model(self.h, _ptr(consts), now[1], now[2], _ptr(np.zeros(8))))
                self._pushed = now

        def _upload(self, ins, t0, dt):
            self._check(lib.tedd_set_state(self.h, *[_ptr(a) for a in ins]))
            if self.streamed:
                f, g = self.semi_discrete_field, self.semi_discrete_field.grid
                for which, t in ((0, t0), (1, t0 + dt)):
                    arrays = [np.ascontiguousarray(fn(t, g.xx, g.yy), dtype=np.float64)
                              for fn in (f.fcp, f.fT, f.fcl, f.fcd, f.fcs)]
                    self._check(lib.tedd_set_forcing(self.h, which, *[_ptr(a) for a in arrays]))

        def _residuals(self, shape):
            r = [np.empty(shape) for _ in range(3)]
            self._check(lib.tedd_get_residuals(self.h, *[_ptr(a) for a in r]))
            return dict(zip(("T", "cl", "cd"), r))

        def step(self, at_t0, *, t0, dt):
            assert dt > 0
            self._refresh_model()
            ins = [np.ascontiguousarray(getattr(at_t0, v), dtype=np.float64) for v in ("cp", "T", "cl", "cd", "cs")]
            tol, sweeps, st = float(self.jacobi_tol), None, TeddStats()
            for _attempt in range(12):
                self._upload(ins, t0, dt)
                self._check(lib.tedd_step_pc(self.h, t0, dt, int(self.num_pc_steps), int(self.num_newton_steps),
                                             int(self._num_newton_iterations), float(self._consec_xs_rtol),
                                             sweeps, tol, int(self.jacobi_max)))
                # spsolve cannot stop short, sweeps can: counts the library chose are judged by the library,
                # counts fixed here (the redo below) from the record of the step
                rc = lib.tedd_check_converged(self.h)
                self._check(lib.tedd_get_stats(self.h, C.byref(st)))
                if rc not in (0, ERR_NOT_CONVERGED):
                    self._check(rc)
                judged = self.num_pc_steps > 0 and self.num_newton_steps > 0
                if rc == 0 and (sweeps is None or not judged or all(
                        u == 0.0 or u <= 10 * tol * i for u, i in zip(st.last_update, st.increment))):
                    break
                if max(st.sweeps) >= self.jacobi_max:
                    raise RuntimeError(lib.tedd_last_error(self.h).decode() or "Jacobi sweeps did not reach jacobi_tol")
                sweeps = (C.c_int32 * 3)(*[min(int(self.jacobi_max), 2 * n) for n in st.sweeps])  # same input, twice the sweeps
            else:
                raise RuntimeError("Jacobi sweeps did not reach jacobi_tol")
            outs = [np.empty_like(a) for a in ins]
            self._check(lib.tedd_get_state(self.h, *[_ptr(a) for a in outs]))
            if self.num_pc_steps > 0 and self.num_newton_steps > 0:
                self.last_residual = LastResidual(lambda shape=ins[0].shape: self._residuals(shape))
            return at_t0.with_changes(cp=outs[0], T=outs[1], cl=outs[2], cd=outs[3], cs=outs[4])

    return HipRegHCsTripleIntegrator
