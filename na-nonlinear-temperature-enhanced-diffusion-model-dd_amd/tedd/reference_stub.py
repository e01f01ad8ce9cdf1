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

import numpy as np

CONST_FIELDS = ("R0", "Ea", "K1", "K2", "K3", "K4", "DT", "Dl_max", "phi_l", "gamma_T", "Kd", "Sd", "Dd_max",
                "phi_d", "phi_T", "r_sp", "T_ref")   # ModelConsts._fields, src/prob1base.py:28-45
MMS_STREAM, MMS_POL, MMS_EXPSIN = -1, 0, 1     # include/tedd.h: tedd_mms_case
REACT_REGHCSTRIPLE = 1
dp = C.POINTER(C.c_double)


class TeddConfig(C.Structure):            # include/tedd.h: tedd_config
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("row_lo", C.c_int32), ("row_hi", C.c_int32),
                ("ghost", C.c_int32), ("model_kind", C.c_int32), ("reaction_kind", C.c_int32),
                ("mms_case", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("eta", C.c_double), ("consts", C.c_double * 17), ("mms_params", C.c_double * 8)]


def load(path):
    lib = C.CDLL(path)
    lib.tedd_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(TeddConfig), dp, dp]
    lib.tedd_destroy.argtypes = [C.c_void_p]
    lib.tedd_destroy.restype = None
    lib.tedd_last_error.argtypes = [C.c_void_p]
    lib.tedd_last_error.restype = C.c_char_p
    lib.tedd_set_state.argtypes = [C.c_void_p] + [dp] * 5
    lib.tedd_get_state.argtypes = [C.c_void_p] + [dp] * 5
    lib.tedd_set_forcing.argtypes = [C.c_void_p, C.c_int32] + [dp] * 5
    lib.tedd_step_pc.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_double, C.POINTER(C.c_int32), C.c_double, C.c_int32]
    return lib


def _ptr(a):
    return a.ctypes.data_as(dp)


def make_integrator_class(library_path, *, mms_case=MMS_STREAM, device=-1):
    """-> a class with the constructor and step() of the reference's RegHCsTriple predictor-corrector.
    mms_case: MMS_STREAM hands the library the forcing arrays of the reference's own field at t0 and t0+dt
    before every step (any MMS case); MMS_POL / MMS_EXPSIN let the library evaluate the forcing of that
    case itself (the field must have been built on that case)."""
    lib = load(library_path)

    class HipRegHCsTripleIntegrator:
        def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1, regularization_factor,
                     num_newton_iterations=5, consec_xs_rtol=1e-6):
            f = self.semi_discrete_field = semi_discrete_field
            g, m = f.grid, f.model
            self.streamed = mms_case == MMS_STREAM
            cfg = TeddConfig(N=g.N, M=g.M, row_lo=0, row_hi=g.N + 1, ghost=0,
                             model_kind=2 if type(m).__name__ == "DefaultModel02" else 1,
                             reaction_kind=REACT_REGHCSTRIPLE,
                             mms_case=mms_case,
                             device=device, eta=float(regularization_factor))
            cfg.consts[:] = [float(getattr(m, k)) for k in CONST_FIELDS]
            self.x = np.ascontiguousarray(g.x, dtype=np.float64)
            self.y = np.ascontiguousarray(g.y, dtype=np.float64)
            self.h = C.c_void_p()
            if lib.tedd_create(C.byref(self.h), C.byref(cfg), _ptr(self.x), _ptr(self.y)) != 0:
                raise RuntimeError(lib.tedd_last_error(None).decode())
            self.args = (int(num_pc_steps), int(num_newton_steps), int(num_newton_iterations), float(consec_xs_rtol))

        def __del__(self):
            if getattr(self, "h", None):
                lib.tedd_destroy(self.h)
                self.h = None

        def _check(self, rc):
            if rc != 0:
                msg = lib.tedd_last_error(self.h).decode()
                raise (ValueError if rc == -1 else RuntimeError)(msg)

        def _forcing(self, which, t):
            f, g = self.semi_discrete_field, self.semi_discrete_field.grid
            arrays = [np.ascontiguousarray(fn(t, g.xx, g.yy), dtype=np.float64)
                      for fn in (f.fcp, f.fT, f.fcl, f.fcd, f.fcs)]
            self._check(lib.tedd_set_forcing(self.h, which, *[_ptr(a) for a in arrays]))

        def step(self, at_t0, *, t0, dt):
            assert dt > 0
            ins = [np.ascontiguousarray(getattr(at_t0, v), dtype=np.float64) for v in ("cp", "T", "cl", "cd", "cs")]
            self._check(lib.tedd_set_state(self.h, *[_ptr(a) for a in ins]))
            if self.streamed:
                self._forcing(0, t0)
                self._forcing(1, t0 + dt)
            self._check(lib.tedd_step_pc(self.h, t0, dt, *self.args, None, 1e-13, 100000))
            outs = [np.empty_like(a) for a in ins]
            self._check(lib.tedd_get_state(self.h, *[_ptr(a) for a in outs]))
            return at_t0.with_changes(cp=outs[0], T=outs[1], cl=outs[2], cd=outs[3], cs=outs[4])

    return HipRegHCsTripleIntegrator
