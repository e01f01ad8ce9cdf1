"""The MMS case library of the reference (src/prob1_mms_cases.py): same class names and constructor
arguments.

    import tedd.prob1_mms_cases as p1mc
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)

The classes derived from MMSCaseDevice only record which closed form of the device library to use and
its parameters; the values (exact solution, its derivatives, and the forcing terms derived from them)
are computed on the GPU.  The classes derived from MMSCaseSymbolic (StiffExpDecay,
PolWithOscilatingTime, CsZeroCrossing) state their solution as SymPy expressions, which are lowered to
register programs that the device library runs at every node (tedd/_symprog.py).
"""
from __future__ import annotations

from typing import List

import numpy as np

try:
    import sympy
except ImportError:  # only the symbolic cases need it
    sympy = None

from . import _ffi
from . import prob1base as p1


def _unit_square_bubble():
    return p1.x_sym * (1 - p1.x_sym) * p1.y_sym * (1 - p1.y_sym)


class MMSCaseStiffExpDecay(p1.MMSCaseSymbolic):
    """u_v = x(1-x) y(1-y) e^{-a_v t} with decay rates a_cl = a_base, a_T = a_base/10,
    a_cd = a_cs = a_base/100, a_cp = a_base/1000: one spatial profile, time scales a decade apart
    (reference src/prob1_mms_cases.py:12-63)."""

    def __init__(self, grid: p1.Grid, model: p1.DefaultModel01, *, a_base: float = 1.0):
        W, t = _unit_square_bubble(), p1.t_sym
        rate = {"cl": a_base, "T": a_base / 10.0, "cd": a_base / 100.0, "cs": a_base / 100.0,
                "cp": a_base / 1000.0}
        sol = {v: W * sympy.exp(-a * t) for v, a in rate.items()}
        super().__init__(grid=grid, model=model, cp_sym_expr=sol["cp"], T_sym_expr=sol["T"],
                         cl_sym_expr=sol["cl"], cd_sym_expr=sol["cd"], cs_sym_expr=sol["cs"])


def make_MMSCaseStiffExpDecay_cls(ref_speed):
    class the_MMSCaseStiffExpDecay(MMSCaseStiffExpDecay):
        def __init__(self, grid: p1.Grid, model: p1.DefaultModel01):
            super().__init__(grid=grid, model=model, a_base=ref_speed)

    return the_MMSCaseStiffExpDecay


class MMSCasePolWithOscilatingTime(p1.MMSCaseSymbolic):
    """All five unknowns equal ampl (1 + shrink sin(speed t)) x(1-x) y(1-y); non-negative for
    ampl > 0 and shrink in (0, 1] (reference src/prob1_mms_cases.py:76-134)."""

    def __init__(self, grid: p1.Grid, model: p1.DefaultModel01, *, ampl: float = 1, speed: float = 1,
                 shrink: float = 1):
        u = ampl * (1 + shrink * sympy.sin(speed * p1.t_sym)) * _unit_square_bubble()
        super().__init__(grid=grid, model=model, cp_sym_expr=u, T_sym_expr=u, cl_sym_expr=u, cd_sym_expr=u,
                         cs_sym_expr=u)


def make_MMSCasePolWithOscilatingTime_cls(*, ampl, speed):
    class the_MMSCasePolWithOscilatingTime(MMSCasePolWithOscilatingTime):
        def __init__(self, grid: p1.Grid, model: p1.DefaultModel01):
            super().__init__(grid=grid, model=model, ampl=ampl, speed=speed)

    return the_MMSCasePolWithOscilatingTime


class MMSCaseCsZeroCrossing(p1.MMSCaseSymbolic):
    """cp = T = cl = cd = 0 and cs = (cs_A - cs_B t) W(x, y), which changes sign at t = cs_A / cs_B;
    W defaults to x(1-x) y(1-y) (reference src/prob1_mms_cases.py:341-392)."""

    def __init__(self, grid: p1.Grid, model: p1.DefaultModel01, *, cs_A: float = 0.5, cs_B: float = 1.0,
                 spatial_profile_expr=None):
        W = _unit_square_bubble() if spatial_profile_expr is None else spatial_profile_expr
        zero = sympy.S(0)
        cs = (sympy.Float(cs_A) - sympy.Float(cs_B) * p1.t_sym) * W
        super().__init__(grid=grid, model=model, cp_sym_expr=zero, T_sym_expr=zero, cl_sym_expr=zero,
                         cd_sym_expr=zero, cs_sym_expr=cs)


class MMSCasePol(p1.MMSCaseDevice):
    """All five unknowns equal x(1-x) y(1-y) / (1+t)   (reference src/prob1_mms_cases.py:250-277)."""

    mms_case_id = _ffi.MMS_POL

    def __init__(self, grid: p1.Grid, *, model: p1.DefaultModel01):
        super().__init__(grid, model)


class MMSCaseExpSin(p1.MMSCaseDevice):
    """W = sin(pi x) sin(pi y); T = e^{-2 pi^2 DT t} W, cl = -e^{-t} W, cd = e^{-t} W,
    cp = W exp(int_0^t (-K1 (1+cl) - K2 T) ds), cs = r_sp W exp(int_0^t -Kd (Sd-cd)(1+cl) ds)
    (reference src/prob1_mms_cases.py:280-337; the time integrals are used in closed form)."""

    mms_case_id = _ffi.MMS_EXPSIN

    def __init__(self, grid: p1.Grid, *, model: p1.DefaultModel01):
        super().__init__(grid, model)


class MMSCaseSlowlyChangingPeaks(p1.MMSCaseDevice):
    """All five unknowns equal Const (x^2+y^2)^3 sin(pi x) sin(pi y) e^{-a t}
    (reference src/prob1_mms_cases.py:151-199)."""

    mms_case_id = _ffi.MMS_PEAKS

    def __init__(self, grid: p1.Grid, model: p1.DefaultModel01, *, leading_spatial_const=1e1,
                 evol_speed: float = 1e-1):
        super().__init__(grid, model)
        self.leading_spatial_const = float(leading_spatial_const)
        self.evol_speed = float(evol_speed)

    def mms_params(self):
        return (self.leading_spatial_const, self.evol_speed)


def make_MMSCaseSlowlyChangingPeaks_cls(*, leading_spatial_const, evol_speed):
    class the_MMSCaseSlowlyChangingPeaks(MMSCaseSlowlyChangingPeaks):
        def __init__(self, grid: p1.Grid, model: p1.DefaultModel01):
            super().__init__(grid=grid, model=model, evol_speed=evol_speed,
                             leading_spatial_const=leading_spatial_const)

    return the_MMSCaseSlowlyChangingPeaks


MMSCaseSlowlyChangingPeaks_Slow1e1 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-1)
MMSCaseSlowlyChangingPeaks_Slow1e2 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-2)
MMSCaseSlowlyChangingPeaks_Slow1e3 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-3)
MMSCaseSlowlyChangingPeaks_Slow1e4 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-4)
MMSCaseSlowlyChangingPeaks_Slow1e8 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-8)
MMSCaseSlowlyChangingPeaks_Slow1e16 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e-16)
MMSCaseSlowlyChangingPeaks_Fast1e1 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e1)
MMSCaseSlowlyChangingPeaks_Fast1e2 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e2)
MMSCaseSlowlyChangingPeaks_Fast1e3 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e3)
MMSCaseSlowlyChangingPeaks_Fast1e4 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e4)
MMSCaseSlowlyChangingPeaks_Fast1e8 = make_MMSCaseSlowlyChangingPeaks_cls(leading_spatial_const=1.0, evol_speed=1e8)


class MMSCaseNonFullySmoothPol(p1.MMSCaseDevice):
    """u_v = x(1-x) y(1-y) |(x-theta)(y-theta)|^{gamma_v} / (1+t), one exponent per unknown
    (reference src/prob1_mms_cases.py:406-511).  gamma: one value (all unknowns), two values
    ((cp, cs) and (T, cl, cd)) or five values (cp, T, cl, cd, cs)."""

    mms_case_id = _ffi.MMS_NFS

    def __init__(self, grid: p1.Grid, *, model: p1.DefaultModel01, gamma: List[float], theta: float = 1 / np.pi):
        super().__init__(grid, model)
        if np.isscalar(gamma):
            gamma = [float(gamma)]
        assert isinstance(gamma, list), "gamma must be a single number or a list of numbers."
        assert len(gamma) in [1, 2, 5], "gamma must be a list of length 1, 2, or 5 numbers."
        if len(gamma) == 1:
            gamma = [gamma[0]] * 5
        elif len(gamma) == 2:
            gamma = [gamma[0], gamma[1], gamma[1], gamma[1], gamma[0]]
        assert all(gamma[j] > 1 for j in [0, 4]), "Cp's and cs' gamma (0, 4) must be greater than 1."
        assert all(gamma[j] > 2 for j in [1, 2, 3]), "T's, cl's, and cd's gammas (1, 2, 3) must be greater than 2."
        assert 0 < theta < 1, "Theta must be in (0, 1)."
        self.gamma = [float(g) for g in gamma]
        self.theta = float(theta)

    def mms_params(self):
        return tuple(self.gamma) + (self.theta,)


def make_MMSCaseNonFullySmoothPol_cls(gamma):
    class the_MMSCaseNonFullySmoothPol(MMSCaseNonFullySmoothPol):
        def __init__(self, grid: p1.Grid, model: p1.DefaultModel01):
            super().__init__(grid=grid, model=model, gamma=gamma)

    return the_MMSCaseNonFullySmoothPol


MMSCaseNonFullySmoothPol_cpcsH2_TclcdH3 = make_MMSCaseNonFullySmoothPol_cls(gamma=[2.1, 3.1])
MMSCaseNonFullySmoothPol_cpcsH1_TclcdH2 = make_MMSCaseNonFullySmoothPol_cls(gamma=[1.1, 2.1])
MMSCaseNonFullySmoothPol_cpcsH2_TclcdH2 = make_MMSCaseNonFullySmoothPol_cls(gamma=2.1)
MMSCaseNonFullySmoothPol_cpcsH3_TclcdH4 = make_MMSCaseNonFullySmoothPol_cls(gamma=[3.1, 4.1])
