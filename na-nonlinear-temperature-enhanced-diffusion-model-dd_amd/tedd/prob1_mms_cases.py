"""The MMS case library of the reference (src/prob1_mms_cases.py) for the cases whose exact
solutions are built into the device library: same class names and constructor arguments.

    import tedd.prob1_mms_cases as p1mc
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)

Each class only records which closed form to use and its parameters; the values (exact solution,
its derivatives, and the forcing terms derived from them) are computed on the GPU.
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import _ffi
from . import prob1base as p1


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
