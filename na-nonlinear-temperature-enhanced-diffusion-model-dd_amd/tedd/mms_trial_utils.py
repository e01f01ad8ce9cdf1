"""MMS trial harness: same entry points as the reference's src/mms_trial_utils.py.

    import tedd.mms_trial_utils as mtu
    trial = mtu.MMSTrial(grid, model, mms_case_cls, field_cls, forcing_terms_cls, integrator_cls, ...)
    summary = trial.run_for_errors(Tf, dt)

`run_simulation_collect_data` keeps the reference's step-by-step loop (one `integrator.step`,
one error record per step) for any integrator.  When the integrator is a device
predictor-corrector with closed-form MMS forcing, `MMSTrial.run_for_errors` runs the whole
time loop and the per-step error norms on the GPU (`advance_with_errors`) and only the
(num_steps+1) x 8 norm series comes back.
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple, Optional, Tuple, Type

import numpy as np

from . import prob1base as p1


class ErrorTimeSeries(NamedTuple):
    t: float
    h_norm_sq_errors: Dict[str, float]
    grad_h_norm_p_sq_errors: Dict[str, float]


def calculate_combined_error_norm(time_series_data: List[ErrorTimeSeries], dt: float, integral_vars: List[str],
                                  all_variables: Optional[List[str]] = None) -> float:
    """max_k sqrt( sum_v ||e_v(t_k)||_H^2 + int_0^{t_k} sum_{v in integral_vars} ||grad e_v||_p^2 dt ),
    the time integral by the trapezoidal rule (reference src/mms_trial_utils.py:15-53)."""
    if all_variables is not None:
        assert all(v in all_variables for v in integral_vars), "integral_vars must be a subset of all_variables."
    integrand = [sum(rec.grad_h_norm_p_sq_errors[v] for v in integral_vars) for rec in time_series_data]
    best = 0.0
    running = 0.0
    for k, rec in enumerate(time_series_data):
        if all_variables is None:
            h_sq = sum(rec.h_norm_sq_errors.values())
        else:
            h_sq = sum(rec.h_norm_sq_errors[v] for v in all_variables)
        if k > 0:
            running += 0.5 * dt * (integrand[k - 1] + integrand[k])
        best = max(best, h_sq + running)
    return np.sqrt(best)


def _adjusted_steps(Tf, dt, t0):
    num_steps = math.ceil((Tf - t0) / dt)
    return num_steps, (Tf - t0) / num_steps


def run_simulation_collect_data(*, grid: p1.Grid, integrator, exact_sol_pack, initial_state: p1.StateVars,
                                Tf: float, dt: float, t0: float = 0.0, variable_names: List[str],
                                integral_vars: List[str]) -> Tuple[List[ErrorTimeSeries], float]:
    """The reference's time loop (src/mms_trial_utils.py:56-147): one step, one error record."""
    num_steps, dt = _adjusted_steps(Tf, dt, t0)

    def record(state, t):
        exact = p1.state_from_mms_when(mms_case=exact_sol_pack, t=t, grid=grid)
        h_sq, g_sq = {}, {}
        for name in variable_names:
            num, ex = getattr(state, name), getattr(exact, name)
            h_sq[name] = grid.norm_H(num - ex) ** 2
            if name in integral_vars:
                gn, ge = grid.grad_H(num), grid.grad_H(ex)
                g_sq[name] = grid.norm_p(gn[0] - ge[0], gn[1] - ge[1]) ** 2
            else:
                g_sq[name] = 0.0
        return ErrorTimeSeries(t=t, h_norm_sq_errors=h_sq, grad_h_norm_p_sq_errors=g_sq)

    t = t0
    state = initial_state
    series = [record(state, t)]
    for _ in range(num_steps):
        state = integrator.step(state, t0=t, dt=dt)
        t += dt
        series.append(record(state, t))
    assert np.isclose(t, Tf), f"Final time mismatch: current_t={t}, Tf={Tf}"
    return series, dt


_SERIES_ORDER = ("cp", "T", "cl", "cd", "cs")
_GRAD_ORDER = ("T", "cl", "cd")


def series_from_device(err_series: np.ndarray, t0: float, dt: float, variable_names, integral_vars):
    """Turns the (num_steps+1) x 8 device norm series into ErrorTimeSeries records."""
    out = []
    t = t0
    for k in range(err_series.shape[0]):
        row = err_series[k]
        h_sq = {v: float(row[_SERIES_ORDER.index(v)]) for v in variable_names}
        g_sq = {v: (float(row[5 + _GRAD_ORDER.index(v)]) if v in integral_vars else 0.0) for v in variable_names}
        out.append(ErrorTimeSeries(t=t, h_norm_sq_errors=h_sq, grad_h_norm_p_sq_errors=g_sq))
        t += dt
    return out


class NumericalErrorSummary:
    """Overall combined error and per-variable sup errors (reference src/mms_trial_utils.py:150-198)."""

    def __init__(self, dt_used: float, time_series_data: List[ErrorTimeSeries], variable_names: List[str],
                 integral_vars: List[str]):
        self.dt_used = dt_used
        self.variable_names = variable_names
        self.integral_vars = integral_vars
        if not time_series_data:
            raise ValueError("time_series_data cannot be empty.")
        self.overall_combined_error: float = calculate_combined_error_norm(time_series_data, dt_used, integral_vars)
        self.per_variable_sup_errors: Dict[str, float] = {}
        for name in variable_names:
            iv = [name] if name in integral_vars else []
            self.per_variable_sup_errors[name] = calculate_combined_error_norm(
                time_series_data, dt_used, integral_vars=iv, all_variables=[name])

    def __repr__(self):
        per = {k: f"{v:.4e}" for k, v in self.per_variable_sup_errors.items()}
        return (f"NumericalErrorSummary(dt={self.dt_used:.2e}, OverallCombinedError={self.overall_combined_error:.4e}, "
                f"PerVariableSupErrors={per})")


class MMSTrial:
    """Set-up and execution of one MMS trial (reference src/mms_trial_utils.py:201-280)."""

    def __init__(self, grid: p1.Grid, model, mms_case_cls: Type, field_cls: Type, forcing_terms_cls: Type,
                 integrator_cls: Type, mms_case_params: Optional[Dict] = {}, integrator_params: Optional[Dict] = {},
                 forcing_terms_params: Optional[Dict] = {}, field_params: Optional[Dict] = {},
                 variable_names: List[str] = None, integral_vars: List[str] = None, fused: bool = True):
        self.grid = grid
        self.model = model
        self.mms_case_cls = mms_case_cls
        self.field_cls = field_cls
        self.forcing_terms_cls = forcing_terms_cls
        self.integrator_cls = integrator_cls
        self.variable_names = variable_names or ["cp", "T", "cl", "cd", "cs"]
        self.integral_vars = integral_vars or ["T", "cl", "cd"]
        self.fused = fused
        self.mms_case = mms_case_cls(grid=self.grid, model=self.model, **mms_case_params)
        self.forcing_terms = forcing_terms_cls(mms_case=self.mms_case, model=self.model, **forcing_terms_params)
        self.field = field_cls(grid=self.grid, model=self.model, forcing_terms=self.forcing_terms, **field_params)
        self.integrator = integrator_cls(semi_discrete_field=self.field, **integrator_params)
        self.initial_state = p1.state_from_mms_when(mms_case=self.mms_case, t=0.0, grid=self.grid)

    def _can_fuse(self):
        if not self.fused or not hasattr(self.integrator, "advance_with_errors"):
            return False
        if set(self.integral_vars) - set(_GRAD_ORDER) or set(self.variable_names) - set(_SERIES_ORDER):
            return False
        field = self.integrator.semi_discrete_field
        return isinstance(field, p1._DeviceField) and field._closed_form_source() is not None \
            and field._closed_form_source().mms_case is self.mms_case

    def run_for_errors(self, Tf: float, dt: float, t0: float = 0.0) -> NumericalErrorSummary:
        if self._can_fuse():
            num_steps, dt_adj = _adjusted_steps(Tf, dt, t0)
            _final, err = self.integrator.advance_with_errors(self.initial_state, t0=t0, dt=dt_adj,
                                                              num_steps=num_steps)
            series = series_from_device(err, t0, dt_adj, self.variable_names, self.integral_vars)
        else:
            series, dt_adj = run_simulation_collect_data(
                grid=self.grid, integrator=self.integrator, exact_sol_pack=self.mms_case,
                initial_state=self.initial_state, Tf=Tf, dt=dt, t0=t0, variable_names=self.variable_names,
                integral_vars=self.integral_vars)
        return NumericalErrorSummary(dt_used=dt_adj, time_series_data=series, variable_names=self.variable_names,
                                     integral_vars=self.integral_vars)
