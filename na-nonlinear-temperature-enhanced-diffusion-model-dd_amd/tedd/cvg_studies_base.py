"""Observed-convergence-rate helper of the reference's src/cvg_studies_base.py:25-106."""
from __future__ import annotations

import math
from typing import List, NamedTuple, Tuple

import numpy as np


class _RateStatus(NamedTuple):
    OK: str = "OK"
    INSUFFICIENT_DATA: str = "Insufficient Data"
    ZERO_DENOMINATOR_ZERO_NUMERATOR: str = "Differences near zero (converged/stalled?)"
    ZERO_DENOMINATOR_NONZERO_NUMERATOR: str = "Unstable rate (denominator near zero)"
    NON_POSITIVE_RATIO: str = "Non-positive ratio (convergence issue?)"
    ERROR_INCREASING: str = "Error increasing significantly"


RateStatus = _RateStatus()


def calculate_observed_rates(errors: List[float], refinement_factor: float = 2.0) -> List[Tuple[float, str]]:
    """3-point observed rates: log_r((E_k - E_{k+1}) / (E_{k+1} - E_{k+2})) for errors ordered from
    the coarsest to the finest level; each rate comes with a status string."""
    assert len(errors) >= 3, "At least 3 error values are required for rate calculation."
    assert refinement_factor > 1.0, "Refinement factor must be > 1.0"
    assert all(e >= 0 for e in errors), "All error values must be positive for rate calculation."
    log_r = math.log(refinement_factor)
    tiny = np.finfo(float).eps
    out = []
    for k in range(len(errors) - 2):
        num = errors[k] - errors[k + 1]
        den = errors[k + 1] - errors[k + 2]
        rate, status = np.nan, RateStatus.OK
        if den < 0:
            status = RateStatus.ERROR_INCREASING
        elif num <= 0:
            status = RateStatus.NON_POSITIVE_RATIO
        else:
            if abs(den) < tiny:
                status = (RateStatus.ZERO_DENOMINATOR_ZERO_NUMERATOR if abs(num) < tiny
                          else RateStatus.ZERO_DENOMINATOR_NONZERO_NUMERATOR)
            ratio = num / den if den != 0 else np.inf
            assert ratio > 0
            rate = math.log(ratio) / log_r
        out.append((rate, status))
    return out


# ------------------------------------------------------------------------------------------
# The study runner of the reference (src/cvg_studies_base.py:117-486) over the device path.
# ------------------------------------------------------------------------------------------
from . import mms_trial_utils as _mtu  # noqa: E402
from . import prob1base as p1  # noqa: E402

TimeStepData = _mtu.ErrorTimeSeries


def calculate_combined_error_norm(time_series_data, dt, integral_vars):
    """reference src/cvg_studies_base.py:224-250 (the same quantity as in mms_trial_utils)."""
    return _mtu.calculate_combined_error_norm(time_series_data, dt, integral_vars)


def run_simulation_collect_data(*, grid, integrator, exact_sol_pack, initial_state_gfp, Tf, dt, variable_names,
                                integral_vars, fused=True):
    """One run from t = 0 to Tf with the error record of every time level
    (reference src/cvg_studies_base.py:117-221; dt is adjusted so that the steps end at Tf).  When the
    integrator is a device predictor-corrector whose forcing the library evaluates itself, the whole loop and
    the norms run on the GPU; otherwise (or with fused=False) step by step, as in the reference."""
    field = getattr(integrator, "semi_discrete_field", None)
    fused = (fused and hasattr(integrator, "advance_with_errors") and isinstance(field, p1._DeviceField)
             and field._closed_form_source() is not None and field._closed_form_source().mms_case is exact_sol_pack
             and not (set(integral_vars) - set(_mtu._GRAD_ORDER)) and not (set(variable_names) - set(_mtu._SERIES_ORDER)))
    if fused:
        num_steps, dt_adj = _mtu._adjusted_steps(Tf, dt, 0.0)
        _final, err = integrator.advance_with_errors(initial_state_gfp, t0=0.0, dt=dt_adj, num_steps=num_steps)
        return _mtu.series_from_device(err, 0.0, dt_adj, variable_names, integral_vars), dt_adj
    return _mtu.run_simulation_collect_data(grid=grid, integrator=integrator, exact_sol_pack=exact_sol_pack,
                                            initial_state=initial_state_gfp, Tf=Tf, dt=dt, variable_names=variable_names,
                                            integral_vars=integral_vars)


def _setup_simulation_instances(*, field_cls, forcing_terms_cls, mms_case_cls, integrator_cls, grid, model,
                                variable_names, num_pc_steps, num_newton_steps):
    """MMS case, integrator and initial state of one level (reference src/cvg_studies_base.py:253-306)."""
    mms_case = mms_case_cls(grid=grid, model=model)
    forcing_terms = forcing_terms_cls(mms_case=mms_case, model=model)
    field = field_cls(grid=grid, model=model, forcing_terms=forcing_terms)
    integrator = integrator_cls(semi_discrete_field=field, num_pc_steps=num_pc_steps, num_newton_steps=num_newton_steps)
    xx, yy = grid.xx, grid.yy
    initial = p1.StateVars(**{v: getattr(mms_case, v)(0.0, xx, yy) for v in variable_names}, model=model,
                           hh=grid.hh, kk=grid.kk)
    return mms_case, integrator, initial


def run_convergence_studies(study_configs, study_params, verbose=True):
    """Spatial and temporal convergence studies for a list of
    (field_cls, mms_case_cls, forcing_terms_cls, integrator_cls, label)
    (reference src/cvg_studies_base.py:319-486).  study_params: Tf, model, N_base_spatial,
    num_spatial_refinements, dt_fixed_spatial, N_fixed_temporal, dt_base_temporal, num_temporal_refinements,
    and optionally variable_names, integral_vars, num_pc_steps, num_newton_steps, fused.  Returns
    {label: {"spatial" | "temporal": {"errors", "rates", "statuses"}}}."""
    say = print if verbose else (lambda *a, **k: None)
    variable_names = study_params.get("variable_names", ["cp", "T", "cl", "cd", "cs"])
    integral_vars = study_params.get("integral_vars", ["T", "cl", "cd"])
    Tf, model = study_params["Tf"], study_params["model"]
    common = dict(model=model, variable_names=variable_names, num_pc_steps=study_params.get("num_pc_steps", 1),
                  num_newton_steps=study_params.get("num_newton_steps", 1))
    factor = 2
    results = {}

    def one_run(grid, mms_case, integrator, initial, dt):
        series, dt_adj = run_simulation_collect_data(
            grid=grid, integrator=integrator, exact_sol_pack=mms_case, initial_state_gfp=initial, Tf=Tf, dt=dt,
            variable_names=variable_names, integral_vars=integral_vars, fused=study_params.get("fused", True))
        return calculate_combined_error_norm(series, dt_adj, integral_vars)

    def report(errors):
        rates = calculate_observed_rates(errors, factor)
        return {"errors": errors, "rates": [r for r, _s in rates], "statuses": [s for _r, s in rates]}

    for field_cls, mms_case_cls, forcing_terms_cls, integrator_cls, label in study_configs:
        say(f"\n===== Running Studies for Case: {label} =====")
        classes = dict(field_cls=field_cls, forcing_terms_cls=forcing_terms_cls, mms_case_cls=mms_case_cls,
                       integrator_cls=integrator_cls)
        spatial = []
        for k in range(study_params["num_spatial_refinements"]):
            N = study_params["N_base_spatial"] * factor ** k
            grid = p1.make_uniform_grid(N, N)
            mms_case, integrator, initial = _setup_simulation_instances(grid=grid, **classes, **common)
            spatial.append(one_run(grid, mms_case, integrator, initial, study_params["dt_fixed_spatial"]))
            say(f"  Spatial Level {k} (N=M={N}, dt={study_params['dt_fixed_spatial']:.1e}): {spatial[-1]:.6e}")
        N_fixed = study_params["N_fixed_temporal"]
        grid = p1.make_uniform_grid(N_fixed, N_fixed)
        mms_case, integrator, initial = _setup_simulation_instances(grid=grid, **classes, **common)
        temporal = []
        for k in range(study_params["num_temporal_refinements"]):
            dt = study_params["dt_base_temporal"] / factor ** k
            temporal.append(one_run(grid, mms_case, integrator, initial, dt))
            say(f"  Temporal Level {k} (dt={dt:.4e}): {temporal[-1]:.6e}")
        results[label] = {"spatial": report(spatial), "temporal": report(temporal)}
        say(f"===== Finished Studies for Case: {label} =====")
    return results
