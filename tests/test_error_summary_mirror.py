"""NumericalErrorSummary / calculate_combined_error_norm of the mirror (reference src/mms_trial_utils.py)
on hand-made error records with hand-computed answers: the known-answer cases of the reference's
tests/test_mms_trial_utils.py:116-404.  Host only.

    combined^2(t_n) = sum_v H0_v^2(t_n) + sum_{v integral} trapezoid_0^{t_n} H1_v^2
    per variable:    sup_n sqrt(H0_v^2(t_n) [+ trapezoid_0^{t_n} H1_v^2 if v is integral])
"""
import math

import pytest

import tedd.mms_trial_utils as mtu


def series(dt, rows):
    """rows: per time level ({var: H0^2}, {var: H1^2})."""
    return [mtu.ErrorTimeSeries(t=k * dt, h_norm_sq_errors=dict(h0), grad_h_norm_p_sq_errors=dict(h1))
            for k, (h0, h1) in enumerate(rows)], dt


BASIC = (0.1, [({"T": 1.0, "cp": 0.25}, {"T": 4.0, "cp": 0.0}),
               ({"T": 0.5, "cp": 0.1}, {"T": 2.0, "cp": 0.0}),
               ({"T": 0.2, "cp": 0.05}, {"T": 1.0, "cp": 0.0})])
T0_ZERO = (0.1, [({"T": 0.0, "cp": 0.0}, {"T": 0.0, "cp": 0.0})] + BASIC[1][1:])
ZERO_GRAD = (0.1, [(h0, {"T": 0.0, "cp": 0.0}) for h0, _h1 in BASIC[1]])
ONE_STEP = (0.5, [({"T": 1.0, "cl": 0.8, "cp": 0.25}, {"T": 4.0, "cl": 2.0, "cp": 0.0}),
                  ({"T": 0.5, "cl": 0.4, "cp": 0.1}, {"T": 2.0, "cl": 1.0, "cp": 0.0})])

# (records, variables, integral variables, overall^2, {variable: sup^2})
KNOWN = {
    # t=0: 1.25; t=.1: .6 + .05(4+2) = .9; t=.2: .25 + .3 + .05(2+1) = .7
    "basic": (BASIC, ["T", "cp"], ["T"], 1.25, {"T": 1.0, "cp": 0.25}),
    # t=0: 0; t=.1: .6 + .05(0+2) = .7; t=.2: .25 + .1 + .15 = .5;  T alone: 0, .6, .45
    "zero_initial_error": (T0_ZERO, ["T", "cp"], ["T"], 0.7, {"T": 0.6, "cp": 0.1}),
    "zero_gradient_error": (ZERO_GRAD, ["T", "cp"], ["T"], 1.25, {"T": 1.0, "cp": 0.25}),
    # t=0: 2.05; t=.5: 1.0 + .25(4+2) + .25(2+1) = 3.25;  T: 1, .5+1.5;  cl: .8, .4+.75
    "one_step": (ONE_STEP, ["T", "cl", "cp"], ["T", "cl"], 3.25, {"T": 2.0, "cl": 1.15, "cp": 0.25}),
}


@pytest.mark.parametrize("name", list(KNOWN))
def test_known_answers(name):
    (dt, rows), variables, integral, overall_sq, per_var_sq = KNOWN[name]
    data, dt = series(dt, rows)
    summary = mtu.NumericalErrorSummary(dt_used=dt, time_series_data=data, variable_names=variables,
                                        integral_vars=integral)
    assert summary.dt_used == dt
    assert summary.overall_combined_error == pytest.approx(math.sqrt(overall_sq))
    assert set(summary.per_variable_sup_errors) == set(variables)
    for v, sq in per_var_sq.items():
        assert summary.per_variable_sup_errors[v] == pytest.approx(math.sqrt(sq)), v
    assert mtu.calculate_combined_error_norm(data, dt, integral, all_variables=variables) == \
        pytest.approx(math.sqrt(overall_sq))


def test_empty_series_is_rejected():
    with pytest.raises(ValueError, match="time_series_data cannot be empty."):
        mtu.NumericalErrorSummary(dt_used=0.1, time_series_data=[], variable_names=["T"], integral_vars=["T"])


def test_variable_set_checks():
    data = [mtu.ErrorTimeSeries(t=0.0, h_norm_sq_errors={"A": 1}, grad_h_norm_p_sq_errors={"A": 1})]
    assert mtu.calculate_combined_error_norm(data, 0.1, integral_vars=[], all_variables=["A"]) == pytest.approx(1.0)
    with pytest.raises(AssertionError, match="integral_vars must be a subset of all_variables."):
        mtu.calculate_combined_error_norm(data, 0.1, integral_vars=["C"], all_variables=["A", "B"])
    with pytest.raises(Exception):    # "B" has no record
        mtu.calculate_combined_error_norm(data, 0.1, integral_vars=["A"], all_variables=["A", "B"])
