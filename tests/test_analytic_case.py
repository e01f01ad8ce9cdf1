"""pack_analytical_txy_with_o2fdm_derivatives / MMSCaseFromAnalytic of the mirror
(reference src/prob1base.py:895-1155): the difference formulas against exact derivatives, the argument
checks, and the accessors of the case.  No device needed."""
import numpy as np
import pytest

import tedd.prob1base as p1


def f(t, x, y):
    return np.exp(0.7 * t) * np.sin(1.3 * x) * np.cos(0.9 * y) + t * t * x * y


EXACT = {
    (1, 0, 0): lambda t, x, y: 0.7 * np.exp(0.7 * t) * np.sin(1.3 * x) * np.cos(0.9 * y) + 2 * t * x * y,
    (2, 0, 0): lambda t, x, y: 0.49 * np.exp(0.7 * t) * np.sin(1.3 * x) * np.cos(0.9 * y) + 2 * x * y,
    (0, 1, 0): lambda t, x, y: 1.3 * np.exp(0.7 * t) * np.cos(1.3 * x) * np.cos(0.9 * y) + t * t * y,
    (0, 2, 0): lambda t, x, y: -1.69 * np.exp(0.7 * t) * np.sin(1.3 * x) * np.cos(0.9 * y),
    (0, 0, 1): lambda t, x, y: -0.9 * np.exp(0.7 * t) * np.sin(1.3 * x) * np.sin(0.9 * y) + t * t * x,
    (0, 0, 2): lambda t, x, y: -0.81 * np.exp(0.7 * t) * np.sin(1.3 * x) * np.cos(0.9 * y),
    (0, 1, 1): lambda t, x, y: -1.17 * np.exp(0.7 * t) * np.cos(1.3 * x) * np.sin(0.9 * y) + t * t,
}


@pytest.mark.parametrize("strategy", ["center", "forward", "backward"])
def test_difference_formulas(strategy):
    g = p1.pack_analytical_txy_with_o2fdm_derivatives(f, time_stepping=strategy)
    xx, yy = np.meshgrid(np.linspace(0, 1, 5), np.linspace(0, 1, 4), indexing="ij")
    t = 0.4
    np.testing.assert_array_equal(g(t, xx, yy), f(t, xx, yy))
    for d, exact in EXACT.items():
        order = sum(d)
        # second-order formulas with step eps: truncation eps^2, rounding 1e-16 / eps^order
        for eps, tol in ((1e-3, 2e-5), (1e-4, 1e-6 if order == 1 else 2e-6)):
            np.testing.assert_allclose(g(t, xx, yy, d=d, small_eps=eps), exact(t, xx, yy), rtol=0, atol=tol,
                                       err_msg=f"{strategy} {d} {eps}")
    lap = EXACT[(0, 2, 0)](t, xx, yy) + EXACT[(0, 0, 2)](t, xx, yy)
    for op in ("lap", "Laplacian"):
        np.testing.assert_allclose(g(t, xx, yy, op=op, small_eps=1e-4), lap, rtol=0, atol=2e-6)
    # a time derivative takes precedence over space derivatives asked for at the same time
    np.testing.assert_array_equal(g(t, xx, yy, d=(1, 1, 0)), g(t, xx, yy, d=(1, 0, 0)))


def test_argument_checks():
    g = p1.pack_analytical_txy_with_o2fdm_derivatives(f)
    with pytest.raises(ValueError):
        g(0.0, 0.5, 0.5, d=(3, 0, 0))
    with pytest.raises(ValueError):
        g(0.0, 0.5, 0.5, d=(1, 1, 1))
    with pytest.raises(ValueError):
        g(0.0, 0.5, 0.5, op="grad")
    with pytest.raises(ValueError):
        p1.pack_analytical_txy_with_o2fdm_derivatives(f, time_stepping="upwind")


def test_case_accessors():
    import api_cases as ac
    model = ac.model_of("nb_b")
    grid = p1.make_uniform_grid(6, 5)
    case = p1.MMSCaseFromAnalytic(model, grid=grid, cp_base=f, T_base=f, cl_base=f, cd_base=f, cs_base=f)
    assert case.grid is grid and case.model is model
    t, xx, yy = 0.2, grid.xx, grid.yy
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_array_equal(getattr(case, v)(t, xx, yy), f(t, xx, yy))
        np.testing.assert_allclose(getattr(case, "dt_" + v)(t, xx, yy), EXACT[(1, 0, 0)](t, xx, yy), atol=1e-8)
        np.testing.assert_allclose(getattr(case, "dx_" + v)(t, xx, yy), EXACT[(0, 1, 0)](t, xx, yy), atol=1e-8)
        np.testing.assert_allclose(getattr(case, "dy_" + v)(t, xx, yy), EXACT[(0, 0, 1)](t, xx, yy), atol=1e-8)
    for v in ("T", "cl", "cd"):
        lap = EXACT[(0, 2, 0)](t, xx, yy) + EXACT[(0, 0, 2)](t, xx, yy)
        np.testing.assert_allclose(getattr(case, "lap_" + v)(t, xx, yy), lap, atol=2e-3)   # eps = 1e-6
