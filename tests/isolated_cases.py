"""Truncation-error and coupled-subsystem convergence cases with degenerate model constants
(the scenarios of reference tests/test_spatial_isolated_T_accuracy.py:88-741), written against a module pair
(p1 = prob1base, mtu = mms_trial_utils) passed in by the caller: tests/golden/make_golden.py runs them
with the reference's modules and stores the numbers, the tests run them with the mirrored modules on the
device path and compare.

The continuous right-hand sides are stated here from the PDE system itself:
    T_t  = DT lap T + K3 cp (.)                      (only used with K3 = 0)
    cl_t = div(Dl(cp) grad cl) - d/dx(V1(T) (cl + 1)) - K4 cp (cl + 1)
    cd_t = div(Dd(cp, T) grad cd) + Kd (Sd - cd) (cl + 1) cs
"""
import numpy as np
import sympy

ZERO = dict(K1=0.0, K2=0.0, K3=0.0, K4=0.0, DT=0.0, Dl_max=0.0, phi_l=0.0, gamma_T=0.0, Kd=0.0, Sd=0.0,
            Dd_max=0.0, phi_d=0.0, phi_T=0.0, r_sp=0.0)


def make_model(p1, **consts):
    c = dict(ZERO, **consts)
    if c.get("phi_T") == "Ea/R0":
        c["phi_T"] = p1.Ea / p1.R0
    return p1.DefaultModel02(p1.ModelConsts(R0=p1.R0, Ea=p1.Ea, **c))


def _sym(p1):
    return p1.t_sym, p1.x_sym, p1.y_sym, sympy.pi


def operator_case(p1, which):
    """(model, expressions {cp,T,cl,cd,cs}, continuous right-hand side, time) of one isolated operator."""
    t, x, y, pi = _sym(p1)
    zero = sympy.S(0)
    if which == "FT":
        model = make_model(p1, DT=1e-3)
        ex = dict(cp=zero, T=sympy.sin(pi * x) * sympy.sin(pi * y) * (1 + t), cl=zero, cd=zero, cs=zero)
        rhs = model.DT * (sympy.diff(ex["T"], x, 2) + sympy.diff(ex["T"], y, 2))
        return model, ex, rhs, 0.0
    if which == "Fcl":
        model = make_model(p1, K4=1e-2, Dl_max=8.0e-4, phi_l=1e-5, gamma_T=1e-9)
        ex = dict(cp=sympy.sin(pi * x) ** 2 * sympy.sin(pi * y) ** 2 * (1 + 0.5 * t),
                  T=sympy.cos(0.5 * pi * x) * sympy.cos(0.5 * pi * y) * (1 + t),
                  cl=sympy.sin(pi * x) * sympy.sin(pi * y) * (1 + t ** 2), cd=zero, cs=zero)
        Dl = model.Dl_max * sympy.exp(-model.phi_l * ex["cp"])
        V1 = model.gamma_T * ex["T"]
        cl = ex["cl"]
        rhs = (sympy.diff(Dl * sympy.diff(cl, x), x) + sympy.diff(Dl * sympy.diff(cl, y), y)
               - sympy.diff(V1 * (cl + 1), x) - model.K4 * ex["cp"] * (cl + 1))
        return model, ex, rhs, 0.1
    if which == "Fcd":
        model = make_model(p1, Kd=1e-3, Sd=10.0, Dd_max=2.5e-6, phi_d=1e-5, phi_T="Ea/R0")
        ex = dict(cp=sympy.sin(pi * x) ** 2 * sympy.sin(pi * y) ** 2 * (3 + t),
                  T=300 * (1 + 0.1 * sympy.sin(pi * x) * sympy.sin(pi * y) * (1 + 0.1 * t)),
                  cl=sympy.sin(pi * x) * sympy.sin(2 * pi * y) * (1 + t),
                  cd=sympy.sin(2 * pi * x) * sympy.sin(pi * y) * (1 + 0.5 * t),
                  cs=sympy.cos(0.5 * pi * x) * sympy.cos(0.5 * pi * y) * (2 + t))
        Dd = model.Dd_max * sympy.exp(-model.phi_d * ex["cp"]) * sympy.exp(-model.phi_T / (ex["T"] + model.T_ref))
        cd = ex["cd"]
        rhs = (sympy.diff(Dd * sympy.diff(cd, x), x) + sympy.diff(Dd * sympy.diff(cd, y), y)
               + model.Kd * (model.Sd - cd) * (ex["cl"] + 1) * ex["cs"])
        return model, ex, rhs, 0.1
    raise KeyError(which)


def operator_truncation_errors(p1, which, N_list=(8, 16, 32, 64)):
    """||F_which(exact state) - continuous right-hand side||_H on uniform meshes, no forcing terms."""
    t, x, y, _pi = _sym(p1)
    model, ex, rhs, when = operator_case(p1, which)
    rhs_fn = sympy.lambdify((t, x, y), rhs, "numpy")
    out = []
    for N in N_list:
        grid = p1.make_uniform_grid(N, N)
        case = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=ex["cp"], T_sym_expr=ex["T"],
                                  cl_sym_expr=ex["cl"], cd_sym_expr=ex["cd"], cs_sym_expr=ex["cs"],
                                  t_var=t, x_var=x, y_var=y)
        field = p1.SemiDiscreteField_CsTriple(grid=grid, model=model, forcing_terms=p1.NoForcingTerms(grid=grid))
        state = p1.StateVars(**{v: getattr(case, v)(when, grid.xx, grid.yy) for v in ("cp", "T", "cl", "cd", "cs")},
                             model=model, hh=grid.hh, kk=grid.kk)
        numeric = getattr(field, which)(state, when)
        exact = np.broadcast_to(np.asarray(rhs_fn(when, grid.xx, grid.yy), dtype=float), grid.xx.shape)
        out.append(float(grid.norm_H(numeric - exact)))
    return out


def coupled_case(p1, which):
    t, x, y, pi = _sym(p1)
    zero = sympy.S(0)
    if which == "T_cl":
        model = make_model(p1, K3=1e-3, K4=5e-3, DT=1e-3, Dl_max=8.0e-4, gamma_T=1e-9)
        bump = (1 - x) * x * (1 - y) * y / (1 + t)
        return model, dict(cp=zero, T=bump, cl=bump, cd=zero, cs=zero), ("T", "cl")
    if which == "T_cl_cd":
        model = make_model(p1, K4=1e-3, DT=1e-3, Dl_max=8.0e-4, gamma_T=1e-9, Kd=5e-4, Sd=10.0, Dd_max=2.5e-6,
                           phi_T="Ea/R0")
        ex = dict(cp=zero, T=5 * sympy.sin(pi * x) * sympy.sin(2 * pi * y) * (1 + t),
                  cl=(sympy.sin(pi * x) * sympy.sin(pi * y) * t) ** 2,
                  cd=sympy.sin(2 * pi * x) * sympy.sin(pi * y) * t, cs=zero)
        return model, ex, ("T", "cl", "cd")
    raise KeyError(which)


def coupled_errors(p1, mtu, which, N_list=(16, 32, 64, 128), Tf=1e-4, dt_const=1e-2):
    """{variable: [sup-in-time H-norm errors]} of the CsTriple predictor-corrector with dt = dt_const h^1.5."""
    t, x, y, _pi = _sym(p1)
    model, ex, variables = coupled_case(p1, which)

    class Case(p1.MMSCaseSymbolic):
        def __init__(self, *, grid, model):
            super().__init__(grid=grid, model=model, cp_sym_expr=ex["cp"], T_sym_expr=ex["T"], cl_sym_expr=ex["cl"],
                             cd_sym_expr=ex["cd"], cs_sym_expr=ex["cs"], t_var=t, x_var=x, y_var=y)

    out = {v: [] for v in variables}
    for N in N_list:
        trial = mtu.MMSTrial(grid=p1.make_uniform_grid(N, N), model=model, mms_case_cls=Case,
                             field_cls=p1.SemiDiscreteField_CsTriple, forcing_terms_cls=p1.ForcingTerms_CsTriple,
                             integrator_cls=p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple)
        summary = trial.run_for_errors(Tf=Tf, dt=dt_const * (1.0 / N) ** 1.5)
        for v in variables:
            out[v].append(float(summary.per_variable_sup_errors[v]))
    return out


def gradient_T_one_step_errors(p1, N_list=(8, 16, 32, 64), dt_const=0.1):
    """p-norm of the error of grad_H(T) after one predictor-corrector step of dt = dt_const h^2, heat equation
    only (DT the one non-zero constant); the field gets a dictionary of forcing callables: fT of the
    CsTriple forcing terms, zeros for the rest
    (the scenario of reference tests/test_spatial_h1_isolated_T_accuracy.py:57-173)."""
    t, x, y, pi = _sym(p1)
    model = make_model(p1, DT=1e-3)
    zero = sympy.S(0)
    T = sympy.sin(pi * x) * sympy.sin(pi * y) * sympy.exp(-t)
    out = []
    for N in N_list:
        dt = dt_const / N ** 2
        grid = p1.make_uniform_grid(N, N)
        case = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=zero, T_sym_expr=T, cl_sym_expr=zero,
                                  cd_sym_expr=zero, cs_sym_expr=zero, t_var=t, x_var=x, y_var=y)
        full = p1.ForcingTerms_CsTriple(model=model, mms_case=case)
        nothing = lambda t_, xx, yy, g=grid: g.make_full0()
        forcing = p1.ForcingTermsFromDict({"fcp": nothing, "fT": full.fT, "fcl": nothing, "fcd": nothing, "fcs": nothing})
        field = p1.SemiDiscreteField_CsTriple(grid=grid, model=model, forcing_terms=forcing)
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple(semi_discrete_field=field, num_pc_steps=1,
                                                                         num_newton_steps=1)
        S0 = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
        S1 = integ.step(S0, t0=0.0, dt=dt)
        gx, gy = grid.grad_H(S1.T)
        ex, ey = grid.grad_H(case.T(dt, grid.xx, grid.yy))
        out.append(float(grid.norm_p(gx - ex, gy - ey)))
    return out


def forward_euler_T_cl_errors(p1, N_list=(8, 16, 32, 64), Tf=1e-3, cfl=0.1):
    """{T, cl: H-norm errors at Tf} of forward Euler on the coupled T-cl subsystem (DefaultModel01, cp = cd =
    cs = 0) with dt = cfl h^2 / max(DT, Dl), adjusted to end at Tf
    (the scenario of reference tests/test_feuler_spatial_accuracy.py:57-186)."""
    import math
    t, x, y, pi = _sym(p1)
    zero = sympy.S(0)
    consts = dict(ZERO, K3=1e-3, K4=5e-3, DT=1e-3, Dl_max=8.0e-4, gamma_T=1e-9)
    model = p1.DefaultModel01(p1.ModelConsts(R0=p1.R0, Ea=p1.Ea, **consts))
    T = 10 * sympy.sin(pi * x) * sympy.sin(pi * y) * (1 + t)
    cl = (sympy.cos(pi * x) ** 2 - 1) * (1 - sympy.cos(pi * y) ** 2) * t
    out = {"T": [], "cl": []}
    for N in N_list:
        grid = p1.make_uniform_grid(N, N)
        case = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=zero, T_sym_expr=T, cl_sym_expr=cl,
                                  cd_sym_expr=zero, cs_sym_expr=zero, t_var=t, x_var=x, y_var=y)
        forcing = p1.ForcingTerms_CsTriple(mms_case=case, model=model)
        field = p1.SemiDiscreteField_CsTriple(grid=grid, model=model, forcing_terms=forcing)
        integ = p1.ForwardEulerIntegrator(semi_discrete_field=field)
        steps = math.ceil(Tf / (cfl / N ** 2 / max(model.DT, float(model.Dl(0.0)))))
        dt = Tf / steps
        S, now = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid), 0.0
        for _ in range(steps):
            S = integ.step(S, t0=now, dt=dt)
            now += dt
        for v in out:
            out[v].append(float(grid.norm_H(getattr(S, v) - getattr(case, v)(Tf, grid.xx, grid.yy))))
    return out
