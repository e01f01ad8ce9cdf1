"""RegHCsTriple scenarios (those of reference tests/test_reghcstriple.py:109-577: ExpSin case, DefaultModel02
with Kd = 1e-4, regularisation factors 10 / 50 / 100) written against a module triple passed in by the
caller: tests/golden/make_golden.py runs them with the reference's modules and stores the numbers, the
tests run them with the mirrored modules on the device path and compare."""
import numpy as np

ETAS = (10.0, 50.0, 100.0)
SIZES = (8, 16, 32)
FIELDS = ("cp", "T", "cl", "cd", "cs")


def model_of(p1):
    return p1.DefaultModel02(mc=p1.ModelConsts(
        R0=8.3144621, Ea=1.60217662e-19, K1=1e-2, K2=1e-2, K3=1e-2, K4=1e-2, DT=1e-3, Dl_max=8.01e-4, phi_l=1e-5,
        gamma_T=1e-9, Kd=1e-4, Sd=10, Dd_max=2.46e-6, phi_d=1e-5, phi_T=1.60217662e-19 / 8.3144621, r_sp=5e-2,
        T_ref=300))


def setup(p1, p1mc, N, eta, case_grid_N=16):
    """(grid, model, case, field); as in the reference's fixtures the case object may live on another grid
    than the field (it is only evaluated at the coordinates it is handed)."""
    model = model_of(p1)
    grid = p1.make_uniform_grid(N=N, M=N)
    case = p1mc.MMSCaseExpSin(grid=grid if case_grid_N is None else p1.make_uniform_grid(N=case_grid_N, M=case_grid_N),
                              model=model)
    forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=eta)
    field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing, regularization_factor=eta)
    return grid, model, case, field


def heaviside_values(p1, eta):
    return [float(v) for v in p1.heaviside_regularized(np.linspace(-5, 5, 101), eta)]


def field_vs_time_derivative(p1, p1mc, N, eta, t=0.1):
    """max interior |F_v(exact state) - d/dt v| per variable (field and case on the same grid)."""
    grid, _model, case, field = setup(p1, p1mc, N, eta, case_grid_N=None)
    S = p1.state_from_mms_when(mms_case=case, t=t, grid=grid)
    out = {}
    for v in FIELDS:
        F = getattr(field, "F" + v)(S, t)
        assert np.all(np.isfinite(F))
        out[v] = float(np.max(np.abs(F[1:-1, 1:-1] - getattr(case, "dt_" + v)(t, grid.xx, grid.yy)[1:-1, 1:-1])))
    return out


def FT_refinement_errors(p1, p1mc, eta, t=0.1):
    """max interior |FT^T - dt_T| on 8 ... 64 with the case object of a 16 x 16 grid evaluated on each mesh
    (the transpose is the reference's: ExpSin is symmetric in x and y on square meshes)."""
    out = []
    for N in (8, 16, 32, 64):
        grid, _model, case, field = setup(p1, p1mc, N, eta)
        S = p1.state_from_mms_when(mms_case=case, t=t, grid=grid)
        FT = field.FT(S, t)
        out.append(float(np.max(np.abs(FT.T[1:-1, 1:-1] - case.dt_T(t, grid.xx, grid.yy)[1:-1, 1:-1]))))
    return out


def one_step_errors(p1, p1mc, N, eta, kind, dt, t0=0.05):
    """max interior error per variable after one step from the exact state (kind: "fe" | "pc")."""
    grid, _model, case, field = setup(p1, p1mc, N, eta, case_grid_N=None)
    if kind == "fe":
        integ = p1.ForwardEulerIntegrator(semi_discrete_field=field)
    else:
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(
            semi_discrete_field=field, num_pc_steps=1, num_newton_steps=1, regularization_factor=eta,
            num_newton_iterations=5, consec_xs_rtol=1e-6)
    S1 = integ.step(at_t0=p1.state_from_mms_when(mms_case=case, t=t0, grid=grid), t0=t0, dt=dt)
    E1 = p1.state_from_mms_when(mms_case=case, t=t0 + dt, grid=grid)
    return {v: float(np.max(np.abs(getattr(S1, v)[1:-1, 1:-1] - getattr(E1, v)[1:-1, 1:-1]))) for v in FIELDS}


def forward_euler_temporal_errors(p1, p1mc, eta, N=32, t0=0.1, base_dt=1e-3):
    grid, _model, case, field = setup(p1, p1mc, N, eta)
    integ = p1.ForwardEulerIntegrator(semi_discrete_field=field)
    out = []
    for k in range(4):
        dt = base_dt / 2.0 ** k
        S1 = integ.step(at_t0=p1.state_from_mms_when(mms_case=case, t=t0, grid=grid), t0=t0, dt=dt)
        out.append(float(grid.norm_H(S1.T - case.T(t0 + dt, grid.xx, grid.yy))))
    return out


def short_trials(p1, p1mc, mtu, eta, N=4, Tf=1e-3, steps=1000):
    """Overall error of 1000 steps to Tf = 1e-3 on a 4 x 4 mesh with forward Euler and the predictor-corrector."""
    grid = p1.make_uniform_grid(N=N, M=N)
    common = dict(grid=grid, model=model_of(p1), mms_case_cls=p1mc.MMSCaseExpSin,
                  field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                  forcing_terms_params={"regularization_factor": eta}, field_params={"regularization_factor": eta})
    fe = mtu.MMSTrial(integrator_cls=p1.ForwardEulerIntegrator, **common)
    pc = mtu.MMSTrial(integrator_cls=p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple,
                      integrator_params={"regularization_factor": eta, "num_newton_iterations": 5,
                                         "consec_xs_rtol": 1e-6}, **common)
    return {"fe": float(fe.run_for_errors(Tf=Tf, dt=Tf / steps).overall_combined_error),
            "pc": float(pc.run_for_errors(Tf=Tf, dt=Tf / steps).overall_combined_error)}


def everything(p1, p1mc, mtu):
    out = {}
    for eta in ETAS:
        e = out[str(eta)] = {}
        e["heaviside"] = heaviside_values(p1, eta)
        e["field_vs_dt"] = {str(N): field_vs_time_derivative(p1, p1mc, N, eta) for N in SIZES}
        e["FT_refinement"] = FT_refinement_errors(p1, p1mc, eta)
        e["fe_one_step"] = {f"{N}/{dt}": one_step_errors(p1, p1mc, N, eta, "fe", dt) for N in SIZES for dt in (1e-3, 2.5e-4)}
        e["pc_one_step"] = {str(N): one_step_errors(p1, p1mc, N, eta, "pc", 1e-3) for N in SIZES}
        e["fe_temporal"] = forward_euler_temporal_errors(p1, p1mc, eta)
        e["short_trials"] = short_trials(p1, p1mc, mtu, eta)
    return out
