"""Test bodies for the Python mirror of the reference API (tedd.prob1base & co.).  They read like
the reference's own tests (tests/test_reghcstriple_system.py, tests/test_reghcstriple.py,
tests/test_time_integration_fwd_euler_full_p1base.py, tests/test_newton_residuals.py) and are
run twice: on the host twin (CPU suite, small sizes) and on the GPU (full sizes)."""
import json
import os

import numpy as np
import pytest

import analytic_fns
import tedd.mms_trial_utils as mtu
import tedd.prob1_mms_cases as p1mc
import tedd.prob1base as p1
from tedd.utils_for_testing import observed_rates_report

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_SETS = json.load(open(os.path.join(GOLDEN, "model_sets.json")))
PUBLISHED = json.load(open(os.path.join(GOLDEN, "published_notebook_errors.json")))
ETA = 50.0

CASES = {
    "expsin": (p1mc.MMSCaseExpSin, "nb_a", 0.01),
    "pol": (p1mc.MMSCasePol, "nb_b", 0.01),
    "peaks_fast1e1": (p1mc.MMSCaseSlowlyChangingPeaks_Fast1e1, "nb_b", 1.0),
    "nfs_h2_h3": (p1mc.MMSCaseNonFullySmoothPol_cpcsH2_TclcdH3, "nb_a", 1.0),
    "nfs_h1_h2": (p1mc.MMSCaseNonFullySmoothPol_cpcsH1_TclcdH2, "nb_a", 1.0),
    "nfs_h2_h2": (p1mc.MMSCaseNonFullySmoothPol_cpcsH2_TclcdH2, "nb_a", 1.0),
}
# cases stated symbolically (forcing evaluated on the host and streamed); sympy is optional
if p1mc.sympy is not None:
    CASES.update({
        "cs_zero_crossing": (p1mc.MMSCaseCsZeroCrossing, "nb_b", 1.0),
        "stiff_exp_decay": (p1mc.MMSCaseStiffExpDecay, "nb_b", 1.0),
        "pol_oscillating": (p1mc.make_MMSCasePolWithOscilatingTime_cls(ampl=1.0, speed=10.0), "nb_b", 1.0),
    })
SYMBOLIC_CASES = ["cs_zero_crossing", "stiff_exp_decay", "pol_oscillating"]


def model_of(name, kind=2):
    mc = p1.ModelConsts(**MODEL_SETS[name])
    return p1.DefaultModel02(mc) if kind == 2 else p1.DefaultModel01(mc)


def reg_trial(case_cls, model, N, eta=ETA, integrator_cls=None, integrator_params=None, fused=True, M=None):
    grid = p1.make_uniform_grid(N=N, M=N if M is None else M)
    common = {"regularization_factor": eta}
    icls = integrator_cls or p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple
    iparams = dict(common) if integrator_params is None else integrator_params
    if icls is p1.ForwardEulerIntegrator:
        iparams = {}
    return mtu.MMSTrial(grid=grid, model=model, mms_case_cls=case_cls,
                        field_cls=p1.SemiDiscreteField_RegHCsTriple,
                        forcing_terms_cls=p1.ForcingTerms_RegHCsTriple, integrator_cls=icls,
                        integrator_params=iparams, forcing_terms_params=common, field_params=common,
                        fused=fused)


# ------------------------------------------------------------------------------------------
def check_classes_instantiation():
    """reference tests/test_reghcstriple_system.py:120-167"""
    model = model_of("nb_a")
    grid = p1.make_uniform_grid(N=5, M=5)
    case = p1mc.MMSCaseExpSin(grid=p1.make_uniform_grid(N=2, M=2), model=model)
    forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
    assert forcing.regularization_factor == ETA
    field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing,
                                              regularization_factor=ETA)
    assert field.regularization_factor == ETA
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(
        semi_discrete_field=field, regularization_factor=ETA, num_pc_steps=1, num_newton_iterations=3)
    assert integ._regularization_factor == ETA
    assert integ._num_newton_iterations == 3
    assert integ.num_pc_steps == 1 and integ.num_newton_steps == 1
    assert integ.semi_discrete_field is field and integ._model is model and integ._grid is grid


def check_published_errors(case_name, levels, rel):
    """The error norms printed in the reference's executed notebooks (spatial studies)."""
    cls, mset, Tf = CASES[case_name]
    model = model_of(mset)
    out = []
    for lev in PUBLISHED[case_name]["levels"]:
        N = lev["N"]
        if N not in levels:
            continue
        trial = reg_trial(cls, model, N)
        summ = trial.run_for_errors(Tf=Tf, dt=(1.0 / N) ** 1.5)
        # (pytest.approx keeps its default absolute tolerance of 1e-12 unless abs is given: give it always.
        #  An error norm is a norm of differences of O(1) fields, so it carries their rounding, ~1e-17 absolute)
        assert summ.overall_combined_error == pytest.approx(lev["overall"], rel=rel, abs=1e-17), (case_name, N)
        for v, val in lev["per_variable"].items():
            assert summ.per_variable_sup_errors[v] == pytest.approx(val, rel=max(rel, 5e-9), abs=2e-17), (case_name, N, v)
        out.append(summ.overall_combined_error)
    # observed 3-point rates of this sweep against the ones the notebook prints (3 decimals)
    all_levels = [lev["N"] for lev in PUBLISHED[case_name]["levels"]]
    if len(out) >= 3 and list(levels)[-len(out):] == all_levels[-len(out):] and len(out) == len(all_levels):
        from tedd.cvg_studies_base import calculate_observed_rates
        rates = [r for r, _status in calculate_observed_rates(out, 2.0)]
        printed = PUBLISHED[case_name]["rates_3pt"]
        for mine, ref in zip(rates[-len(printed):], printed):
            assert mine == pytest.approx(ref, abs=2e-3), (case_name, rates, printed)
    return out


def check_published_temporal_study(case_name, rel, max_n=None):
    """The temporal studies printed in the reference's executed notebooks: fixed mesh (N = 256, or 32 for
    ExpSin) and dt halved from 1e-2 to Tf = 0.01; the Peaks notebook instead refines the mesh with the step
    (N = ceil(1 / (2 dt)), Tf = 10: from a 1 x 1 mesh without interior nodes, error 0, through O(1) errors at
    dt = 1/4 to the asymptotic range).  Errors to the printed digits, and the printed 3-point rates."""
    cls, mset, _Tf = CASES[case_name]
    study = PUBLISHED[case_name]["temporal"]
    model = model_of(mset)
    errors = []
    for lev in study["levels"]:
        N = lev.get("N", study["N"])
        if max_n is not None and N > max_n:
            break
        summ = reg_trial(cls, model, N).run_for_errors(Tf=study["Tf"], dt=lev["dt"])
        assert summ.overall_combined_error == pytest.approx(lev["overall"], rel=rel, abs=1e-17), (case_name, lev)
        errors.append(summ.overall_combined_error)
    if len(errors) == len(study["levels"]) and study["rates_3pt"]:
        from tedd.cvg_studies_base import calculate_observed_rates
        rates = [r for r, _status in calculate_observed_rates(errors, 2.0) if np.isfinite(r)]
        assert rates == pytest.approx(study["rates_3pt"], abs=2e-3), (case_name, rates)
    return errors


def check_tiny_meshes():
    """Meshes down to a single cell (1 x 1: no interior node; the Peaks notebook's temporal study starts there):
    a step returns the boundary values of the exact solution and, where there are interior nodes, stays close
    to it."""
    model = model_of("nb_b")
    for N, M in ((1, 1), (1, 3), (3, 1), (2, 2), (1, 2)):
        grid = p1.make_uniform_grid(N, M)
        for name in ("peaks_fast1e1", "expsin"):
            trial = reg_trial(CASES[name][0], model, N, M=M)
            case = trial.mms_case
            S = trial.integrator.step(p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid), t0=0.0, dt=0.01)
            E = p1.state_from_mms_when(mms_case=case, t=0.01, grid=grid)
            for v in ("cp", "T", "cl", "cd", "cs"):
                got, want = getattr(S, v), getattr(E, v)
                assert got.shape == (N + 1, M + 1) and np.all(np.isfinite(got))
                bd = grid.null_bd_mask == 0
                if v in ("T", "cl", "cd"):
                    np.testing.assert_allclose(got[bd], want[bd], rtol=0, atol=1e-15, err_msg=f"{N}x{M} {name} {v}")
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-2, err_msg=f"{N}x{M} {name} {v}")
            summ = trial.run_for_errors(Tf=0.02, dt=0.01)
            assert np.isfinite(summ.overall_combined_error)
            if N == 1 or M == 1:
                assert summ.per_variable_sup_errors["T"] < 1e-15


SCALARS = json.load(open(os.path.join(GOLDEN, "golden_scalars.json")))


def check_reference_run_errors(case_name, max_n, rel):
    """MMSTrial.run_for_errors summaries of the reference itself (tests/golden/make_golden.py --scalars),
    all 16 digits: the same spatial-study schedule as the notebooks."""
    cls, mset, Tf = CASES[case_name]
    ref = SCALARS[case_name]
    assert ref["Tf"] == Tf and ref["model_set"] == mset
    model = model_of(mset)
    for lev in ref["levels"]:
        N = lev["N"]
        if N > max_n:
            continue
        summ = reg_trial(cls, model, N).run_for_errors(Tf=Tf, dt=(1.0 / N) ** 1.5)
        assert summ.dt_used == pytest.approx(lev["dt_used"], rel=1e-15), (case_name, N)
        assert summ.overall_combined_error == pytest.approx(lev["overall"], rel=rel, abs=1e-17), (case_name, N)
        for v, val in lev["per_variable"].items():
            # errors far below the O(1) fields carry the rounding of the fields themselves (~1e-16 absolute)
            assert summ.per_variable_sup_errors[v] == pytest.approx(val, rel=10 * rel, abs=1e-16), (case_name, N, v)


def check_reference_temporal_study(rel):
    """The notebooks' temporal study for ExpSin (N=32, Tf=0.01, dt = 1e-2 / 2^i) run with the reference."""
    cls, mset, _ = CASES["expsin"]
    model = model_of(mset)
    for row in SCALARS["expsin_temporal_N32"]:
        summ = reg_trial(cls, model, 32).run_for_errors(Tf=0.01, dt=row["dt"])
        assert summ.overall_combined_error == pytest.approx(row["overall"], rel=rel, abs=1e-17), row["dt"]


ISOLATED = json.load(open(os.path.join(GOLDEN, "golden_isolated.json")))


def _final_rate_at_least(errors, order=2.0, tol=0.1):
    from tedd.cvg_studies_base import calculate_observed_rates
    rate, status = calculate_observed_rates(errors, refinement_factor=2.0)[-1]
    assert rate >= order - tol, (errors, rate, status)


def check_isolated_operator_truncation(which):
    """reference tests/test_spatial_isolated_T_accuracy.py:88-548: the field operators FT, Fcl, Fcd applied
    to an exact state against the continuous right-hand side, with every unrelated model constant zero;
    second order, and the error norms equal to those of the reference's operators."""
    pytest.importorskip("sympy")
    import isolated_cases
    errors = isolated_cases.operator_truncation_errors(p1, which)
    assert errors == pytest.approx(ISOLATED["operators"][which], rel=1e-7, abs=0), which
    _final_rate_at_least(errors)


def check_coupled_subsystem_convergence(which):
    """reference tests/test_spatial_isolated_T_accuracy.py:551-741: T-cl and T-cl-cd subsystems (cp = cs = 0,
    several diffusivities and rate constants zero) through MMSTrial and the CsTriple predictor-corrector;
    second order in space and the reference's own error norms.  Some of these norms are 1e-14 ... 1e-17
    (one or two short steps from an exact state): the absolute tolerance is the rounding of the O(0.1)
    fields."""
    pytest.importorskip("sympy")
    import isolated_cases
    errors = isolated_cases.coupled_errors(p1, mtu, which)
    for v, ref in ISOLATED["coupled"][which].items():
        assert errors[v] == pytest.approx(ref, rel=1e-6, abs=5e-17), (which, v)
        # the reference asserts the last rate of its own numbers; do the same where the mirrored
        # numbers are the same numbers to the tolerance above
        _final_rate_at_least(ref)
        if min(ref) > 1e-13:
            _final_rate_at_least(errors[v])
    return errors


def check_foreign_mesh_evaluation():
    """A case / forcing-terms object built on one grid and asked for values at the nodes of other
    tensor-product meshes (the reference's objects are plain functions of the coordinates; its tests build the
    case once and refine the mesh: tests/test_reghcstriple.py:277-336)."""
    model = model_of("nb_a")
    home = p1.make_uniform_grid(N=6, M=5)
    case = p1mc.MMSCaseExpSin(grid=home, model=model)
    forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
    x = np.array([0.0, 0.1, 0.35, 0.5, 0.8, 1.0])
    for other in (p1.make_uniform_grid(N=9, M=7), p1.Grid(x, x[:4] / 0.5), p1.make_uniform_grid(N=3, M=4),
                  p1.make_uniform_grid(N=11, M=3), p1.make_uniform_grid(N=4, M=12), p1.make_uniform_grid(N=9, M=7)):
        there = p1mc.MMSCaseExpSin(grid=other, model=model)
        forcing_there = p1.ForcingTerms_RegHCsTriple(mms_case=there, model=model, regularization_factor=ETA)
        for name in ("cp", "T", "cl", "cd", "cs", "dt_cs", "dx_T", "dy_cl", "lap_cd"):
            np.testing.assert_array_equal(getattr(case, name)(0.3, other.xx, other.yy), getattr(there, name)(0.3, other.xx, other.yy))
        for name in ("fcp", "fT", "fcl", "fcd", "fcs"):
            np.testing.assert_array_equal(getattr(forcing, name)(0.3, other.xx, other.yy),
                                          getattr(forcing_there, name)(0.3, other.xx, other.yy))
        # and the object still answers on its own grid
        assert case.T(0.3, home.xx, home.yy).shape == home.xx.shape
        assert forcing.fT(0.3, home.xx, home.yy).shape == home.xx.shape
    # a field on another grid than its forcing terms' case steps like one whose case shares its grid
    grid = p1.make_uniform_grid(N=8, M=8)
    steps = []
    for case_grid in (home, grid):
        c = p1mc.MMSCaseExpSin(grid=case_grid, model=model)
        f = p1.ForcingTerms_RegHCsTriple(mms_case=c, model=model, regularization_factor=ETA)
        field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=f, regularization_factor=ETA)
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(field, regularization_factor=ETA)
        S = integ.step(p1.state_from_mms_when(mms_case=c, t=0.0, grid=grid), t0=0.0, dt=1e-3)
        steps.append(S)
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_array_equal(getattr(steps[0], v), getattr(steps[1], v))
    # scattered points, a line of points, a single point: evaluated through the mesh of their coordinates
    rng = np.random.default_rng(0)
    big = p1mc.MMSCaseExpSin(grid=p1.make_uniform_grid(N=9, M=7), model=model)
    gx, gy = big.grid.xx, big.grid.yy
    pick = (np.array([3, 0, 9, 5, 5]), np.array([2, 7, 0, 4, 2]))
    for name in ("cp", "T", "dt_cs", "dx_T", "lap_cd"):
        on_mesh = getattr(big, name)(0.3, gx, gy)
        np.testing.assert_array_equal(getattr(case, name)(0.3, gx[pick], gy[pick]), on_mesh[pick])
        single = getattr(case, name)(0.3, gx[3, 2], gy[3, 2])
        assert np.shape(single) == () and float(single) == on_mesh[3, 2]
        line = getattr(case, name)(0.3, gx[:, 4], gy[:, 4])
        np.testing.assert_array_equal(line, on_mesh[:, 4])
    px, py = rng.random((4, 5)), rng.random((4, 5))
    vals = case.T(0.3, px, py)
    assert vals.shape == (4, 5) and np.all(np.isfinite(vals))
    one = p1mc.MMSCaseExpSin(grid=p1.Grid(np.sort(np.append(px[0], [0, 1])), np.sort(np.append(py[0], [0, 1]))), model=model)
    np.testing.assert_allclose(vals[0], [float(one.T(0.3, a, b)) for a, b in zip(px[0], py[0])], rtol=1e-15)
    # a host-evaluated case: cell averages follow the mesh handed in
    if p1mc.sympy is not None:
        t, x, y = p1.t_sym, p1.x_sym, p1.y_sym
        ex = dict(cp_sym_expr=1 + x * y * t, T_sym_expr=p1mc.sympy.sin(x + t) * y, cl_sym_expr=x ** 2 + y,
                  cd_sym_expr=1 + t * y, cs_sym_expr=1 - t + x, device_program=False)
        sym = p1.MMSCaseSymbolic(grid=home, model=model, **ex)
        sym_there = p1.MMSCaseSymbolic(grid=grid, model=model, **ex)
        a = p1.ForcingTerms_CsTriple(mms_case=sym, model=model).fcp(0.2, grid.xx, grid.yy)
        b = p1.ForcingTerms_CsTriple(mms_case=sym_there, model=model).fcp(0.2, grid.xx, grid.yy)
        np.testing.assert_array_equal(a, b)


def check_interaction_law_accessors(kind):
    """reference tests/test_semidiscrete_field_hcs_triple.py:118-542 (and the same law for the other two
    families): the accessors of the Cs/Cd interaction against their closed form, and against what the device
    operators do with it: Fcs = -[Cs-Cd-int] and, on constant fields (no diffusion), Fcd = +[Cs-Cd-int]."""
    model = model_of("stress")
    grid = p1.make_uniform_grid(N=4, M=5)
    common = {"regularization_factor": 7.0} if kind == "reg" else {}
    field_cls = {"reg": p1.SemiDiscreteField_RegHCsTriple, "cs3": p1.SemiDiscreteField_CsTriple,
                 "hcs": p1.SemiDiscreteField_HCsTriple}[kind]
    field = field_cls(grid=grid, model=model, forcing_terms=p1.NoForcingTerms(grid=grid), **common)
    gain = {"reg": lambda c: p1.heaviside_regularized(c, 7.0), "cs3": lambda c: c, "hcs": lambda c: float(c > 0)}[kind]
    assert field.cscd_reaction_T() == (0, 1) and field.cscd_reaction_cl() == (1, 1)
    assert field.cscd_reaction_cd() == (-1, model.Sd)
    np.testing.assert_array_equal(field.cscd_reaction_cp(np.zeros(grid.full_shape)), grid.null_bd_mask)
    cp, T, cl, cd = 0.1, 0.2, 0.3, 0.4
    for cs in (2.0, 1e-10, 0.0, -1e-10, -2.0):
        full = lambda v: np.full(grid.full_shape, v, dtype=float)
        np.testing.assert_allclose(field.cscd_reaction_cs(full(cs)), full(model.Kd * gain(cs)), rtol=1e-14)
        S = p1.StateVars(full(cp), full(T), full(cl), full(cd), full(cs), model=model, hh=grid.hh, kk=grid.kk)
        expected = model.Kd * gain(cs) * (1 + cl) * (model.Sd - cd) * grid.null_bd_mask
        term = field.cscd_reaction_term(S)
        np.testing.assert_allclose(term, expected, rtol=1e-13, atol=0)
        np.testing.assert_allclose(field.Fcs(S, 0.0), -expected, rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(field.Fcd(S, 0.0), expected, rtol=1e-12, atol=1e-22)
        np.testing.assert_array_equal(field.delT_ab_cscd_reaction_ij(S, a=0, b=0), 0.0)     # a_T = 0
        np.testing.assert_allclose(field.delcl_ab_cscd_reaction_ij(S, a=0, b=0),
                                   model.Kd * gain(cs) * (model.Sd - cd) * grid.null_bd_mask, rtol=1e-13)
        np.testing.assert_allclose(field.delcd_ab_cscd_reaction_ij(S, a=0, b=0),
                                   -model.Kd * gain(cs) * (1 + cl) * grid.null_bd_mask, rtol=1e-13)
        # the reaction enters the diagonal of d Fcd / d cd (constant fields: the diffusion part is the
        # discrete Laplacian's diagonal, the same for every value of cs)
        diag = field.delcd_ab_Fcd_ij(S, a=0, b=0)
        S0 = S.with_changes(cs=full(-2.0 if kind != "cs3" else 0.0))     # no reaction
        np.testing.assert_allclose((diag - field.delcd_ab_Fcd_ij(S0, a=0, b=0))[1:-1, 1:-1],
                                   field.delcd_ab_cscd_reaction_ij(S, a=0, b=0)[1:-1, 1:-1]
                                   - field.delcd_ab_cscd_reaction_ij(S0, a=0, b=0)[1:-1, 1:-1], rtol=1e-9,
                                   atol=4e-16 * np.max(np.abs(diag)))     # a difference of two diagonals
        for a, b in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            for fn in (field.delT_ab_cscd_reaction_ij, field.delcl_ab_cscd_reaction_ij, field.delcd_ab_cscd_reaction_ij):
                np.testing.assert_array_equal(fn(S, a=a, b=b), 0.0)
        with pytest.raises(AssertionError):
            field.delcl_ab_cscd_reaction_ij(S, a=1, b=1)


def check_notebook_helpers(tmp_path):
    """reference src/notebook_studies_helpers.py: the rate report that returns instead of raising, the two
    figures (drawn off-screen), the template-cell replacement."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import tedd.notebook_studies_helpers as nsh
    errors = [1.0, 0.26, 0.0675, 0.0171]
    rates, ok = nsh.report_on_rates(errors, expected_rate=2.0, tolerance=0.1, title="Spatial")
    assert ok and len(rates) == 2 and rates[-1] == pytest.approx(1.93, abs=0.03)
    assert nsh.report_on_rates([1.0, 0.6, 0.35, 0.2], expected_rate=2.0) == ([], False)
    assert nsh.report_on_rates([1.0, 0.5, 0.25, 0.125], expected_rate=1.0, cmp_type="equal")[1]
    nsh.plot_errors_and_rates(errors, rates, [8, 16, 32, 64], "N", "study", expected_rate=2.0)
    nsh.plot_errors_and_rates(errors[:2], [], [8, 16], "N", "too short", log_scale=False)
    model = model_of("nb_a")
    nsh.visualize_mms_solution(p1mc.MMSCaseExpSin(grid=p1.make_uniform_grid(6, 6), model=model), time_point=0.1)
    assert len(plt.get_fignums()) == 3
    assert len(plt.figure(plt.get_fignums()[-1]).axes) == 11      # 6 panels + 5 colour bars
    plt.close("all")
    template = {"cells": [{"cell_type": "markdown", "metadata": {}, "source": ["# title"]},
                          {"cell_type": "code", "metadata": {}, "outputs": [], "execution_count": None,
                           "source": ["mms_case_cls = None\n"]}],
                "metadata": {"kernelspec": {"name": "python3"}}, "nbformat": 4, "nbformat_minor": 5}
    src, dst = tmp_path / "template.ipynb", tmp_path / "made.ipynb"
    src.write_text(json.dumps(template), encoding="utf-8")
    nsh.modify_notebook_cell(str(src), str(dst), 1, "import prob1_mms_cases as p1mc\nmms_case_cls = p1mc.MMSCasePol\n")
    made = json.loads(dst.read_text(encoding="utf-8"))
    assert "".join(made["cells"][1]["source"]).endswith("p1mc.MMSCasePol\n")
    assert made["cells"][0] == template["cells"][0] and made["metadata"] == template["metadata"]
    with pytest.raises(AssertionError):
        nsh.modify_notebook_cell(str(src), str(dst), 2, "x")


REGHCS = json.load(open(os.path.join(GOLDEN, "golden_reghcs.json")))


def check_reghcs_scenarios(eta):
    """reference tests/test_reghcstriple.py:109-577 for one regularisation factor, with the reference's own
    numbers (tests/golden/make_golden.py --only reghcs) beside its assertions."""
    import reghcs_cases as rc
    ref = REGHCS[str(eta)]
    close = lambda a, b, rel=1e-9, abs_=1e-15: a == pytest.approx(b, rel=rel, abs=abs_)
    # the regularised Heaviside function: bounded, monotone limits, steeper than a factor of 5
    h = np.array(rc.heaviside_values(p1, eta))
    assert close(list(h), ref["heaviside"], rel=1e-14)
    assert np.all(h >= 0) and np.all(h <= 1) and h[0] < 0.01 and h[-1] > 0.99 and 0.45 < h[50] < 0.55
    assert h[51] > p1.heaviside_regularized(np.linspace(-5, 5, 101), 5.0)[51]
    # the field applied to the exact solution approximates the time derivative
    for N in rc.SIZES:
        mine = rc.field_vs_time_derivative(p1, p1mc, N, eta)
        for v, val in ref["field_vs_dt"][str(N)].items():
            assert close(mine[v], val, rel=1e-8), (N, v)
            assert N < 32 or mine[v] < 1e-2
    errs = rc.FT_refinement_errors(p1, p1mc, eta)
    assert close(errs, ref["FT_refinement"], rel=1e-8)
    observed_rates_report(errs, expected_rate=2.0, tolerance=0.3, cmp_type="least", halt_print=True)
    # one step of each integrator from the exact state
    for key, val in ref["fe_one_step"].items():
        N, dt = key.split("/")
        mine = rc.one_step_errors(p1, p1mc, int(N), eta, "fe", float(dt))
        assert all(close(mine[v], val[v], rel=1e-8) for v in val), key
    for N, val in ref["pc_one_step"].items():
        mine = rc.one_step_errors(p1, p1mc, int(N), eta, "pc", 1e-3)
        assert all(close(mine[v], val[v], rel=1e-6) for v in val), (N, mine, val)
        assert int(N) < 32 or max(mine.values()) < 1e-2
    errs = rc.forward_euler_temporal_errors(p1, p1mc, eta)
    assert close(errs, ref["fe_temporal"], rel=1e-8)
    observed_rates_report(errs, expected_rate=1.0, tolerance=0.15, cmp_type="equal", halt_print=True)
    mine = rc.short_trials(p1, p1mc, mtu, eta)
    for k in ("fe", "pc"):
        assert close(mine[k], ref["short_trials"][k], rel=1e-8), k
        assert mine[k] < 1e-1


def check_gradient_T_one_step():
    """reference tests/test_spatial_h1_isolated_T_accuracy.py:57-173: error of the discrete gradient of T after
    one step, forcing given as a dictionary of callables (fT of the forcing-terms class, zeros otherwise:
    the field then streams host-evaluated arrays); the reference's numbers, and its order assertion."""
    pytest.importorskip("sympy")
    import isolated_cases
    errors = isolated_cases.gradient_T_one_step_errors(p1)
    assert errors == pytest.approx(ISOLATED["gradient_T_one_step"], rel=1e-6, abs=1e-16)
    _final_rate_at_least(errors)


def check_forward_euler_T_cl():
    """reference tests/test_feuler_spatial_accuracy.py:57-186: forward Euler on the coupled T-cl subsystem
    (DefaultModel01, every constant of the other equations zero)."""
    pytest.importorskip("sympy")
    import isolated_cases
    from tedd.cvg_studies_base import RateStatus, calculate_observed_rates
    errors = isolated_cases.forward_euler_T_cl_errors(p1)
    for v, ref in ISOLATED["forward_euler_T_cl"].items():
        assert errors[v] == pytest.approx(ref, rel=1e-6, abs=1e-16), v
        rate, status = calculate_observed_rates(errors[v], refinement_factor=2.0)[-1]
        assert status == RateStatus.OK and rate == pytest.approx(2.0, abs=0.2), (v, rate, status)


def check_isolated_correctors_temporal_order():
    """reference tests/test_isolated_correctors_cp_cs.py:53-142: the cp and cs correctors alone, fed the exact
    T, cl, cd of the new time level (DefaultModel01, CsTriple, 30 x 30, dt = 2 / (10 * 2^k)); second order."""
    model = p1.DefaultModel01(mc=p1.default_model_consts)   # the constants that test sets are the defaults
    grid = p1.make_uniform_grid(30, 30)
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)
    forcing = p1.ForcingTerms_CsTriple(mms_case=case, model=model)
    field = p1.SemiDiscreteField_CsTriple(grid=grid, model=model, forcing_terms=forcing)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple(semi_discrete_field=field, num_pc_steps=1,
                                                                     num_newton_steps=1)
    xx, yy = grid.xx, grid.yy
    errors = {"cp": [], "cs": []}
    for k in range(5):
        dt = 2.0 / (10 * 2 ** k)
        S0 = p1.StateVars(case.cp(0.0, xx, yy), case.T(0.0, xx, yy), case.cl(0.0, xx, yy), case.cd(0.0, xx, yy),
                          case.cs(0.0, xx, yy), model=model, hh=grid.hh, kk=grid.kk)
        T1, cl1, cd1 = case.T(dt, xx, yy), case.cl(dt, xx, yy), case.cd(dt, xx, yy)
        cp1 = integ.corrector_cp_step(T1, cl1, _cd1_ignroed=None, at_t0=S0, t0=0.0, dt=dt)
        cs1 = integ.corrector_cs_step(_T1_ignored=None, cl1=cl1, cd1=cd1, at_t0=S0, t0=0.0, dt=dt)
        errors["cp"].append(grid.norm_H(cp1 - case.cp(dt, xx, yy)))
        errors["cs"].append(grid.norm_H(cs1 - case.cs(dt, xx, yy)))
    for v in ("cp", "cs"):
        observed_rates_report(errors=errors[v], cmp_type="least", expected_rate=2.0, tolerance=0.1)
    return errors


STUDIES = json.load(open(os.path.join(GOLDEN, "golden_studies.json")))


def check_run_convergence_studies(label, rel, fused=True):
    """The legacy study runner (reference src/cvg_studies_base.py:319-486) against its own report from
    the reference (tests/golden/make_golden.py --only studies): errors, 3-point rates and status strings."""
    import tedd.cvg_studies_base as cvg
    ref = STUDIES["studies"][label]
    family = ref["family"]
    cfg = (getattr(p1, f"SemiDiscreteField_{family}"), CASES[ref["case"]][0], getattr(p1, f"ForcingTerms_{family}"),
           getattr(p1, f"P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_{family}"), label)
    params = dict(STUDIES["params"], model=model_of(ref["model_set"]), fused=fused)
    rep = cvg.run_convergence_studies([cfg], params, verbose=False)
    assert list(rep) == [label]
    for study in ("spatial", "temporal"):
        mine, theirs = rep[label][study], ref["report"][study]
        assert set(mine) == {"errors", "rates", "statuses"}
        assert mine["errors"] == pytest.approx(theirs["errors"], rel=rel, abs=0), (label, study)
        assert mine["statuses"] == theirs["statuses"], (label, study)
        for a, b in zip(mine["rates"], theirs["rates"]):
            if b is None:
                assert np.isnan(a)
            else:
                # a rate is a log of a ratio of error differences: it amplifies rel by E/(E_k - E_k+1)
                assert a == pytest.approx(b, abs=1e-6), (label, study, mine["rates"], theirs["rates"])
    return rep


def check_fused_equals_stepwise(case_name, N, nsteps_Tf):
    """The device-fused time loop + norms and the reference's step-by-step loop agree."""
    cls, mset, _ = CASES[case_name]
    model = model_of(mset)
    dt = (1.0 / N) ** 1.5
    a = reg_trial(cls, model, N, fused=True).run_for_errors(Tf=nsteps_Tf * dt, dt=dt)
    b = reg_trial(cls, model, N, fused=False).run_for_errors(Tf=nsteps_Tf * dt, dt=dt)
    assert a.dt_used == b.dt_used
    assert a.overall_combined_error == pytest.approx(b.overall_combined_error, rel=1e-9, abs=0)
    for v in a.per_variable_sup_errors:
        assert a.per_variable_sup_errors[v] == pytest.approx(b.per_variable_sup_errors[v], rel=1e-7, abs=1e-18)


def check_streamed_equals_closed_form(N=8):
    """Forcing streamed from host callables gives the same step as forcing evaluated in-kernel."""
    model = model_of("nb_a")
    grid = p1.make_uniform_grid(N, N)
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)
    forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
    f_closed = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing,
                                                 regularization_factor=ETA)
    f_stream = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model,
                                                 forcing_terms=p1.ForcingTermsFromDict(forcing.asdict()),
                                                 regularization_factor=ETA)
    assert f_closed._closed_form_source() is forcing and f_stream._closed_form_source() is None
    S0 = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    dt = (1.0 / N) ** 1.5
    ia = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(f_closed, regularization_factor=ETA)
    ib = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(f_stream, regularization_factor=ETA)
    Sa, Sb, t = S0, S0, 0.0
    for _ in range(3):
        Sa = ia.step(Sa, t0=t, dt=dt)
        Sb = ib.step(Sb, t0=t, dt=dt)
        t += dt
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_allclose(getattr(Sa, v), getattr(Sb, v), rtol=0, atol=1e-15)
    for v in ("T", "cl", "cd"):
        np.testing.assert_allclose(ia.last_residual[v], ib.last_residual[v], rtol=0, atol=1e-15)
    for a, b in zip((f_closed.Fcp, f_closed.FT, f_closed.Fcl, f_closed.Fcd, f_closed.Fcs),
                    (f_stream.Fcp, f_stream.FT, f_stream.Fcl, f_stream.Fcd, f_stream.Fcs)):
        np.testing.assert_allclose(a(S0, 0.1), b(S0, 0.1), rtol=0, atol=1e-15)


def check_field_correctness(sizes):
    """||F(u_exact) - d/dt u_exact||_H decreases with refinement (reference
    tests/test_reghcstriple_system.py:175-259)."""
    model = model_of("nb_a")
    errs = []
    for n in sizes:
        grid = p1.make_uniform_grid(N=n, M=n)
        case = p1mc.MMSCaseExpSin(grid=grid, model=model)
        forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
        field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing,
                                                  regularization_factor=ETA)
        t = 0.1
        S = p1.state_from_mms_when(mms_case=case, t=t, grid=grid)
        F = {"cp": field.Fcp(S, t), "T": field.FT(S, t), "cl": field.Fcl(S, t), "cd": field.Fcd(S, t),
             "cs": field.Fcs(S, t)}
        dte = {v: getattr(case, "dt_" + v)(t, grid.xx, grid.yy) for v in F}
        errs.append(np.sqrt(sum(grid.norm_H(F[v] - dte[v]) ** 2 for v in F)))
    assert errs[-1] < errs[0]
    if len(errs) >= 3:
        observed_rates_report(errors=errs, expected_rate=2.0, tolerance=0.5, cmp_type="least")
    return errs


def check_spatial_convergence(sizes):
    """reference tests/test_reghcstriple_system.py:279-342"""
    model = model_of("nb_a")
    dt = 5e-6
    params = {"regularization_factor": ETA, "num_pc_steps": 1, "num_newton_iterations": 5, "consec_xs_rtol": 1e-7}
    errs = [reg_trial(p1mc.MMSCaseExpSin, model, n, integrator_params=params).run_for_errors(Tf=dt, dt=dt)
            .overall_combined_error for n in sizes]
    observed_rates_report(errors=errs, expected_rate=2.0, tolerance=0.3, cmp_type="least")
    return errs


def check_temporal_convergence(N, dts):
    """reference tests/test_reghcstriple_system.py:345-416 (num_newton_iterations=1000, rtol 1e-9)"""
    model = model_of("nb_a")
    params = {"regularization_factor": ETA, "num_pc_steps": 1, "num_newton_iterations": 1000,
              "consec_xs_rtol": 1e-9}
    errs = [reg_trial(p1mc.MMSCaseExpSin, model, N, integrator_params=params).run_for_errors(Tf=dt, dt=dt)
            .overall_combined_error for dt in dts]
    observed_rates_report(errors=errs, expected_rate=2.0, tolerance=0.3, cmp_type="least")
    return errs


def check_pc_beats_forward_euler():
    """reference tests/test_reghcstriple_system.py:419-483"""
    model = model_of("nb_a")
    dt = 5e-4
    params = {"regularization_factor": ETA, "num_pc_steps": 1, "num_newton_iterations": 1, "consec_xs_rtol": 1e-7}
    e_pc = reg_trial(p1mc.MMSCaseExpSin, model, 4, integrator_params=params).run_for_errors(Tf=3 * dt, dt=dt)
    e_fe = reg_trial(p1mc.MMSCaseExpSin, model, 4, integrator_cls=p1.ForwardEulerIntegrator).run_for_errors(Tf=3 * dt, dt=dt)
    assert e_pc.overall_combined_error < e_fe.overall_combined_error


def check_forward_euler_order(grid_n, t_test):
    """reference tests/test_time_integration_fwd_euler_full_p1base.py:123-198 (CsTriple, ExpSin)"""
    model = model_of("fe")
    grid = p1.make_uniform_grid(N=grid_n, M=grid_n)
    trial = mtu.MMSTrial(grid=grid, model=model, mms_case_cls=p1mc.MMSCaseExpSin,
                         field_cls=p1.SemiDiscreteField_CsTriple, forcing_terms_cls=p1.ForcingTerms_CsTriple,
                         integrator_cls=p1.ForwardEulerIntegrator)
    max_D = max(model.DT, model.Dl_max, model.Dd_max, 1e-9)
    base_dt = min(4e-4, 0.5 * (1.0 / grid_n) ** 2 / (4.0 * max_D))
    # one-step consistency (…:77-120)
    dt_test = min(1e-2, (1.0 / grid_n) ** 2 / (4.0 * max_D) * 0.5)
    summ = trial.run_for_errors(Tf=t_test + dt_test, dt=dt_test, t0=t_test)
    for v in ("cp", "T", "cl", "cd", "cs"):
        assert summ.per_variable_sup_errors[v] == pytest.approx(0.0, abs=100 * dt_test, rel=1e-3)
    return base_dt


def check_newton_residuals(case_name, N=8, dt=1e-4, steps=(1, 20)):
    """||last_residual||_H < 1e-9 after 1 and 20 steps with 5 Newton steps
    (reference tests/test_newton_residuals.py:116-124, run there for CsTriple/HCsTriple)."""
    cls, mset, _ = CASES[case_name]
    model = model_of(mset)
    grid = p1.make_uniform_grid(N, N)
    case = cls(grid=grid, model=model)
    forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
    field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing,
                                              regularization_factor=ETA)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(
        field, regularization_factor=ETA, num_newton_steps=5)
    S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    t = 0.0
    for k in range(1, max(steps) + 1):
        S = integ.step(S, t0=t, dt=dt)
        t += dt
        if k in steps:
            for v in ("T", "cl", "cd"):
                assert grid.norm_H(integ.last_residual[v]) < 1e-9, (case_name, k, v)


def check_hcs_newton_residuals(case_name, steps=(1, 20)):
    """reference tests/test_newton_residuals.py:127-198: ||last_residual||_H < 1e-9 after 1 and 20 steps
    of the HCsTriple integrator with 5 Newton steps (N=8, dt=1e-4, its model constants)."""
    cls = CASES[case_name][0]
    mc = p1.default_model_consts._replace(K1=1e-2, K2=1e-2, K3=1e-2, K4=1e-2, DT=1e-3, Kd=1e-3, Sd=10)
    model = p1.DefaultModel02(mc=mc)
    grid = p1.make_uniform_grid(8, 8)
    case = cls(grid=grid, model=model)
    field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model,
                                           forcing_terms=p1.ForcingTerms_HCsTriple(mms_case=case, model=model))
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(field, num_pc_steps=1, num_newton_steps=5)
    S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    t, dt = 0.0, 1e-4
    for k in range(1, max(steps) + 1):
        S = integ.step(S, t0=t, dt=dt)
        t += dt
        if k in steps:
            for v in ("T", "cl", "cd"):
                assert grid.norm_H(integ.last_residual[v]) < 1e-9, (case_name, k, v)


def check_hcs_cs_ramp_single_step(cs_initial, cs_slope, dt=0.01):
    """reference tests/test_time_integrator_hcs_triple_full_step.py:133-283: cp = T = cl = cd = 0 and
    cs = (cs_initial + cs_slope t) x(1-x) y(1-y); one step reproduces the exact solution (the ramp is
    linear in t, the corrector is exact for it) whatever the sign of cs before and after the step.
    The manufactured forcing is supplied as host arrays (streamed to the device)."""
    mc = p1.ModelConsts(R0=8.3144621, Ea=1.60217662e-19, K1=1.52e-2, K2=1.52e-2, K3=1.52e-2, K4=1.52e-2,
                        DT=1.22e-3, Dl_max=8.22e-4, phi_l=1.22e-5, gamma_T=1.22e-9, Kd=1.22e-8, Sd=15,
                        Dd_max=2.62e-6, phi_d=1.22e-5, phi_T=1.60217662e-19 / 8.3144621, r_sp=5.22e-2, T_ref=303)
    model = p1.DefaultModel01(mc=mc)
    grid = p1.make_uniform_grid(N=4, M=4)
    P = grid.xx * (1 - grid.xx) * grid.yy * (1 - grid.yy)
    zero = grid.make_full0()

    def cs_exact(t):
        return (cs_initial + cs_slope * t) * P

    forcing = p1.ForcingTermsFromDict({
        "fcp": lambda t, xx, yy: zero.copy(), "fT": lambda t, xx, yy: zero.copy(),
        "fcl": lambda t, xx, yy: zero.copy(),
        "fcd": lambda t, xx, yy: -mc.Kd * mc.Sd * (cs_exact(t) > 0),
        "fcs": lambda t, xx, yy: cs_slope * P + mc.Kd * mc.Sd * (cs_exact(t) > 0)})
    field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model, forcing_terms=forcing)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(field, num_pc_steps=1, num_newton_steps=1)
    S0 = p1.StateVars(zero.copy(), zero.copy(), zero.copy(), zero.copy(), cs_exact(0.0), model=model,
                      hh=grid.hh, kk=grid.kk)
    S1 = integ.step(S0, t0=0.0, dt=dt)
    for v in ("cp", "T", "cl", "cd"):
        np.testing.assert_allclose(getattr(S1, v), zero, rtol=1e-8, atol=1e-12, err_msg=v)
    np.testing.assert_allclose(S1.cs, cs_exact(dt), rtol=1e-6, atol=1e-12)


def check_hcs_denominator_value_error():
    """reference src/prob1base.py:3406-3409: step() raises ValueError when 2 - dt Kd (Sd - cd1)(1 + cl1)
    is not positive."""
    mc = p1.default_model_consts._replace(Kd=1e4)
    model = p1.DefaultModel02(mc=mc)
    grid = p1.make_uniform_grid(5, 5)
    case = p1mc.MMSCasePol(grid=grid, model=model)
    field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model,
                                           forcing_terms=p1.ForcingTerms_HCsTriple(mms_case=case, model=model))
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(field, jacobi_sweeps=(50, 50, 50))
    with pytest.raises(ValueError):
        integ.step(p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid), t0=0.0, dt=0.1)


def _hcs_integrator_with_cs_source(grid, model, at_t0_value, later_value):
    """HCsTriple integrator whose only forcing is fcs(t) = at_t0_value at t = 0 and later_value afterwards."""
    nothing = lambda t, xx, yy: grid.make_full0()
    fcs = lambda t, xx, yy: np.full(grid.full_shape, at_t0_value if np.isclose(t, 0.0) else later_value, dtype=float)
    forcing = p1.ForcingTermsFromDict({"fcp": nothing, "fT": nothing, "fcl": nothing, "fcd": nothing, "fcs": fcs})
    field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model, forcing_terms=forcing)
    return p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(semi_discrete_field=field)


def _constant_state(grid, model, cs, cl=-1.0):
    """cl = -1 switches the Cs/Cd interaction off at this state: Fcs then is the source alone."""
    full = lambda v: np.full(grid.full_shape, v, dtype=float)
    return p1.StateVars(full(0.1), full(0.2), full(cl), full(0.4), full(cs), model=model, hh=grid.hh, kk=grid.kk)


def _zero_boundary(a):
    return np.all(a[0, :] == 0) and np.all(a[-1, :] == 0) and np.all(a[:, 0] == 0) and np.all(a[:, -1] == 0)


def check_hcs_cs_predictor_table():
    """The known answers of reference tests/test_time_integrator_hcs_triple.py:182-256: the Heun predictor
    cs0 + dt/2 (Fcs(t0) + Fcs*(t1)) is not clipped at zero, only the boundary is.  The reference prescribes
    Fcs by overriding the field's method; here the same values come from the source term, with the
    interaction switched off by cl = -1."""
    model = model_of("nb_a", kind=1)
    grid = p1.make_uniform_grid(4, 4)
    for cs0, F0, F1, dt in ((1.0, 0.1, 0.2, 0.1), (0.01, -0.2, -0.2, 0.1), (-0.1, -0.1, -0.1, 0.1), (0.0, 0.0, 0.0, 0.1)):
        integ = _hcs_integrator_with_cs_source(grid, model, F0, F1)
        cs1 = integ.initial_cs_pred(_constant_state(grid, model, cs0), 0.0, dt=dt)
        np.testing.assert_allclose(cs1[1:-1, 1:-1], cs0 + 0.5 * dt * (F0 + F1), rtol=1e-9, atol=1e-12)
        assert _zero_boundary(cs1)


def check_hcs_cs_corrector_table():
    """The known answers of reference tests/test_time_integrator_hcs_triple.py:258-401: with
    Y0 = 2 cs0 + dt (Fcs(t0) + fcs(t1)) and R1 = Kd (Sd - cd1)(1 + cl1) the corrector returns
    Y0 / (2 - dt R1) where Y0 > 0 and Y0 / 2 elsewhere; and :403-498: ValueError when 2 - dt R1 <= 0."""
    base = model_of("nb_a", kind=1)
    grid = p1.make_uniform_grid(4, 4)
    full = lambda v: np.full(grid.full_shape, v, dtype=float)
    table = ((0.1, 0.5, 0.5, 0.5, 0.1), (0.01, 0.02, 0.02, 0.1, 0.01), (-0.1, -0.5, -0.5, 0.5, 0.1),
             (-0.01, -0.02, -0.02, 10.0, 0.1), (0.0, 0.0, 0.0, 0.5, 0.1), (0.001, -0.02, 0.0, 1.0, 0.1),
             (0.0, 1e-12, -1e-12, 1.0, 0.1))
    for cs0, F0, f1, R1, dt in table:
        model = base.with_changes(Kd=R1)
        integ = _hcs_integrator_with_cs_source(grid, model, F0, f1)
        cs1 = integ.corrector_cs_step(grid.make_full0(), full(0.0), full(model.Sd - 1.0),
                                      at_t0=_constant_state(grid, model, cs0), t0=0.0, dt=dt)
        Y0 = 2 * cs0 + dt * (F0 + f1)
        expected = Y0 / (2.0 - dt * R1) if Y0 > 0 else Y0 / 2.0
        np.testing.assert_allclose(cs1[1:-1, 1:-1], expected, rtol=1e-9, atol=1e-12, err_msg=str((cs0, F0, f1, R1, dt)))
        assert _zero_boundary(cs1)
    for dt, R1 in ((0.1, 20.0), (0.1, 21.0), (0.01, 200.0), (0.01, 200.1)):
        model = base.with_changes(Kd=R1)
        integ = _hcs_integrator_with_cs_source(grid, model, 0.1, 0.1)
        with pytest.raises(ValueError, match="Denominator 2 - Δt Kd .* below positiveness treshold"):
            integ.corrector_cs_step(grid.make_full0(), full(0.0), full(model.Sd - 1.0),
                                    at_t0=_constant_state(grid, model, 0.1), t0=0.0, dt=dt)


def check_integrator_pieces(kind):
    """The methods step() is made of (reference src/prob1base.py:2946-3115), called through the mirrored
    integrator in step()'s order (:3117-3149), against the reference's results for the same calls and
    against the mirrored step() itself."""
    z = np.load(os.path.join(GOLDEN, "golden_substeps.npz"))
    tag = f"{kind}/expsin_u8"
    model = model_of("nb_a")
    grid = p1.Grid(z[f"{tag}/x"], z[f"{tag}/y"])
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)
    if kind == "reg":
        forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=ETA)
        field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing, regularization_factor=ETA)
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(field, regularization_factor=ETA)
    elif kind == "cs3":
        field = p1.SemiDiscreteField_CsTriple(grid=grid, model=model,
                                              forcing_terms=p1.ForcingTerms_CsTriple(mms_case=case, model=model))
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple(field)
    else:
        field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model,
                                               forcing_terms=p1.ForcingTerms_HCsTriple(mms_case=case, model=model))
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(field)
    t0, dt = float(z[f"{tag}/t0"]), float(z[f"{tag}/dt"])
    S0 = p1.state_from_mms_when(mms_case=case, t=t0, grid=grid)
    tol = dict(rtol=0, atol=5e-14)
    YT0 = dt * field.FT(S0, t0) + 2 * S0.T
    Ycl0 = dt * field.Fcl(S0, t0) + 2 * S0.cl
    Ycd0 = dt * field.Fcd(S0, t0) + 2 * S0.cd
    np.testing.assert_allclose(YT0, z[f"{tag}/YT"], **tol)
    cp1 = integ.initial_cp_pred(S0, t0, dt=dt)
    cs1 = integ.initial_cs_pred(S0, t0, dt=dt)
    np.testing.assert_allclose(cp1, z[f"{tag}/cp_pred"], **tol)
    np.testing.assert_allclose(cs1, z[f"{tag}/cs_pred"], **tol)
    lag = S0.with_changes(cp=cp1, T=S0.T, cl=S0.cl, cd=S0.cd, cs=cs1)
    T1 = integ.newton_step_T(lag, t0=t0, dt=dt, YT0=YT0)
    np.testing.assert_allclose(integ.last_residual["T"], z[f"{tag}/res_T"], rtol=0, atol=1e-13)
    cl1 = integ.newton_step_cl(lag, T1, t0=t0, dt=dt, Ycl0=Ycl0)
    cd1 = integ.newton_step_cd(lag, T1, cl1, t0=t0, dt=dt, Ycd0=Ycd0)
    for a, name in ((T1, "T1"), (cl1, "cl1"), (cd1, "cd1")):
        np.testing.assert_allclose(a, z[f"{tag}/{name}"], **tol)
    assert set(integ.last_residual) == {"T", "cl", "cd"}
    cpc = integ.corrector_cp_step(T1, cl1, cd1, at_t0=S0, t0=t0, dt=dt)
    csc = integ.corrector_cs_step(T1, cl1, cd1, at_t0=S0, t0=t0, dt=dt)
    np.testing.assert_allclose(cpc, z[f"{tag}/cp_corr"], **tol)
    np.testing.assert_allclose(csc, z[f"{tag}/cs_corr"], **tol)
    full = integ.step(S0, t0=t0, dt=dt)
    for a, v in ((cpc, "cp"), (T1, "T"), (cl1, "cl"), (cd1, "cd"), (csc, "cs")):
        np.testing.assert_allclose(a, getattr(full, v), rtol=0, atol=1e-15)
    with pytest.raises(AssertionError):
        integ.corrector_cp_step(T1, cl1, cd1, at_t0=S0, t0=t0, dt=0.0)


def check_symbolic_case_equals_builtin(kind, device_program=True):
    """An MMSCaseSymbolic with the expressions of MMSCasePol (u = x(1-x) y(1-y) / (1+t) for all five,
    reference src/prob1_mms_cases.py:151-186): its forcing terms -- evaluated on the device by the lowered
    expression programs, or on the host and streamed (device_program=False) -- and the steps taken with
    them agree with the library's closed-form Pol case; so do the device-resident run with error norms
    and the exact solution."""
    sympy = pytest.importorskip("sympy")
    model = model_of("nb_b")
    grid = p1.Grid(np.linspace(0, 1, 9) ** 1.2, np.linspace(0, 1, 7))
    u = p1.x_sym * (1 - p1.x_sym) * p1.y_sym * (1 - p1.y_sym) / (1 + p1.t_sym)
    sym = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=u, T_sym_expr=u, cl_sym_expr=u, cd_sym_expr=u,
                             cs_sym_expr=u, device_program=device_program)
    assert (sym._program is not None) == device_program
    pol = p1mc.MMSCasePol(grid=grid, model=model)
    classes = {"reg": (p1.ForcingTerms_RegHCsTriple, p1.SemiDiscreteField_RegHCsTriple,
                       p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple, {"regularization_factor": ETA}),
               "cs3": (p1.ForcingTerms_CsTriple, p1.SemiDiscreteField_CsTriple,
                       p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple, {}),
               "hcs": (p1.ForcingTerms_HCsTriple, p1.SemiDiscreteField_HCsTriple,
                       p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple, {})}
    F, Fld, Integ, common = classes[kind]
    fa, fb = F(mms_case=sym, model=model, **common), F(mms_case=pol, model=model, **common)
    for t in (0.0, 0.37):
        for name in ("fcp", "fT", "fcl", "fcd", "fcs"):
            np.testing.assert_allclose(getattr(fa, name)(t, grid.xx, grid.yy), getattr(fb, name)(t, grid.xx, grid.yy),
                                       rtol=0, atol=1e-13, err_msg=f"{name} t={t}")
    field_a = Fld(grid=grid, model=model, forcing_terms=fa, **common)
    field_b = Fld(grid=grid, model=model, forcing_terms=fb, **common)
    assert field_a._closed_form_source() is (fa if device_program else None) and field_b._closed_form_source() is fb
    ia, ib = Integ(field_a, **common), Integ(field_b, **common)
    Sa = p1.state_from_mms_when(mms_case=sym, t=0.0, grid=grid)
    Sb = p1.state_from_mms_when(mms_case=pol, t=0.0, grid=grid)
    t, dt = 0.0, 0.02
    for _ in range(3):
        Sa, Sb = ia.step(Sa, t0=t, dt=dt), ib.step(Sb, t0=t, dt=dt)
        t += dt
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_allclose(getattr(Sa, v), getattr(Sb, v), rtol=0, atol=1e-14, err_msg=v)
    for v in ("T", "cl", "cd"):
        np.testing.assert_allclose(ia.last_residual[v], ib.last_residual[v], rtol=0, atol=1e-14, err_msg="res " + v)
    if not device_program:
        with pytest.raises(NotImplementedError):
            ia.advance_with_errors(Sa, t0=t, dt=dt, num_steps=2)
        return
    # device-resident run with the fused error norms, and the field evaluation
    S0a = p1.state_from_mms_when(mms_case=sym, t=0.0, grid=grid)
    S0b = p1.state_from_mms_when(mms_case=pol, t=0.0, grid=grid)
    (Fa, ea), (Fb, eb) = (ia.advance_with_errors(S0a, t0=0.0, dt=dt, num_steps=4),
                          ib.advance_with_errors(S0b, t0=0.0, dt=dt, num_steps=4))
    np.testing.assert_allclose(ea, eb, rtol=1e-9, atol=1e-26)
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_allclose(getattr(Fa, v), getattr(Fb, v), rtol=0, atol=1e-14, err_msg=v)
    np.testing.assert_allclose(field_a.Fcl(S0a, 0.1), field_b.Fcl(S0b, 0.1), rtol=0, atol=1e-13)
    s = field_a._device()
    for what in range(5):
        for a, b in zip(s.ctx.eval_mms(0.37, what), pol._eval(what, 0.37, None, None)):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-13, equal_nan=True)


def check_symbolic_cs_ramp(cs_initial, cs_slope, dt=0.01):
    """reference tests/test_time_integrator_hcs_triple_full_step.py:50-283 as written there: the ramp case
    built with MMSCaseSymbolic and ForcingTerms_HCsTriple."""
    pytest.importorskip("sympy")
    mc = p1.ModelConsts(R0=8.3144621, Ea=1.60217662e-19, K1=1.52e-2, K2=1.52e-2, K3=1.52e-2, K4=1.52e-2,
                        DT=1.22e-3, Dl_max=8.22e-4, phi_l=1.22e-5, gamma_T=1.22e-9, Kd=1.22e-8, Sd=15,
                        Dd_max=2.62e-6, phi_d=1.22e-5, phi_T=1.60217662e-19 / 8.3144621, r_sp=5.22e-2, T_ref=303)
    model = p1.DefaultModel01(mc=mc)
    grid = p1.make_uniform_grid(N=4, M=4)
    profile = p1.x_sym * (1 - p1.x_sym) * p1.y_sym * (1 - p1.y_sym)
    case = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=0, T_sym_expr=0, cl_sym_expr=0, cd_sym_expr=0,
                              cs_sym_expr=(cs_initial + cs_slope * p1.t_sym) * profile)
    forcing = p1.ForcingTerms_HCsTriple(mms_case=case, model=model)
    field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model, forcing_terms=forcing)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(field, num_pc_steps=1, num_newton_steps=1)
    S0 = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    S1 = integ.step(S0, t0=0.0, dt=dt)
    for v in ("cp", "T", "cl", "cd"):
        np.testing.assert_allclose(getattr(S1, v), getattr(case, v)(dt, grid.xx, grid.yy), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(S1.cs, case.cs(dt, grid.xx, grid.yy), rtol=1e-6, atol=1e-12)


def check_symbolic_library_case_golden(cname, kname):
    """The library cases stated only as SymPy expressions (reference src/prob1_mms_cases.py:12-148,
    :341-392), built through the mirrored classes: forcing arrays at two times and 1/2/4
    predictor-corrector steps against vectors of the reference itself (tests/golden/make_golden.py,
    gen_symbolic), for the three reaction kinds."""
    pytest.importorskip("sympy")
    z = np.load(os.path.join(GOLDEN, "golden_symbolic.npz"))
    g = lambda k: z[f"{cname}/{k}"]
    par = {k.split("/")[-1]: float(z[k]) for k in z.files if k.startswith(f"{cname}/param/")}
    ctor = {"stiff_exp_decay": p1mc.MMSCaseStiffExpDecay, "pol_oscillating": p1mc.MMSCasePolWithOscilatingTime,
            "cs_zero_crossing": p1mc.MMSCaseCsZeroCrossing,
            "from_analytic": lambda grid, model: p1.MMSCaseFromAnalytic(model, grid=grid, **analytic_fns.BASES)}[cname]
    consts = dict(zip(p1.ModelConsts._fields, g("consts")))
    model = p1.DefaultModel02(p1.ModelConsts(**consts))
    grid = p1.Grid(g("x"), g("y"))
    case = ctor(grid=grid, model=model, **par)
    eta = float(g("eta"))
    common = {"regularization_factor": eta} if kname == "reg" else {}
    field_cls, forcing_cls, integ_cls = {
        "reg": (p1.SemiDiscreteField_RegHCsTriple, p1.ForcingTerms_RegHCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple),
        "cst": (p1.SemiDiscreteField_CsTriple, p1.ForcingTerms_CsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple),
        "hcs": (p1.SemiDiscreteField_HCsTriple, p1.ForcingTerms_HCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple)}[kname]
    forcing = forcing_cls(mms_case=case, model=model, **common)
    for it in (0, 1):
        t = float(g(f"{kname}/t{it}/t"))
        for v in ("fcp", "fT", "fcl", "fcd", "fcs"):
            np.testing.assert_allclose(getattr(forcing, v)(t, grid.xx, grid.yy), g(f"{kname}/t{it}/{v}"),
                                       rtol=1e-12, atol=1e-15, err_msg=f"{cname}/{kname}/{v}@{t}")
    field = field_cls(grid=grid, model=model, forcing_terms=forcing, **common)
    integ = integ_cls(field, **common)
    t, dt = float(g(f"{kname}/steps/t0")), float(g(f"{kname}/steps/dt"))
    S = p1.state_from_mms_when(mms_case=case, t=t, grid=grid)
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_allclose(getattr(S, v), g(f"{kname}/steps/s0/{v}"), rtol=1e-13, atol=1e-16)
    for k in range(1, 5):
        S = integ.step(S, t0=t, dt=dt)
        t += dt
        if k in (1, 2, 4):
            for v in ("cp", "T", "cl", "cd", "cs"):
                np.testing.assert_allclose(getattr(S, v), g(f"{kname}/steps/s{k}/{v}"), rtol=1e-11, atol=1e-15,
                                           err_msg=f"{cname}/{kname}/s{k}/{v}")
            for v in ("T", "cl", "cd"):
                np.testing.assert_allclose(integ.last_residual[v], g(f"{kname}/steps/s{k}/res_{v}"),
                                           rtol=1e-6, atol=1e-13, err_msg=f"{cname}/{kname}/s{k}/res_{v}")


def check_error_behaviour():
    model = model_of("nb_a")
    trial = reg_trial(p1mc.MMSCaseExpSin, model, 4)
    S0 = trial.initial_state
    with pytest.raises(AssertionError):
        trial.integrator.step(S0, t0=0.0, dt=0.0)          # reference: assert dt > 0
    with pytest.raises(AttributeError):
        S0.T = np.zeros_like(S0.T)                          # StateVars is immutable
    with pytest.raises(ValueError):
        S0.with_changes(foo=1)
    # model constants mutated in place are seen by the next step
    S1 = trial.integrator.step(S0, t0=0.0, dt=1e-3)
    trial.integrator._model.Kd = 0.5
    S1b = trial.integrator.step(S0, t0=0.0, dt=1e-3)
    assert np.max(np.abs(S1b.cs - S1.cs)) > 1e-8
    # a non-device field is refused, not silently run on the CPU
    class Dummy:
        pass
    dummy = Dummy()
    dummy.model, dummy.grid = model, trial.grid
    with pytest.raises(NotImplementedError):
        p1.ForwardEulerIntegrator(dummy).step(S0, t0=0.0, dt=1e-3)


def check_overridden_field_methods_are_refused():
    """The reference's tests change a step by overriding a method of the field (Fcs in
    tests/test_time_integrator_hcs_triple.py:208-226, forcing callables patched onto a live field in
    tests/test_time_integrator_hcs_triple_full_step.py:189-196).  The device step cannot call back into Python:
    an overridden field method is refused by name (subclass or instance attribute), for every integrator and for
    the pieces of a step; a re-assigned forcing callable is honoured (streamed to the device)."""
    model = model_of("nb_a")
    grid = p1.make_uniform_grid(N=6, M=6)
    case = p1mc.MMSCaseExpSin(grid, model=model)

    class FieldWithFcs(p1.SemiDiscreteField_HCsTriple):
        def Fcs(self, at_t, t):
            return np.zeros(self.grid.full_shape)

    ft = p1.ForcingTerms_HCsTriple(mms_case=case, model=model)
    S0 = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    sub = FieldWithFcs(grid=grid, model=model, forcing_terms=ft)
    for integ in (p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(sub), p1.ForwardEulerIntegrator(sub)):
        with pytest.raises(NotImplementedError, match="overrides Fcs"):
            integ.step(S0, t0=0.0, dt=1e-3)
    with pytest.raises(NotImplementedError, match="overrides Fcs"):
        p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(sub).initial_cs_pred(S0, 0.0, dt=1e-3)
    # the same on an instance, and for the interaction law and a Jacobian stencil
    plain = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, regularization_factor=50.0,
                                              forcing_terms=p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model,
                                                                                        regularization_factor=50.0))
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(plain, regularization_factor=50.0)
    S1 = integ.step(S0, t0=0.0, dt=1e-3)                       # untouched field: runs
    plain.cscd_reaction_cs = lambda cs: cs
    plain.delcd_ab_Fcd_ij = lambda state, *, a, b: 0.0
    with pytest.raises(NotImplementedError, match="delcd_ab_Fcd_ij.*cscd_reaction_cs|cscd_reaction_cs.*delcd_ab_Fcd_ij"):
        integ.step(S0, t0=0.0, dt=1e-3)
    del plain.cscd_reaction_cs, plain.delcd_ab_Fcd_ij
    # a re-assigned forcing callable is not an override of a kernel: it changes the step
    plain.fcs = lambda t, xx, yy: np.full(grid.full_shape, 0.25)
    S2 = integ.step(S0, t0=0.0, dt=1e-3)
    assert np.max(np.abs(S2.cs - S1.cs)) > 1e-5
