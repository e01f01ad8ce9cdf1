#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the development container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src \
        python3 tests/golden/make_golden.py [--scalars]

The fixtures are plain data (inputs + the reference's outputs, float64 .npz and
.json).  Nothing of the reference's source is stored.  Files written:

  golden_ops.npz      per-function vectors: F* field evaluations, Jacobian stencil
                      coefficient arrays, coupling mat-vecs, discrete norms, on small
                      uniform / non-uniform grids with seeded random states
  golden_mms.npz      exact MMS fields, their derivatives and the five forcing arrays
                      for the four configured MMS cases at two times
  golden_steps.npz    states after 1/2/5 integrator.step() calls (+ last_residual)
                      for forward Euler and the predictor-corrector integrators
  golden_hcs.npz      the same three kinds of vectors for the HCsTriple (sharp Heaviside)
                      field, forcing terms and predictor-corrector integrator
  golden_substeps.npz every piece of one predictor-corrector step called on its own
                      (initial_*_pred, newton_step_*, corrector_*_step) for the three integrators
  golden_symbolic.npz forcing arrays and integrator steps for the three cases the reference states
                      symbolically only (StiffExpDecay, PolWithOscilatingTime, CsZeroCrossing)
  golden_isolated.json truncation errors of FT / Fcl / Fcd on exact states and errors of the coupled T-cl and
                      T-cl-cd subsystems, all with degenerate model constants
  golden_reghcs.json  RegHCsTriple scenarios for three regularisation factors: Heaviside values, field vs time
                      derivative, one-step errors, forward-Euler temporal errors, short MMSTrial runs
  golden_statevars.npz the derived arrays of a StateVars on random fields (both model kinds)
  golden_studies.json the report of cvg_studies_base.run_convergence_studies (legacy study runner) for
                      two small CsTriple / HCsTriple studies
  golden_scalars.json MMSTrial.run_for_errors summaries (only with --scalars; slow)
  published_notebook_errors.json   numbers printed in the reference's executed
                      notebooks (parsed from the .ipynb stream outputs)
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

import prob1base as p1
import prob1_mms_cases as p1mc
import mms_trial_utils as mtu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analytic_fns  # noqa: E402  (tests/analytic_fns.py)
import isolated_cases  # noqa: E402  (tests/isolated_cases.py)
import reghcs_cases  # noqa: E402  (tests/reghcs_cases.py)

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("TEDD_REFERENCE_ROOT", "/root/reference")

R0 = 8.3144621
Ea = 1.60217662e-19

CONST_ORDER = list(p1.ModelConsts._fields)  # 17 names, reference order

MODEL_SETS = {
    # notebooks: ExpSin and NonFullySmoothPol studies
    "nb_a": dict(R0=R0, Ea=Ea, K1=1e-3, K2=1e-3, K3=1e-3, K4=1e-3, DT=1e-3, Dl_max=1e-5,
                 phi_l=1e-5, gamma_T=1e-9, Kd=1e-2, Sd=1, Dd_max=1e-6, phi_d=1e-5,
                 phi_T=Ea / R0, r_sp=5e-2, T_ref=300),
    # notebooks: Pol and SlowlyChangingPeaks studies
    "nb_b": dict(R0=R0, Ea=Ea, K1=1e-3, K2=1e-3, K3=1e-3, K4=1e-3, DT=1e-3, Dl_max=8.01e-4,
                 phi_l=1e-5, gamma_T=1e-9, Kd=1e-2, Sd=1, Dd_max=2.46e-6, phi_d=1e-5,
                 phi_T=Ea / R0, r_sp=5e-2, T_ref=300),
    # tests/test_time_integration_fwd_euler_full_p1base.py
    "fe": dict(R0=R0, Ea=Ea, K1=1e-2, K2=1e-2, K3=1e-2, K4=1e-2, DT=1e-3, Dl_max=8.01e-4,
               phi_l=1e-5, gamma_T=1e-9, Kd=1e-8, Sd=10, Dd_max=2.46e-6, phi_d=1e-5,
               phi_T=Ea / R0, r_sp=5e-2, T_ref=300),
    # exaggerated nonlinearity so every term of every Jacobian is visible in float64
    "stress": dict(R0=R0, Ea=Ea, K1=0.11, K2=0.23, K3=0.37, K4=0.19, DT=2e-2, Dl_max=3e-2,
                   phi_l=0.4, gamma_T=0.03, Kd=0.2, Sd=1.5, Dd_max=1e-2, phi_d=0.3,
                   phi_T=50.0, r_sp=5e-2, T_ref=300),
}


def consts_vec(name):
    d = MODEL_SETS[name]
    return np.array([float(d[k]) for k in CONST_ORDER])


def make_model(name, kind=2):
    mc = p1.ModelConsts(**MODEL_SETS[name])
    return p1.DefaultModel02(mc) if kind == 2 else p1.DefaultModel01(mc)


def nonuniform_axis(n, power, wiggle):
    s = np.linspace(0.0, 1.0, n + 1)
    x = s ** power + wiggle * np.sin(2 * np.pi * s) / (2 * np.pi * 3)
    x[0] = 0.0
    x[-1] = 1.0
    assert np.all(np.diff(x) > 0)
    return x


MMS_CASES = {
    "pol": (p1mc.MMSCasePol, "nb_b"),
    "expsin": (p1mc.MMSCaseExpSin, "nb_a"),
    "peaks_fast1e1": (p1mc.MMSCaseSlowlyChangingPeaks_Fast1e1, "nb_b"),
    "nfs_h2_h3": (p1mc.MMSCaseNonFullySmoothPol_cpcsH2_TclcdH3, "nb_a"),
}

OFFS = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)]  # order of the 5 coefficient arrays


# ----------------------------------------------------------------------------------
def gen_ops():
    out = {}
    rng = np.random.default_rng(20250503)
    grids = {
        "u4x4": (np.linspace(0, 1, 5), np.linspace(0, 1, 5)),
        "u7x5": (np.linspace(0, 1, 8), np.linspace(0, 1, 6)),
        "nu6x9": (nonuniform_axis(6, 1.3, 0.2), nonuniform_axis(9, 0.8, -0.15)),
    }
    for gname, (x, y) in grids.items():
        grid = p1.Grid(x, y)
        Np1, Mp1 = grid.N + 1, grid.M + 1
        for mname, kind, Tshift in [("stress", 2, 0.0), ("fe", 1, 273.15), ("nb_a", 2, 0.0)]:
            tag = f"{gname}/{mname}{kind}"
            model = make_model(mname, kind)
            cp = rng.random((Np1, Mp1)) + 0.1
            T = rng.random((Np1, Mp1)) * (100 if Tshift else 1.0) + Tshift
            cl = rng.random((Np1, Mp1))
            cd = rng.random((Np1, Mp1))
            cs = rng.random((Np1, Mp1)) * 0.2 - 0.05  # crosses zero: exercises H_eta
            w1 = rng.standard_normal((Np1, Mp1))
            w2 = rng.standard_normal((Np1, Mp1))
            eta = 50.0
            out[f"{tag}/x"] = x
            out[f"{tag}/y"] = y
            out[f"{tag}/consts"] = consts_vec(mname)
            out[f"{tag}/model_kind"] = np.array(kind)
            out[f"{tag}/eta"] = np.array(eta)
            for nm, arr in [("cp", cp), ("T", T), ("cl", cl), ("cd", cd), ("cs", cs),
                            ("w1", w1), ("w2", w2)]:
                out[f"{tag}/{nm}"] = arr
            state = p1.StateVars(cp, T, cl, cd, cs, model=model, hh=grid.hh, kk=grid.kk)
            forcing = p1.NoForcingTerms(grid)
            fields = {
                "reg": p1.SemiDiscreteField_RegHCsTriple(
                    grid=grid, model=model, forcing_terms=forcing, regularization_factor=eta),
                "cs3": p1.SemiDiscreteField_CsTriple(grid=grid, model=model, forcing_terms=forcing),
            }
            for fname, field in fields.items():
                ft = f"{tag}/{fname}"
                out[f"{ft}/Fcp"] = field.Fcp(state, 0.3)
                out[f"{ft}/FT"] = field.FT(state, 0.3)
                out[f"{ft}/Fcl"] = field.Fcl(state, 0.3)
                out[f"{ft}/Fcd"] = field.Fcd(state, 0.3)
                out[f"{ft}/Fcs"] = field.Fcs(state, 0.3)
                for opname in ["delT_ab_FT_ij", "delT_ab_Fcl_ij", "delcl_ab_Fcl_ij",
                               "delT_ab_Fcd_ij", "delcl_ab_Fcd_ij", "delcd_ab_Fcd_ij"]:
                    fn = getattr(field, opname)
                    out[f"{ft}/{opname}"] = np.stack([fn(state, a=a, b=b) for a, b in OFFS])
                out[f"{ft}/delT_Fcl_at_w"] = field.delT_Fcl_at_w(state, w1)
                out[f"{ft}/delT_Fcd_at_w"] = field.delT_Fcd_at_w(state, w1)
                out[f"{ft}/delcl_Fcd_at_w"] = field.delcl_Fcd_at_w(state, w2)
            out[f"{tag}/norm_H_w1"] = np.array(grid.norm_H(w1))
            gx, gy = grid.grad_H(w1)
            out[f"{tag}/norm_p_grad_w1"] = np.array(grid.norm_p(gx, gy))
            out[f"{tag}/heaviside_cs"] = p1.heaviside_regularized(cs, eta)
    np.savez_compressed(os.path.join(HERE, "golden_ops.npz"), **out)
    print("golden_ops.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def gen_mms():
    out = {}
    grids = {
        "u6x5": (np.linspace(0, 1, 7), np.linspace(0, 1, 6)),
        "nu5x7": (nonuniform_axis(5, 1.2, 0.1), nonuniform_axis(7, 0.9, -0.1)),
    }
    eta = 50.0
    for cname, (cls, mname) in MMS_CASES.items():
        for gname, (x, y) in grids.items():
            grid = p1.Grid(x, y)
            for mm, kind in [(mname, 2), ("stress", 2)]:
                model = make_model(mm, kind)
                case = cls(grid=grid, model=model)
                tag = f"{cname}/{gname}/{mm}{kind}"
                out[f"{tag}/x"] = x
                out[f"{tag}/y"] = y
                out[f"{tag}/consts"] = consts_vec(mm)
                out[f"{tag}/model_kind"] = np.array(kind)
                out[f"{tag}/eta"] = np.array(eta)
                freg = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model,
                                                    regularization_factor=eta)
                fcs3 = p1.ForcingTerms_CsTriple(mms_case=case, model=model)
                for it, t in enumerate([0.0, 0.37]):
                    tt = f"{tag}/t{it}"
                    out[f"{tt}/t"] = np.array(t)
                    xx, yy = grid.xx, grid.yy
                    for v in ["cp", "T", "cl", "cd", "cs"]:
                        out[f"{tt}/{v}"] = getattr(case, v)(t, xx, yy)
                        out[f"{tt}/dt_{v}"] = getattr(case, f"dt_{v}")(t, xx, yy)
                    for v in ["cp", "T", "cl", "cd"]:
                        out[f"{tt}/dx_{v}"] = getattr(case, f"dx_{v}")(t, xx, yy)
                        out[f"{tt}/dy_{v}"] = getattr(case, f"dy_{v}")(t, xx, yy)
                    for v in ["T", "cl", "cd"]:
                        out[f"{tt}/lap_{v}"] = getattr(case, f"lap_{v}")(t, xx, yy)
                    for fname, f in [("reg", freg), ("cs3", fcs3)]:
                        out[f"{tt}/{fname}/fcp"] = f.fcp(t, xx, yy)
                        out[f"{tt}/{fname}/fT"] = f.fT(t, xx, yy)
                        out[f"{tt}/{fname}/fcl"] = f.fcl(t, xx, yy)
                        out[f"{tt}/{fname}/fcd"] = f.fcd(t, xx, yy)
                        out[f"{tt}/{fname}/fcs"] = f.fcs(t, xx, yy)
                    out[f"{tt}/fcp_ptwise"] = fcs3.fcp_ptwise(t, xx, yy)
    np.savez_compressed(os.path.join(HERE, "golden_mms.npz"), **out)
    print("golden_mms.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def _trial(cls, model, grid, integ, field_kind, eta, integ_params):
    if field_kind == "reg":
        common = {"regularization_factor": eta}
        return mtu.MMSTrial(
            grid=grid, model=model, mms_case_cls=cls,
            field_cls=p1.SemiDiscreteField_RegHCsTriple,
            forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
            integrator_cls=integ, integrator_params=integ_params,
            forcing_terms_params=common, field_params=common)
    return mtu.MMSTrial(
        grid=grid, model=model, mms_case_cls=cls,
        field_cls=p1.SemiDiscreteField_CsTriple,
        forcing_terms_cls=p1.ForcingTerms_CsTriple,
        integrator_cls=integ, integrator_params=integ_params)


def _record_steps(out, tag, trial, t0, dt, save_at, with_residual):
    state = trial.initial_state if t0 == 0.0 else p1.state_from_mms_when(
        mms_case=trial.mms_case, t=t0, grid=trial.grid)
    out[f"{tag}/t0"] = np.array(t0)
    out[f"{tag}/dt"] = np.array(dt)
    for v in ["cp", "T", "cl", "cd", "cs"]:
        out[f"{tag}/s0/{v}"] = getattr(state, v)
    t = t0
    for k in range(1, max(save_at) + 1):
        state = trial.integrator.step(state, t0=t, dt=dt)
        t += dt
        if k in save_at:
            for v in ["cp", "T", "cl", "cd", "cs"]:
                out[f"{tag}/s{k}/{v}"] = getattr(state, v)
            if with_residual:
                for v in ["T", "cl", "cd"]:
                    out[f"{tag}/s{k}/res_{v}"] = trial.integrator.last_residual[v]


def gen_steps():
    out = {}
    PC_REG = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple
    PC_CS3 = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple
    FE = p1.ForwardEulerIntegrator
    eta = 50.0

    def put_common(tag, x, y, mname, kind, eta_, extra=None):
        out[f"{tag}/x"] = x
        out[f"{tag}/y"] = y
        out[f"{tag}/consts"] = consts_vec(mname)
        out[f"{tag}/model_kind"] = np.array(kind)
        out[f"{tag}/eta"] = np.array(eta_)
        for k, v in (extra or {}).items():
            out[f"{tag}/{k}"] = np.array(v)

    # (1) predictor-corrector RegHCsTriple, the four configured cases, dt = h^1.5
    for cname, (cls, mname) in MMS_CASES.items():
        for N in [2, 5, 8, 16]:
            x = np.linspace(0, 1, N + 1)
            grid = p1.Grid(x, x.copy())
            model = make_model(mname, 2)
            tag = f"pc_reg/{cname}/N{N}"
            put_common(tag, x, x, mname, 2, eta,
                       dict(num_pc_steps=1, num_newton_steps=1, num_newton_iterations=5,
                            consec_xs_rtol=1e-6))
            trial = _trial(cls, model, grid, PC_REG, "reg", eta, {"regularization_factor": eta})
            _record_steps(out, tag, trial, 0.0, (1.0 / N) ** 1.5, [1, 2, 5], True)
    # (1b) ExpSin at the notebooks' FIXED_GRID_SIZE
    N = 32
    x = np.linspace(0, 1, N + 1)
    grid = p1.Grid(x, x.copy())
    tag = "pc_reg/expsin/N32"
    put_common(tag, x, x, "nb_a", 2, eta, dict(num_pc_steps=1, num_newton_steps=1,
                                               num_newton_iterations=5, consec_xs_rtol=1e-6))
    trial = _trial(p1mc.MMSCaseExpSin, make_model("nb_a", 2), grid, PC_REG, "reg", eta,
                   {"regularization_factor": eta})
    _record_steps(out, tag, trial, 0.0, (1.0 / N) ** 1.5, [1, 2], True)

    # (2) variants: more Newton / PC iterations, 1 cs-Newton iteration, other eta, stress
    variants = [
        ("v_newton2", "expsin", "nb_a", 2, 8, 50.0, dict(num_newton_steps=2)),
        ("v_pc2", "expsin", "nb_a", 2, 8, 50.0, dict(num_pc_steps=2)),
        ("v_csit1", "pol", "nb_b", 2, 8, 50.0, dict(num_newton_iterations=1)),
        ("v_eta10", "expsin", "nb_a", 2, 8, 10.0, dict()),
        ("v_eta1000", "expsin", "nb_a", 2, 8, 1000.0, dict()),
        ("v_stress_expsin", "expsin", "stress", 2, 8, 50.0, dict()),
        ("v_stress_pol", "pol", "stress", 2, 6, 50.0, dict()),
        ("v_stress_peaks", "peaks_fast1e1", "stress", 2, 6, 50.0, dict()),
        ("v_stress_nfs", "nfs_h2_h3", "stress", 2, 6, 50.0, dict()),
        ("v_model01", "pol", "fe", 1, 6, 50.0, dict()),
        ("v_bigdt", "pol", "nb_b", 2, 24, 50.0, dict(_dt=1.0)),
    ]
    for vname, cname, mname, kind, N, eta_v, ip in variants:
        ip = dict(ip)
        dt = ip.pop("_dt", (1.0 / N) ** 1.5)
        cls = MMS_CASES[cname][0]
        x = np.linspace(0, 1, N + 1)
        grid = p1.Grid(x, x.copy())
        model = make_model(mname, kind)
        tag = f"pc_reg/{vname}"
        params = dict(num_pc_steps=1, num_newton_steps=1, num_newton_iterations=5,
                      consec_xs_rtol=1e-6)
        params.update(ip)
        put_common(tag, x, x, mname, kind, eta_v, dict(params, case=cname))
        trial = _trial(cls, model, grid, PC_REG, "reg", eta_v,
                       dict(regularization_factor=eta_v, **ip))
        _record_steps(out, tag, trial, 0.0, dt, [1, 2, 5], True)

    # (3) non-uniform, non-square grid
    x = nonuniform_axis(6, 1.3, 0.2)
    y = nonuniform_axis(9, 0.8, -0.15)
    for cname, mname in [("expsin", "nb_a"), ("pol", "stress")]:
        grid = p1.Grid(x, y)
        tag = f"pc_reg/nonuniform_{cname}"
        put_common(tag, x, y, mname, 2, eta, dict(num_pc_steps=1, num_newton_steps=1,
                                                 num_newton_iterations=5, consec_xs_rtol=1e-6,
                                                 case=cname))
        trial = _trial(MMS_CASES[cname][0], make_model(mname, 2), grid, PC_REG, "reg", eta,
                       {"regularization_factor": eta})
        _record_steps(out, tag, trial, 0.0, 0.02, [1, 2, 5], True)

    # (4) forward Euler: the config-1 test path (CsTriple/ExpSin) and RegHCsTriple/Pol
    for N in [5, 15]:
        x = np.linspace(0, 1, N + 1)
        grid = p1.Grid(x, x.copy())
        tag = f"fe_cs3/expsin/N{N}"
        put_common(tag, x, x, "fe", 2, 0.0)
        trial = _trial(p1mc.MMSCaseExpSin, make_model("fe", 2), grid, FE, "cs3", 0.0, {})
        h2 = (1.0 / N) ** 2
        dt = min(4e-4, 0.5 * h2 / (4 * 1e-3))
        _record_steps(out, tag, trial, 0.05, dt, [1, 3], False)
    for N in [2, 5]:
        x = np.linspace(0, 1, N + 1)
        grid = p1.Grid(x, x.copy())
        tag = f"fe_reg/pol/N{N}"
        put_common(tag, x, x, "nb_b", 2, eta)
        trial = _trial(p1mc.MMSCasePol, make_model("nb_b", 2), grid, FE, "reg", eta, {})
        _record_steps(out, tag, trial, 0.0, (1.0 / N) ** 1.5 * 0.1, [1, 3], False)

    # (5) CsTriple predictor-corrector (closed-form cs corrector)
    N = 8
    x = np.linspace(0, 1, N + 1)
    grid = p1.Grid(x, x.copy())
    tag = "pc_cs3/expsin/N8"
    put_common(tag, x, x, "nb_a", 2, 0.0, dict(num_pc_steps=1, num_newton_steps=1))
    trial = _trial(p1mc.MMSCaseExpSin, make_model("nb_a", 2), grid, PC_CS3, "cs3", 0.0, {})
    _record_steps(out, tag, trial, 0.0, (1.0 / N) ** 1.5, [1, 2, 5], True)

    np.savez_compressed(os.path.join(HERE, "golden_steps.npz"), **out)
    print("golden_steps.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def gen_hcs():
    """HCsTriple: field / Jacobians on random states, forcing terms, integrator steps."""
    out = {}
    PC_HCS = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple
    rng = np.random.default_rng(20250611)
    # (a) field and Jacobian stencils, zero forcing, cs crossing zero
    grids = {"u5x4": (np.linspace(0, 1, 6), np.linspace(0, 1, 5)),
             "nu6x9": (nonuniform_axis(6, 1.3, 0.2), nonuniform_axis(9, 0.8, -0.15))}
    for gname, (x, y) in grids.items():
        grid = p1.Grid(x, y)
        shp = (grid.N + 1, grid.M + 1)
        for mname, kind in [("stress", 2), ("nb_a", 2)]:
            tag = f"ops/{gname}/{mname}{kind}"
            model = make_model(mname, kind)
            S = {"cp": rng.random(shp) + 0.1, "T": rng.random(shp), "cl": rng.random(shp),
                 "cd": rng.random(shp), "cs": rng.random(shp) * 0.2 - 0.05}
            S["cs"][1, 1] = 0.0  # H(0) = 0
            out[f"{tag}/x"], out[f"{tag}/y"] = x, y
            out[f"{tag}/consts"] = consts_vec(mname)
            out[f"{tag}/model_kind"] = np.array(kind)
            out[f"{tag}/eta"] = np.array(0.0)
            for v, a in S.items():
                out[f"{tag}/{v}"] = a
            state = p1.StateVars(S["cp"], S["T"], S["cl"], S["cd"], S["cs"], model=model, hh=grid.hh, kk=grid.kk)
            field = p1.SemiDiscreteField_HCsTriple(grid=grid, model=model, forcing_terms=p1.NoForcingTerms(grid))
            for v in ["cp", "T", "cl", "cd", "cs"]:
                out[f"{tag}/hcs/F{v}"] = getattr(field, f"F{v}")(state, 0.3)
            for opname in ["delT_ab_FT_ij", "delT_ab_Fcl_ij", "delcl_ab_Fcl_ij", "delT_ab_Fcd_ij",
                           "delcl_ab_Fcd_ij", "delcd_ab_Fcd_ij"]:
                fn = getattr(field, opname)
                out[f"{tag}/hcs/{opname}"] = np.stack([fn(state, a=a, b=b) for a, b in OFFS])
    # (b) forcing terms of the four configured cases at two times
    x, y = np.linspace(0, 1, 7), nonuniform_axis(7, 0.9, -0.1)
    for cname, (cls, mname) in MMS_CASES.items():
        grid = p1.Grid(x, y)
        model = make_model(mname, 2)
        case = cls(grid=grid, model=model)
        tag = f"mms/{cname}"
        out[f"{tag}/x"], out[f"{tag}/y"] = x, y
        out[f"{tag}/consts"] = consts_vec(mname)
        out[f"{tag}/model_kind"] = np.array(2)
        out[f"{tag}/eta"] = np.array(0.0)
        f = p1.ForcingTerms_HCsTriple(mms_case=case, model=model)
        for it, t in enumerate([0.0, 0.37]):
            out[f"{tag}/t{it}/t"] = np.array(t)
            out[f"{tag}/t{it}/cp"] = case.cp(t, grid.xx, grid.yy)
            for v in ["fcp", "fT", "fcl", "fcd", "fcs"]:
                out[f"{tag}/t{it}/hcs/{v}"] = getattr(f, v)(t, grid.xx, grid.yy)

    def hcs_trial(cls, model, grid, integ, integ_params):
        return mtu.MMSTrial(grid=grid, model=model, mms_case_cls=cls,
                            field_cls=p1.SemiDiscreteField_HCsTriple,
                            forcing_terms_cls=p1.ForcingTerms_HCsTriple,
                            integrator_cls=integ, integrator_params=integ_params)

    def put_common(tag, x, y, mname, extra=None):
        out[f"{tag}/x"], out[f"{tag}/y"] = x, y
        out[f"{tag}/consts"] = consts_vec(mname)
        out[f"{tag}/model_kind"] = np.array(2)
        out[f"{tag}/eta"] = np.array(0.0)
        for k, v in (extra or {}).items():
            out[f"{tag}/{k}"] = np.array(v)

    # (c) predictor-corrector steps, dt = h^1.5
    for cname, (cls, mname) in MMS_CASES.items():
        for N in [5, 8, 16]:
            xu = np.linspace(0, 1, N + 1)
            tag = f"steps/pc_hcs/{cname}/N{N}"
            put_common(tag, xu, xu, mname, dict(num_pc_steps=1, num_newton_steps=1))
            trial = hcs_trial(cls, make_model(mname, 2), p1.Grid(xu, xu.copy()), PC_HCS, {})
            _record_steps(out, tag, trial, 0.0, (1.0 / N) ** 1.5, [1, 2, 5], True)
    # two Newton steps and two predictor-corrector iterations; a non-uniform mesh
    xu = np.linspace(0, 1, 9)
    for vname, ip in [("v_newton2", dict(num_newton_steps=2)), ("v_pc2", dict(num_pc_steps=2))]:
        tag = f"steps/pc_hcs/{vname}"
        params = dict(num_pc_steps=1, num_newton_steps=1)
        params.update(ip)
        put_common(tag, xu, xu, "nb_a", dict(params, case="expsin"))
        trial = hcs_trial(p1mc.MMSCaseExpSin, make_model("nb_a", 2), p1.Grid(xu, xu.copy()), PC_HCS, ip)
        _record_steps(out, tag, trial, 0.0, (1.0 / 8) ** 1.5, [1, 2, 5], True)
    xn, yn = nonuniform_axis(6, 1.3, 0.2), nonuniform_axis(9, 0.8, -0.15)
    tag = "steps/pc_hcs/nonuniform_pol"
    put_common(tag, xn, yn, "stress", dict(num_pc_steps=1, num_newton_steps=1, case="pol"))
    trial = hcs_trial(p1mc.MMSCasePol, make_model("stress", 2), p1.Grid(xn, yn), PC_HCS, {})
    _record_steps(out, tag, trial, 0.0, 0.02, [1, 2, 5], True)
    # cs of both signs and exact zeros at t0: all three branches of the sign-case corrector
    xu = np.linspace(0, 1, 9)
    tag = "steps/pc_hcs/v_signs"
    put_common(tag, xu, xu, "nb_b", dict(num_pc_steps=1, num_newton_steps=1, case="pol"))
    trial = hcs_trial(p1mc.MMSCasePol, make_model("nb_b", 2), p1.Grid(xu, xu.copy()), PC_HCS, {})
    s0 = trial.initial_state
    cs = s0.cs - 0.03
    cs[2, :] = 0.0
    trial.initial_state = s0.with_changes(cs=cs)
    _record_steps(out, tag, trial, 0.0, (1.0 / 8) ** 1.5, [1, 2, 5], True)
    # (d) forward Euler on the HCsTriple field
    xu = np.linspace(0, 1, 6)
    tag = "steps/fe_hcs/pol/N5"
    put_common(tag, xu, xu, "nb_b")
    trial = hcs_trial(p1mc.MMSCasePol, make_model("nb_b", 2), p1.Grid(xu, xu.copy()), p1.ForwardEulerIntegrator, {})
    _record_steps(out, tag, trial, 0.0, (1.0 / 5) ** 1.5 * 0.1, [1, 3], False)

    np.savez_compressed(os.path.join(HERE, "golden_hcs.npz"), **out)
    print("golden_hcs.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def symbolic_cases():
    return {
        "stiff_exp_decay": (lambda grid, model: p1mc.MMSCaseStiffExpDecay(grid=grid, model=model, a_base=50.0),
                            dict(a_base=50.0), 0.0),
        "pol_oscillating": (p1mc.make_MMSCasePolWithOscilatingTime_cls(ampl=0.7, speed=12.0),
                            dict(ampl=0.7, speed=12.0), 0.0),
        # starts just before the sign change of cs at t = cs_A / cs_B = 0.5
        "cs_zero_crossing": (p1mc.MMSCaseCsZeroCrossing, {}, 0.49),
        # plain callables differentiated by finite differences (MMSCaseFromAnalytic)
        "from_analytic": (lambda grid, model: p1.MMSCaseFromAnalytic(model, grid=grid, **analytic_fns.BASES), {}, 0.1),
    }


def gen_symbolic():
    """The cases that exist only as SymPy expressions in the reference: forcing arrays of the three
    reaction kinds at two times, and 1/2/4 predictor-corrector steps per kind."""
    out = {}
    x, y = np.linspace(0, 1, 9), nonuniform_axis(7, 0.9, -0.1)
    eta = 1e-3
    kinds = {
        "reg": (p1.SemiDiscreteField_RegHCsTriple, p1.ForcingTerms_RegHCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple, {"regularization_factor": eta}),
        "cst": (p1.SemiDiscreteField_CsTriple, p1.ForcingTerms_CsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple, {}),
        "hcs": (p1.SemiDiscreteField_HCsTriple, p1.ForcingTerms_HCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple, {}),
    }
    for cname, (cls, params, t0) in symbolic_cases().items():
        tag = f"{cname}"
        out[f"{tag}/x"], out[f"{tag}/y"] = x, y
        out[f"{tag}/consts"] = consts_vec("nb_b")
        out[f"{tag}/eta"] = np.array(eta)
        out[f"{tag}/t0"] = np.array(t0)
        out[f"{tag}/dt"] = np.array(5e-3)
        for k, v in params.items():
            out[f"{tag}/param/{k}"] = np.array(v)
        for kname, (field_cls, forcing_cls, integ_cls, common) in kinds.items():
            grid = p1.Grid(x.copy(), y.copy())
            model = make_model("nb_b", 2)
            trial = mtu.MMSTrial(grid=grid, model=model, mms_case_cls=cls, field_cls=field_cls,
                                 forcing_terms_cls=forcing_cls, integrator_cls=integ_cls,
                                 integrator_params=dict(common), forcing_terms_params=dict(common),
                                 field_params=dict(common))
            for it, t in enumerate([t0, t0 + 0.013]):
                out[f"{tag}/{kname}/t{it}/t"] = np.array(t)
                for v in ["fcp", "fT", "fcl", "fcd", "fcs"]:
                    out[f"{tag}/{kname}/t{it}/{v}"] = getattr(trial.forcing_terms, v)(t, grid.xx, grid.yy)
            _record_steps(out, f"{tag}/{kname}/steps", trial, t0, 5e-3, [1, 2, 4], True)
    np.savez_compressed(os.path.join(HERE, "golden_symbolic.npz"), **out)
    print("golden_symbolic.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def gen_reghcs():
    """tests/reghcs_cases.py run with the reference's modules."""
    out = reghcs_cases.everything(p1, p1mc, mtu)
    with open(os.path.join(HERE, "golden_reghcs.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("golden_reghcs.json written")


# ----------------------------------------------------------------------------------
def gen_statevars():
    """The 24 derived arrays of StateVars (face means, backward differences, Dl / V / Dd at the faces) on
    seeded random fields over a non-uniform 8 x 6 node mesh, for both model kinds."""
    rng = np.random.default_rng(20250503)
    x, y = nonuniform_axis(7, 0.9, -0.1), nonuniform_axis(5, 0.8, 0.15)
    grid = p1.Grid(x, y)
    shape = grid.xx.shape
    out = {"x": x, "y": y, "hh": grid.hh, "kk": grid.kk,
           "cp": rng.random(shape) + 0.1, "T": rng.random(shape) * 100 + 273.15, "cl": rng.random(shape),
           "cd": rng.random(shape), "cs": rng.random(shape), "consts": consts_vec("nb_a")}
    out["T"][2, 3] = 0.0          # Dd := 0 where its temperature argument is 0 (model kind 1)
    for kind in (1, 2):
        S = p1.StateVars(out["cp"], out["T"], out["cl"], out["cd"], out["cs"], model=make_model("nb_a", kind),
                         hh=grid.hh, kk=grid.kk)
        for name in p1.StateVars._COMPUTED_PROPERTIES:
            out[f"kind{kind}/{name}"] = np.asarray(getattr(S, name))
    np.savez_compressed(os.path.join(HERE, "golden_statevars.npz"), **out)
    print("golden_statevars.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def gen_isolated():
    """Isolated-operator truncation errors and coupled-subsystem errors with degenerate model constants
    (tests/isolated_cases.py run with the reference's modules)."""
    out = {"operators": {w: isolated_cases.operator_truncation_errors(p1, w) for w in ("FT", "Fcl", "Fcd")},
           "coupled": {w: isolated_cases.coupled_errors(p1, mtu, w) for w in ("T_cl", "T_cl_cd")},
           "gradient_T_one_step": isolated_cases.gradient_T_one_step_errors(p1),
           "forward_euler_T_cl": isolated_cases.forward_euler_T_cl_errors(p1)}
    with open(os.path.join(HERE, "golden_isolated.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("golden_isolated.json written", out)


# ----------------------------------------------------------------------------------
def study_setups():
    return {
        "cstriple_pol": ("CsTriple", "pol", "nb_b"),
        "hcstriple_expsin": ("HCsTriple", "expsin", "nb_a"),
    }


STUDY_PARAMS = dict(Tf=0.02, N_base_spatial=4, num_spatial_refinements=4, dt_fixed_spatial=2.5e-3,
                    N_fixed_temporal=16, dt_base_temporal=1e-2, num_temporal_refinements=4)


def gen_studies():
    import contextlib
    import io

    import cvg_studies_base as cvg
    out = {"params": STUDY_PARAMS, "studies": {}}
    for label, (family, cname, mname) in study_setups().items():
        cfg = (getattr(p1, f"SemiDiscreteField_{family}"), MMS_CASES[cname][0], getattr(p1, f"ForcingTerms_{family}"),
               getattr(p1, f"P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_{family}"), label)
        with contextlib.redirect_stdout(io.StringIO()):
            rep = cvg.run_convergence_studies([cfg], dict(STUDY_PARAMS, model=make_model(mname, 2)))
        out["studies"][label] = {"family": family, "case": cname, "model_set": mname,
                                 "report": {k: {kk: [None if (isinstance(x, float) and x != x) else x for x in vv]
                                                for kk, vv in v.items()} for k, v in rep[label].items()}}
    with open(os.path.join(HERE, "golden_studies.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("golden_studies.json written")


# ----------------------------------------------------------------------------------
def gen_substeps():
    """The methods step() is made of, called one by one in step()'s order (src/prob1base.py:3117-3149)."""
    out = {}
    eta = 50.0
    kinds = {
        "reg": (p1.SemiDiscreteField_RegHCsTriple, p1.ForcingTerms_RegHCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple, {"regularization_factor": eta}),
        "cs3": (p1.SemiDiscreteField_CsTriple, p1.ForcingTerms_CsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple, {}),
        "hcs": (p1.SemiDiscreteField_HCsTriple, p1.ForcingTerms_HCsTriple,
                p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple, {}),
    }
    xu = np.linspace(0, 1, 9)
    setups = [("expsin_u8", "expsin", "nb_a", xu, xu.copy(), 0.1, (1.0 / 8) ** 1.5),
              ("pol_nu6x9", "pol", "stress", nonuniform_axis(6, 1.3, 0.2), nonuniform_axis(9, 0.8, -0.15), 0.0, 0.02)]
    for kname, (field_cls, forcing_cls, integ_cls, common) in kinds.items():
        for sname, cname, mname, x, y, t0, dt in setups:
            tag = f"{kname}/{sname}"
            grid = p1.Grid(x, y)
            model = make_model(mname, 2)
            case = MMS_CASES[cname][0](grid=grid, model=model)
            forcing = forcing_cls(mms_case=case, model=model, **common)
            field = field_cls(grid=grid, model=model, forcing_terms=forcing, **common)
            integ = integ_cls(semi_discrete_field=field, **common)
            S0 = p1.state_from_mms_when(mms_case=case, t=t0, grid=grid)
            out[f"{tag}/x"], out[f"{tag}/y"] = x, y
            out[f"{tag}/consts"] = consts_vec(mname)
            out[f"{tag}/model_kind"] = np.array(2)
            out[f"{tag}/eta"] = np.array(eta if kname == "reg" else 0.0)
            out[f"{tag}/case"] = np.array(cname)
            out[f"{tag}/t0"], out[f"{tag}/dt"] = np.array(t0), np.array(dt)
            for v in ["cp", "T", "cl", "cd", "cs"]:
                out[f"{tag}/s0/{v}"] = getattr(S0, v)
            Y = {"T": dt * field.FT(S0, t0) + 2 * S0.T, "cl": dt * field.Fcl(S0, t0) + 2 * S0.cl,
                 "cd": dt * field.Fcd(S0, t0) + 2 * S0.cd}
            for v, a in Y.items():
                out[f"{tag}/Y{v}"] = a
            cp1 = integ.initial_cp_pred(S0, t0, dt=dt)
            cs1 = integ.initial_cs_pred(S0, t0, dt=dt)
            out[f"{tag}/cp_pred"], out[f"{tag}/cs_pred"] = cp1, cs1
            lag = S0.with_changes(cp=cp1, T=S0.T, cl=S0.cl, cd=S0.cd, cs=cs1)
            T1 = integ.newton_step_T(lag, t0=t0, dt=dt, YT0=Y["T"])
            out[f"{tag}/T1"], out[f"{tag}/res_T"] = T1, integ.last_residual["T"]
            cl1 = integ.newton_step_cl(lag, T1, t0=t0, dt=dt, Ycl0=Y["cl"])
            out[f"{tag}/cl1"], out[f"{tag}/res_cl"] = cl1, integ.last_residual["cl"]
            cd1 = integ.newton_step_cd(lag, T1, cl1, t0=t0, dt=dt, Ycd0=Y["cd"])
            out[f"{tag}/cd1"], out[f"{tag}/res_cd"] = cd1, integ.last_residual["cd"]
            out[f"{tag}/cp_corr"] = integ.corrector_cp_step(T1, cl1, cd1, at_t0=S0, t0=t0, dt=dt)
            out[f"{tag}/cs_corr"] = integ.corrector_cs_step(T1, cl1, cd1, at_t0=S0, t0=t0, dt=dt)
            full = integ.step(S0, t0=t0, dt=dt)
            for v in ["cp", "T", "cl", "cd", "cs"]:  # the pieces really are the step
                assert np.array_equal(getattr(full, v), out[f"{tag}/" + {"cp": "cp_corr", "T": "T1", "cl": "cl1",
                                                                       "cd": "cd1", "cs": "cs_corr"}[v]])
    np.savez_compressed(os.path.join(HERE, "golden_substeps.npz"), **out)
    print("golden_substeps.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------
def gen_scalars(max_n_short=128, max_n_long=32):
    """MMSTrial.run_for_errors known answers on the notebooks' spatial-study schedule."""
    PC_REG = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple
    eta = 50.0
    res = {}
    sched = {"expsin": (0.01, max_n_short), "pol": (0.01, max_n_short),
             "peaks_fast1e1": (1.0, max_n_long), "nfs_h2_h3": (1.0, max_n_long)}
    for cname, (cls, mname) in MMS_CASES.items():
        Tf, nmax = sched[cname]
        res[cname] = {"Tf": Tf, "eta": eta, "model_set": mname, "levels": []}
        N = 2
        while N <= nmax:
            grid = p1.make_uniform_grid(N=N, M=N)
            trial = _trial(cls, make_model(mname, 2), grid, PC_REG, "reg", eta,
                           {"regularization_factor": eta})
            t0 = time.time()
            summ = trial.run_for_errors(Tf=Tf, dt=(1.0 / N) ** 1.5)
            wall = time.time() - t0
            res[cname]["levels"].append({
                "N": N, "dt_used": summ.dt_used, "overall": summ.overall_combined_error,
                "per_variable": summ.per_variable_sup_errors, "wall_s": wall})
            print(cname, N, summ.overall_combined_error, f"{wall:.1f}s", flush=True)
            N *= 2
    # notebooks' temporal study for ExpSin (N=32, Tf=0.01)
    res["expsin_temporal_N32"] = []
    for i in range(6):
        dt = 1e-2 / 2 ** i
        grid = p1.make_uniform_grid(N=32, M=32)
        trial = _trial(p1mc.MMSCaseExpSin, make_model("nb_a", 2), grid, PC_REG, "reg", eta,
                       {"regularization_factor": eta})
        summ = trial.run_for_errors(Tf=0.01, dt=dt)
        res["expsin_temporal_N32"].append({"dt": dt, "overall": summ.overall_combined_error})
    with open(os.path.join(HERE, "golden_scalars.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("golden_scalars.json written")


def gen_published():
    """Numbers the reference's executed notebooks print (12 significant digits)."""
    res = {}
    nbs = {
        "expsin": "MMSCaseExpSin_reghcstriple_convergence_study.ipynb",
        "pol": "MMSCasePol_reghcstriple_convergence_study.ipynb",
        "peaks_fast1e1": "MMSCaseSlowlyChangingPeaks_Fast1e1_reghcstriple_convergence_study.ipynb",
        "nfs_h2_h3": "MMSCaseNonFullySmoothPol_cpcsH2_TclcdH3_reghcstriple_convergence_study.ipynb",
        "nfs_h1_h2": "MMSCaseNonFullySmoothPol_cpcsH1_TclcdH2_reghcstriple_convergence_study.ipynb",
        "nfs_h2_h2": "MMSCaseNonFullySmoothPol_cpcsH2_TclcdH2_reghcstriple_convergence_study.ipynb",
    }
    for cname, fn in nbs.items():
        nb = json.load(open(os.path.join(REF_ROOT, fn)))
        for cell in nb["cells"]:
            if cell["cell_type"] != "code":
                continue
            src = "".join(cell["source"])
            if "Spatial Convergence Study" not in src:
                continue
            text = "".join("".join(o.get("text", "")) for o in cell.get("outputs", [])
                           if o.get("output_type") == "stream")
            blocks = re.split(r"Running simulation with grid size ", text)[1:]
            levels = []
            for b in blocks:
                n = int(b.split("x")[0])
                overall = float(re.search(r"Overall error: ([0-9.e+-]+)", b).group(1))
                wall = float(re.search(r"Completed in ([0-9.e+-]+) seconds", b).group(1))
                per = {m.group(1): float(m.group(2))
                       for m in re.finditer(r"^\s+(cp|T|cl|cd|cs): ([0-9.e+-]+)", b, re.M)}
                levels.append({"N": n, "overall": overall, "per_variable": per, "wall_s": wall})
            rates = [float(m.group(1)) for m in
                     re.finditer(r"Levels \d+,\d+,\d+: .* = ([0-9.]+)", text)]
            tf = float(re.search(r"^Tf = ([0-9.e+-]+)", src, re.M).group(1))
            res[cname] = {"notebook": fn, "Tf": tf, "levels": levels, "rates_3pt": rates}
        # the temporal study: fixed mesh, dt halved, overall error per level
        for cell in nb["cells"]:
            src = "".join(cell["source"]) if cell["cell_type"] == "code" else ""
            if "Temporal Convergence Study" not in src:
                continue
            text = "".join("".join(o.get("text", "")) for o in cell.get("outputs", [])
                           if o.get("output_type") == "stream")
            ns = {}
            for name in ("FIXED_GRID_SIZE", "BASE_DT", "Tf"):
                expr = re.search(rf"^{name} = ([^#\n]+)", src, re.M).group(1)
                assert re.fullmatch(r"[0-9.e+\-* ]+", expr.strip()), expr     # plain arithmetic only
                ns[name] = eval(expr, {"__builtins__": {}})
            # (the Peaks notebook refines the mesh with the step, N = ceil(1 / (2 dt)), and prints N per level)
            rows = [dict({"dt": float(m.group(1)), "overall": float(m.group(3))}, **({"N": int(m.group(2))} if m.group(2) else {}))
                    for m in re.finditer(r"Running simulation with dt = ([0-9.e+-]+)\n(?:(\d+)\n)?\s+Completed in [0-9.e+-]+ seconds"
                                         r"\n\s+Overall error: ([0-9.e+-]+)", text)]
            trates = [float(m.group(1)) for m in re.finditer(r"Levels \d+,\d+,\d+: .* = ([0-9.]+)", text)]
            res[cname]["temporal"] = {"N": int(ns["FIXED_GRID_SIZE"]), "Tf": float(ns["Tf"]), "base_dt": float(ns["BASE_DT"]),
                                      "levels": rows, "rates_3pt": trates}
    with open(os.path.join(HERE, "published_notebook_errors.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("published_notebook_errors.json written")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scalars", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    np.seterr(all="ignore")
    todo = a.only.split(",") if a.only else ["ops", "mms", "steps", "hcs", "substeps", "symbolic", "studies", "isolated", "statevars", "reghcs", "published"]
    if "ops" in todo:
        gen_ops()
    if "mms" in todo:
        gen_mms()
    if "steps" in todo:
        gen_steps()
    if "hcs" in todo:
        gen_hcs()
    if "substeps" in todo:
        gen_substeps()
    if "symbolic" in todo:
        gen_symbolic()
    if "studies" in todo:
        gen_studies()
    if "isolated" in todo:
        gen_isolated()
    if "statevars" in todo:
        gen_statevars()
    if "reghcs" in todo:
        gen_reghcs()
    if "published" in todo:
        gen_published()
    if a.scalars:
        gen_scalars()
