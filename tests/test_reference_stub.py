"""The reference-side binding (tedd/reference_stub.py: ctypes and NumPy only, the listing a maintainer of the
reference copies) over the host-compiled library.  tests/golden/make_integration_pin.py ran it inside the
reference's own MMSTrial, with the reference's classes, against the reference's integrator and stored the
reference's numbers; here the same trial is rebuilt from this package's mirror of those classes (the reference
does not travel) and must reproduce the reference's error summary."""
import json
import os

import numpy as np
import pytest

from test_twin_cabi_parity import TWIN, build_twin
from tedd import _ffi, reference_stub
from tedd import mms_trial_utils as mtu
from tedd import prob1_mms_cases as p1mc
from tedd import prob1base as p1

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = json.load(open(os.path.join(HERE, "golden", "golden_integration_pin.json")))


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)       # (the mirror's field and forcing-term classes evaluate through it)
    yield


def _trial(integrator_cls, case_cls=p1mc.MMSCaseExpSin):
    consts = json.load(open(os.path.join(HERE, "golden", "model_sets.json")))[PIN["model_set"]]
    model = p1.DefaultModel02(p1.ModelConsts(**consts))
    grid = p1.make_uniform_grid(N=PIN["N"], M=PIN["N"])
    common = {"regularization_factor": PIN["eta"]}
    return mtu.MMSTrial(grid=grid, model=model, mms_case_cls=case_cls,
                        field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                        integrator_cls=integrator_cls, integrator_params=dict(common),
                        forcing_terms_params=common, field_params=common)


@pytest.mark.parametrize("mms_case", [reference_stub.MMS_STREAM, reference_stub.MMS_EXPSIN], ids=["streamed", "closed-form"])
def test_binding_reproduces_the_reference_trial(mms_case):
    cls = reference_stub.make_integrator_class(TWIN, mms_case=mms_case)
    tr = _trial(cls)
    assert type(tr.integrator) is cls
    s = tr.run_for_errors(Tf=PIN["Tf"], dt=(1.0 / PIN["N"]) ** 1.5)
    ref = PIN["reference"]
    assert s.dt_used == ref["dt_used"]
    assert abs(s.overall_combined_error - ref["overall"]) <= 1e-9 * ref["overall"]
    for v, e in ref["per_variable"].items():
        # (the mirror's exact solution differs from the reference's lambdified one in the last bit of O(1)
        # values, which is all that is left of an error of 3e-11)
        assert abs(s.per_variable_sup_errors[v] - e) <= 1e-9 * e + 1e-15, v
    out = tr.integrator.step(tr.initial_state, t0=0.0, dt=(1.0 / PIN["N"]) ** 1.5)
    sums = [float(np.abs(getattr(out, v)).sum()) for v in ("cp", "T", "cl", "cd", "cs")]
    np.testing.assert_allclose(sums, PIN["one_step_state_checksums"], rtol=1e-13)


def test_binding_reports_library_errors():
    cls = reference_stub.make_integrator_class(TWIN)
    tr = _trial(cls)
    tr.integrator._num_newton_iterations = -3          # a negative iteration count is the library's TEDD_ERR_INVALID
    with pytest.raises(ValueError):
        tr.integrator.step(tr.initial_state, t0=0.0, dt=1e-3)
    tr.integrator._num_newton_iterations = 5
    with pytest.raises(AssertionError):               # the reference asserts dt > 0 (src/prob1base.py:3118)
        tr.integrator.step(tr.initial_state, t0=0.0, dt=0.0)
    rc = {reference_stub.ERR_INVALID: ValueError, reference_stub.ERR_ASSERT: AssertionError}
    assert rc.get(reference_stub.ERR_RUNTIME, RuntimeError) is RuntimeError


def _residual_trial(cls, num_newton_steps):
    res = PIN["newton_residuals"]
    consts = json.load(open(os.path.join(HERE, "golden", "model_sets.json")))[PIN["model_set"]]
    model = p1.DefaultModel02(p1.ModelConsts(**consts))
    grid = p1.make_uniform_grid(N=res["N"], M=res["N"])
    common = {"regularization_factor": PIN["eta"]}
    return mtu.MMSTrial(grid=grid, model=model, mms_case_cls=p1mc.MMSCaseExpSin,
                        field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                        integrator_cls=cls, integrator_params=dict(common, num_newton_steps=num_newton_steps),
                        forcing_terms_params=common, field_params=common)


def check_last_residual(library):
    """The side channel of src/prob1base.py:2942-2943: the reference's own residual check
    (tests/test_newton_residuals.py:56-124) through the binding, and the residual arrays after ONE Newton step
    (where they are not zero) against what the reference's integrator left there."""
    res = PIN["newton_residuals"]
    for mms_case in (reference_stub.MMS_STREAM, reference_stub.MMS_EXPSIN):
        cls = reference_stub.make_integrator_class(library, mms_case=mms_case)
        for nsteps in (1, 20):
            tr = _residual_trial(cls, res["num_newton_steps"])
            assert len(tr.integrator.last_residual) == 0            # empty before the first step, as the reference's
            state, t = tr.initial_state, 0.0
            for _ in range(nsteps):
                state = tr.integrator.step(at_t0=state, t0=t, dt=res["dt"])
                t += res["dt"]
                assert sorted(tr.integrator.last_residual) == ["T", "cd", "cl"]
                for v in ("T", "cl", "cd"):
                    assert tr.grid.norm_H(tr.integrator.last_residual[v]) < res["atol"]
        tr = _residual_trial(cls, 1)
        state = tr.integrator.step(at_t0=tr.initial_state, t0=0.0, dt=res["dt"])
        state = tr.integrator.step(at_t0=state, t0=res["dt"], dt=res["dt"])
        for v in ("T", "cl", "cd"):
            r = tr.integrator.last_residual[v]
            assert r.shape == (res["N"] + 1, res["N"] + 1)
            # entry by entry the arrays agreed with the reference's to 2e-16 where the fixture was made; here the
            # reference's arrays are represented by their largest entries and their sums
            assert abs(float(np.abs(r).max()) - res["one_newton_step_max_abs"][v]) <= 1e-15
            assert abs(float(np.abs(r[1:-1, 1:-1]).max()) - res["one_newton_step_interior_max_abs"][v]) <= 2e-15
            assert abs(float(np.abs(r).sum()) - res["one_newton_step_checksums"][v]) <= 1e-14


def check_model_changed_in_place(library):
    """tests/test_time_integrator_hcs_triple.py:377-387 changes integrator._model.Kd between calls: the streamed
    binding re-reads the model at every step (tedd_set_model) and reproduces the reference's states through the
    change and back; the closed-form binding refuses."""
    mk = PIN["mutated_Kd"]
    dt = (1.0 / PIN["N"]) ** 1.5
    assert mk["case"] == "MMSCasePol"      # (an exact solution that does not depend on the constants)
    tr = _trial(reference_stub.make_integrator_class(library), p1mc.MMSCasePol)
    state = tr.initial_state
    for k in range(3):
        kd = tr.integrator._model.Kd
        if k == mk["step"]:
            tr.integrator._model.Kd = mk["factor"] * kd
        state = tr.integrator.step(state, t0=k * dt, dt=dt)
        tr.integrator._model.Kd = kd
        sums = [float(np.abs(getattr(state, v)).sum()) for v in ("cp", "T", "cl", "cd", "cs")]
        np.testing.assert_allclose(sums, mk["state_checksums"][k], rtol=1e-13)
    assert mk["effect_on_cs"] > 1e-9       # (the change is far above that tolerance)
    tr = _trial(reference_stub.make_integrator_class(library, mms_case=reference_stub.MMS_EXPSIN))
    tr.integrator._model.Kd *= 2.0
    with pytest.raises(NotImplementedError):
        tr.integrator.step(tr.initial_state, t0=0.0, dt=dt)


def check_unconverged_solves_do_not_pass_silently(library):
    """spsolve (src/prob1base.py:2103, 2130) cannot stop short; Jacobi sweeps can.  With the sweeps capped below
    what the tolerance needs the binding raises instead of returning an under-solved state; a first choice of
    sweeps that misses the tolerance is redone from the same input with twice the sweeps."""
    import cabi_cases as cc
    consts = dict(zip(reference_stub.CONST_FIELDS, cc.STIFF))
    model = p1.DefaultModel02(p1.ModelConsts(**consts))
    grid = p1.make_uniform_grid(N=40, M=33)
    common = {"regularization_factor": 50.0}
    cls = reference_stub.make_integrator_class(library, mms_case=reference_stub.MMS_EXPSIN)
    tr = mtu.MMSTrial(grid=grid, model=model, mms_case_cls=p1mc.MMSCaseExpSin,
                      field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                      integrator_cls=cls, integrator_params=dict(common), forcing_terms_params=common, field_params=common)
    tr.integrator.jacobi_max = 2
    with pytest.raises(RuntimeError, match="tolerance|jacobi_tol"):
        tr.integrator.step(tr.initial_state, t0=0.0, dt=1e-2)
    tr.integrator.jacobi_max = 100000
    capped = tr.integrator.step(tr.initial_state, t0=0.0, dt=1e-2)
    # the mirror class over the same library (its own redo loop) gives the same state
    mirror = _trial_like(tr, p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple)
    ref = mirror.integrator.step(mirror.initial_state, t0=0.0, dt=1e-2)
    for v in ("cp", "T", "cl", "cd", "cs"):
        np.testing.assert_allclose(getattr(capped, v), getattr(ref, v), rtol=0, atol=1e-12)


def _trial_like(tr, integrator_cls):
    common = {"regularization_factor": 50.0}
    return mtu.MMSTrial(grid=tr.grid, model=tr.model, mms_case_cls=p1mc.MMSCaseExpSin,
                        field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                        integrator_cls=integrator_cls, integrator_params=dict(common), forcing_terms_params=common,
                        field_params=common)


def test_last_residual_side_channel():
    check_last_residual(TWIN)


def test_model_constants_are_read_at_every_step():
    check_model_changed_in_place(TWIN)


def test_unconverged_solves_do_not_pass_silently():
    check_unconverged_solves_do_not_pass_silently(TWIN)


def test_listing_in_the_integration_notes_is_the_tested_file():
    """The integration guide at the repository root prints the binding; what it prints is the file these tests run."""
    root = os.path.dirname(HERE)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    src = open(reference_stub.__file__).read()
    body = src[src.index("import ctypes as C"):]
    assert body.strip() in doc
