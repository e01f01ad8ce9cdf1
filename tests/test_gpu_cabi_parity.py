"""GPU parity tests: the HIP path, called through the C ABI, against vectors produced by the
reference (tests/golden) -- the same fixtures that pin the CPU oracle.

Tolerances (float64, stated per check): the device code evaluates the same formulas as the
reference with fused multiply-adds, reciprocal metric tables and its own exp(); the Newton
systems are solved by Jacobi sweeps to a relative update of 1e-13 instead of SuperLU."""
import numpy as np
import pytest

import cabi_cases as cc
from tedd import _ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    from tedd import _ffi

    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


@pytest.mark.parametrize("tag", cc.tags("golden_ops.npz", "/consts"))
def test_field_and_jacobian_stencils(tag):
    assert cc.check_ops(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_mms.npz", "/consts"))
def test_mms_exact_and_forcing(tag):
    e_exact, e_forcing = cc.check_mms(tag)
    assert e_exact < 1e-13
    assert e_forcing < 1e-13   # relative to the O(1) exact fields the forcing is a residual of


# solver variants: the register-resident solve chain (the default) with its automatic depth and
# forced to 1, 2 and 3 sweeps per launch (first / middle / last launches all exercised); one launch
# per Jacobi sweep; forcing evaluated in the step kernels instead of once per time level; forcing
# arrays made by a launch of their own instead of by the predictor launch
SOLVERS = {"strip": {"strip_solve": 1}, "strip_large": {"strip_shape": 0}, "strip_small": {"strip_shape": 1},
           "strip_small_h3": {"strip_shape": 1, "strip_hmax": 3}, "strip_mid": {"strip_shape": 2}, "strip_two_rows": {"strip_shape": 3},
           "strip_eight_rows": {"strip_shape": 4}, "strip_eight_rows_h3": {"strip_shape": 4, "strip_hmax": 3},
           "strip_two_rows_h3": {"strip_shape": 3, "strip_hmax": 3}, "strip_h1": {"strip_solve": 1, "strip_hmax": 1},
           "strip_h2": {"strip_solve": 1, "strip_hmax": 2}, "strip_h3": {"strip_solve": 1, "strip_hmax": 3},
           "sweeps": {"strip_solve": 0}, "forcing_in_kernels": {"cache_forcing": 0},
           "forcing_launch": {"fuse_forcing": 0}, "rhs_launch": {"fuse_rhs": 0}, "corrector_launch": {"fuse_corrector": 0}, "march": {"march": 1, "fuse_forcing": 0},
           "sweeps_forcing_in_kernels": {"strip_solve": 0, "cache_forcing": 0}}


@pytest.mark.parametrize("solver", list(SOLVERS))
@pytest.mark.parametrize("tag", cc.tags("golden_steps.npz", "/t0"))
def test_integrator_steps(tag, solver):
    if not tag.startswith("pc") and solver != "strip":
        pytest.skip("forward Euler has no linear solve")
    err, res_err, stats = cc.run_steps(tag, options=SOLVERS[solver])
    # state after 1, 2 and 5 steps, relative to the largest field magnitude
    assert err < (2e-12 if "bigdt" in tag else 5e-14), (tag, err, stats)
    # Newton residuals: absolute (they are differences of O(1) quantities)
    assert res_err < 1e-13, (tag, res_err)


# ---- HCsTriple (sharp Heaviside) field, forcing terms and predictor-corrector integrator
@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/hcs/Fcp"))
def test_hcs_field_and_jacobian_stencils(tag):
    assert cc.check_ops_hcs(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/t0/hcs/fcp"))
def test_hcs_forcing(tag):
    assert cc.check_mms_hcs(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/t0"))
def test_hcs_integrator_steps(tag):
    err, res_err, stats = cc.run_steps_hcs(tag)
    assert err < 5e-14, (tag, err, stats)
    assert res_err < 1e-13, (tag, res_err)


def test_hcs_corrector_reports_nonpositive_denominator():
    cc.check_hcs_denominator_error()


def test_prefetch_forcing_changes_nothing():
    cc.check_prefetch_forcing()


# ---- the verdict on the Jacobi solves and their warm start (reference: direct solves, src/prob1base.py:2088-2130)
def test_too_few_sweeps_through_advance_is_an_error():
    cc.check_under_converged_advance_raises()


@pytest.mark.parametrize("solver", ["strip", "strip_small_h3", "strip_mid", "sweeps"])
def test_warm_start_needs_fewer_sweeps_for_the_same_tolerance(solver):
    cc.check_warm_start_same_answer_fewer_sweeps(SOLVERS[solver])


def test_fixed_sweeps_are_not_judged_and_not_warm_started():
    cc.check_fixed_sweeps_are_plain_jacobi()


def test_null_output_pointers_are_refused():
    cc.check_null_output_pointers_are_refused()


def test_failed_create_leaves_the_library_usable():
    cc.check_failed_create_releases_what_it_took()


def test_switching_streams_keeps_the_order_of_the_work():
    """Steps enqueued on one caller stream, then the stream is switched without a synchronisation in
    between: the steps on the new stream must see the results of the old one's."""
    import torch
    x = np.linspace(0, 1, 1025)
    ref = _ffi.Context(x, x, cc.STIFF, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
    ref.set_state_exact(0.0)
    for k in range(6):
        ref.step_pc(k * 1e-3, 1e-3, jacobi_sweeps=(6, 4, 3))
    want = ref.get_state()
    ref.close()
    ctx = _ffi.Context(x, x, cc.STIFF, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    for k in range(6):
        ctx.set_stream(streams[k % 3].cuda_stream)
        ctx.step_pc(k * 1e-3, 1e-3, jacobi_sweeps=(6, 4, 3))
    got = ctx.get_state()
    ctx.close()
    for a, b in zip(want, got):
        np.testing.assert_array_equal(a, b)


def test_calls_leave_the_callers_current_device_alone():
    import torch
    before = torch.cuda.current_device()
    ctx = _ffi.Context(np.linspace(0, 1, 17), np.linspace(0, 1, 17), cc.STIFF, model_kind=2,
                       reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN, device=0)
    ctx.set_state_exact(0.0)
    ctx.step_pc(0.0, 1e-3)
    ctx.close()
    assert torch.cuda.current_device() == before


# ---- exact solutions given as expression programs (TEDD_MMS_PROGRAM)
@pytest.mark.parametrize("react", ["reg", "cs3", "hcs"])
def test_mms_program_equals_closed_form(react):
    cc.check_mms_program(react)


# ---- the same programs compiled at run time (hiprtc) instead of interpreted
@pytest.mark.parametrize("which", ["expsin", "all_ops"])
@pytest.mark.parametrize("react", ["reg", "cs3", "hcs"])
def test_generated_program_kernels_equal_interpreter(react, which):
    cc.check_generated_kernels_equal_interpreter(react, which)


def test_residuals_of_a_step_that_did_not_store_its_Y_T():
    cc.check_residuals_survive_other_calls()


def test_generated_kernels_refuse_a_header_tree_that_is_not_the_librarys(tmp_path):
    cc.check_skewed_header_tree_is_refused(tmp_path)


# ---- the pieces of a predictor-corrector step called on their own
@pytest.mark.parametrize("streamed", [False, True], ids=["closed_form", "streamed"])
@pytest.mark.parametrize("tag", cc.tags("golden_substeps.npz", "/t0"))
def test_substeps(tag, streamed):
    err, res_err = cc.check_substeps(tag, streamed)
    assert err < 5e-14, (tag, err)
    assert res_err < 1e-13, (tag, res_err)
