"""The persistent step kernel (option `persistent`: the launches of one or many predictor-corrector steps run
as one launch, with a workgroup barrier or a barrier over the grid between the phases) against the separate
launches of the per-sweep solver it records: the same node functors on the same values in the same order, so
states, error-norm series, convergence records and the cs corrector's iteration count must be bit-identical --
with one workgroup and with several, on meshes narrower and wider than a wave, with the reference's global
early exit of the cs Newton iteration firing (src/prob1base.py:3654-3663), and when more launches are recorded
than one persistent launch holds.  Configurations the kernel is not built for fall back to separate launches."""
import numpy as np
import pytest

from tedd import _ffi

pytestmark = pytest.mark.gpu

NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]
STRESS = [8.3144621, 1.60217662e-19, 0.11, 0.23, 0.37, 0.19, 2e-2, 3e-2, 0.4, 0.03, 0.2, 1.5, 1e-2, 0.3, 50.0,
          5e-2, 300.0]
PER_SWEEP = {"persistent": 0, "strip_solve": 0, "warm_start": 0}


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


def _run(N, M, options, *, consts=NB_A, case=_ffi.MMS_EXPSIN, params=(), react=_ffi.REACT_REGHCSTRIPLE, steps=6,
         dt=None, **step_kw):
    x = np.linspace(0, 1, N + 1) ** 1.05
    y = np.linspace(0, 1, M + 1) ** 0.95
    ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=react, eta=50.0, mms_case=case, mms_params=params)
    for k, v in options.items():
        ctx.set_option(k, v)
    dt = dt or (1.0 / max(N, M)) ** 1.5
    ctx.set_state_exact(0.0)
    series = ctx.advance(0.0, dt, steps, **step_kw)
    ctx.step_pc(steps * dt, dt, **step_kw)
    stats = ctx.get_stats()
    state = ctx.get_state()
    ctx.check_converged()
    ctx.close()
    return series, state, stats


def _same(ref, got, what):
    np.testing.assert_array_equal(ref[0], got[0], err_msg=f"{what}: error-norm series")
    for a, b in zip(ref[1], got[1]):
        np.testing.assert_array_equal(a, b, err_msg=f"{what}: state")
    for k in ("sweeps", "last_update", "increment", "cs_iterations"):
        assert list(np.ravel(ref[2][k])) == list(np.ravel(got[2][k])), (what, k, ref[2][k], got[2][k])


@pytest.mark.parametrize("N,M", [(4, 4), (12, 9), (32, 32), (20, 150), (97, 70)])
@pytest.mark.parametrize("groups", [1, 3, 16])
def test_persistent_kernel_equals_the_launches_it_records(N, M, groups):
    ref = _run(N, M, PER_SWEEP)
    got = _run(N, M, {"persistent": groups})
    _same(ref, got, f"{N}x{M}, {groups} workgroups")


@pytest.mark.parametrize("case,params", [(_ffi.MMS_POL, ()), (_ffi.MMS_PEAKS, (1.0, 10.0)), (_ffi.MMS_NFS, (2.1, 3.1))])
def test_persistent_kernel_other_cases(case, params):
    kw = dict(consts=STRESS, case=case, params=params, dt=2e-3)
    _same(_run(24, 31, PER_SWEEP, **kw), _run(24, 31, {"persistent": 2}, **kw), f"case {case}")


def _early_exit_run(options, N=47, M=21, nsteps=3):
    """Streamed (zero) forcing from a state whose cs is positive on every node, boundary included: min |x| of
    the scalar Newton iteration is then not the rounding noise of a zero boundary value and the reference's
    global test stops the iteration early.  Only the first step starts from such a state (the corrector masks
    the boundary of cs), so every step is taken from it."""
    i = np.arange(N + 1)[:, None] / N
    j = np.arange(M + 1)[None, :] / M
    state = [np.ascontiguousarray(a) for a in (0.3 + 0.2 * np.sin(2 * i + j), 0.5 + 0.1 * np.cos(i - 3 * j),
                                               0.2 * np.sin(3 * i * j), 0.4 + 0.1 * i * j, 0.02 + 0.9 * i ** 2 + 0.05 * j)]
    ctx = _ffi.Context(np.linspace(0, 1, N + 1), np.linspace(0, 1, M + 1), STRESS, model_kind=2,
                       reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=-1)
    for k, v in options.items():
        ctx.set_option(k, v)
    zero = np.zeros((N + 1, M + 1))
    kept, states = [], []
    for k in range(nsteps):
        ctx.set_state(*state)
        ctx.set_forcing(0, zero, zero, zero, zero, zero)
        ctx.set_forcing(1, zero, zero, zero, zero, zero)
        ctx.step_pc(0.0, 0.02 * (k + 1), cs_newton_iterations=8, cs_consec_xs_rtol=2e-3, jacobi_sweeps=[4, 3, 2])
        kept.append(ctx.get_stats()["cs_iterations"])
        states.append(ctx.get_state())
    ctx.close()
    return kept, states


def test_persistent_kernel_global_early_exit_of_the_cs_iteration():
    """The reference's global test stops the scalar Newton iteration early: the decide phase must iterate
    again, fewer times, exactly as the second pass of the separate launches does."""
    kept_ref, ref = _early_exit_run(PER_SWEEP)
    assert all(1 <= k < 8 for k in kept_ref), kept_ref
    for groups in (1, 5):
        kept, got = _early_exit_run({"persistent": groups})
        assert kept == kept_ref
        for sa, sb in zip(ref, got):
            for a, b in zip(sa, sb):
                np.testing.assert_array_equal(a, b, err_msg=f"early exit, {groups} workgroups")


def test_persistent_kernel_many_steps_span_several_launches():
    """More recorded launches than one persistent launch holds (1024 descriptors): the list is sent in pieces."""
    ref = _run(10, 10, PER_SWEEP, steps=150)
    _same(ref, _run(10, 10, {"persistent": 1}, steps=150), "150 steps")


def test_persistent_option_falls_back_where_the_kernel_does_not_apply():
    """CsTriple (closed-form cs corrector) and more than 8 cs Newton iterations are not recorded: the option
    must leave the results of the separate launches untouched."""
    kw = dict(react=_ffi.REACT_CSTRIPLE)
    _same(_run(16, 16, {"persistent": 0}, **kw), _run(16, 16, {"persistent": 4}, **kw), "CsTriple")
    kw = dict(cs_newton_iterations=12)
    _same(_run(16, 16, {"persistent": 0}, **kw), _run(16, 16, {"persistent": 4}, **kw), "12 iterations")
