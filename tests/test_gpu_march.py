"""The marching launcher of the stencil kernels (a wavefront walks down 64 columns, stencil rows in registers,
column neighbours by lane shifts) against the plain one-thread-per-node launcher: the same node bodies on
the same values, so every result must be bit-identical -- on meshes whose width is not a multiple of the
wave or of the workgroup, with one-column and one-row remainders, on segments shorter than a workgroup's
share, for every reaction kind and every register budget of the kernel.  (The plain launcher is what the
golden-vector tests pin.)"""
import numpy as np
import pytest

from tedd import _ffi

pytestmark = pytest.mark.gpu

STRESS = [8.3144621, 1.60217662e-19, 0.11, 0.23, 0.37, 0.19, 2e-2, 3e-2, 0.4, 0.03, 0.2, 1.5, 1e-2, 0.3, 50.0,
          5e-2, 300.0]


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


def _run(N, M, react, options, row_lo=0, row_hi=None, ghost=0):
    x = np.linspace(0, 1, N + 1) ** 1.07
    y = np.linspace(0, 1, M + 1) ** 0.93
    ctx = _ffi.Context(x, y, STRESS, model_kind=2, reaction_kind=react, eta=30.0, mms_case=_ffi.MMS_PEAKS,
                       mms_params=(1.0, 10.0), row_lo=row_lo, row_hi=row_hi, ghost=ghost)
    ctx.set_option("fuse_forcing", 0)      # (array-fed predictor: the variant that marches)
    for k, v in options.items():
        ctx.set_option(k, v)
    ctx.set_state_exact(0.0)
    ctx.step_pc(0.0, 2e-3, jacobi_sweeps=(5, 4, 3), cs_newton_iterations=3)
    if ghost == 0:
        ctx.step_pc(2e-3, 2e-3, jacobi_sweeps=(5, 4, 3), cs_newton_iterations=3)
    out = ctx.get_state()
    ctx.close()
    return out


@pytest.mark.parametrize("N,M", [(70, 300), (257, 64), (40, 520), (9, 255), (130, 256)])
@pytest.mark.parametrize("react", [_ffi.REACT_CSTRIPLE, _ffi.REACT_REGHCSTRIPLE, _ffi.REACT_HCSTRIPLE])
def test_marching_launcher_is_bit_identical(N, M, react):
    ref = _run(N, M, react, {"march": 0})
    for opts in ({"march": 1}, {"march": 1, "march_occ": 2, "march_rows": 8}, {"march": 1, "march_occ": 3, "march_rows": 16},
                 {"march": 1, "march_occ": 4, "march_rows": 5},
                 # the wait for the next row's loads left to the compiler instead of placed before the row's stores
                 {"march": 1, "march_split": 0}, {"march": 1, "march_occ": 2, "march_rows": 8, "march_split": 0}):
        got = _run(N, M, react, opts)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b, err_msg=str(opts))


def test_marching_launcher_on_a_slab_with_ghost_rows():
    """A slab in the middle of the mesh: the launch range starts and ends inside the local rows."""
    kw = dict(row_lo=40, row_hi=90, ghost=16)
    ref = _run(150, 200, _ffi.REACT_REGHCSTRIPLE, {"march": 0}, **kw)
    got = _run(150, 200, _ffi.REACT_REGHCSTRIPLE, {"march": 1, "march_rows": 16}, **kw)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a[16:-16], b[16:-16])


# ---- the cl / cd solve that evaluates its right-hand side itself against right-hand-side launch + solve launch
def _run_solve(N, M, react, options, steps=3, sweeps=None, consts=None):
    x = np.linspace(0, 1, N + 1) ** 1.07
    y = np.linspace(0, 1, M + 1) ** 0.93
    ctx = _ffi.Context(x, y, STRESS if consts is None else consts, model_kind=2, reaction_kind=react, eta=30.0, mms_case=_ffi.MMS_PEAKS,
                       mms_params=(1.0, 10.0))
    for k, v in options.items():
        ctx.set_option(k, v)
    ctx.set_state_exact(0.0)
    for k in range(steps):
        ctx.step_pc(k * 2e-3, 2e-3, jacobi_sweeps=sweeps, cs_newton_iterations=3)
    out = ctx.get_state()
    stats = ctx.get_stats()
    ctx.close()
    return out, stats


@pytest.mark.parametrize("shape", [1, 2], ids=["32-row tiles", "48-row tiles"])
@pytest.mark.parametrize("react", [_ffi.REACT_CSTRIPLE, _ffi.REACT_REGHCSTRIPLE, _ffi.REACT_HCSTRIPLE])
@pytest.mark.parametrize("sweeps", [(5, 4, 3), None], ids=["fixed sweeps", "library's sweeps (warm start)"])
def test_fused_right_hand_side_is_bit_identical(shape, react, sweeps):
    _fused_against_two_launches(shape, react, sweeps, STRESS if sweeps else MILD)


# notebook-like constants: the library's own sweep counts then are a handful (single-launch chains, which fuse)
MILD = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-4, 1e-2, 1e-3, 1e-2, 1.0, 1e-4, 1e-2, 50.0, 5e-2, 300.0]


def _fused_against_two_launches(shape, react, sweeps, consts):
    """... against the two-launch solve whose matrix rows are built from the same face values ("exact_faces":
    the right-hand-side kernel's arithmetic; by default the two-launch solve uses a cheaper form of the same
    diffusivities for its rows, which moves the result by rounding only)."""
    steps = 3 if sweeps else 7
    ref, st0 = _run_solve(131, 200, react, {"fuse_rhs": 0, "exact_faces": 1, "strip_shape": shape}, steps, sweeps, consts)
    got, st1 = _run_solve(131, 200, react, {"fuse_rhs": 1, "strip_shape": shape}, steps, sweeps, consts)
    plain, _ = _run_solve(131, 200, react, {"fuse_rhs": 0, "strip_shape": shape}, steps, sweeps, consts)
    assert st0["sweeps"] == st1["sweeps"] and st0["warm_started"] == st1["warm_started"]
    assert sweeps is not None or max(st1["sweeps"][1:]) <= 9      # (single-launch cl / cd chains: the fused path ran)
    for a, b, c in zip(ref, got, plain):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(c, b, rtol=0, atol=1e-15 * max(1.0, float(np.max(np.abs(b)))))


# ---- forcing arrays by column-walking threads against one thread per node
@pytest.mark.parametrize("case,params,t", [
    (_ffi.MMS_POL, None, 0.3), (_ffi.MMS_PEAKS, (1.0, 10.0), 0.3), (_ffi.MMS_NFS, (2.1, 3.1, 2.1, 3.1, 2.1, 1.0 / np.pi), 0.3),
    (_ffi.MMS_EXPSIN, None, 1e-3),    # cell average from the moment tables
    (_ffi.MMS_EXPSIN, None, 0.25),    # ... from the nine quadrature points
], ids=["pol", "peaks", "nfs", "expsin-moments", "expsin-points"])
@pytest.mark.parametrize("react", [_ffi.REACT_CSTRIPLE, _ffi.REACT_REGHCSTRIPLE])
def test_forcing_by_columns_is_bit_identical(case, params, t, react):
    """A mesh large enough for the column launch (2101 x 2051 nodes: ragged in both directions), every
    closed-form case, both evaluations of the ExpSin cell average."""
    x = np.linspace(0, 1, 2101) ** 1.07
    y = np.linspace(0, 1, 2051) ** 0.93
    out = []
    for columns in (0, 1):
        kw = {} if params is None else {"mms_params": params}
        ctx = _ffi.Context(x, y, STRESS, model_kind=2, reaction_kind=react, eta=30.0, mms_case=case, **kw)
        ctx.set_option("forcing_columns", columns)
        out.append(ctx.eval_forcing(t))
        ctx.close()
    for a, b in zip(*out):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


def test_expsin_cell_average_by_moments_matches_the_nine_point_sum():
    """The two evaluations of the ExpSin cp-forcing cell average agree to rounding where both apply: the host
    picks the moment form for |beta(t)| < 2^-10; with K1 = K2 and DT = 1/(2 pi^2) beta vanishes identically, and
    the nine-point sum is what the NumPy restatement computes."""
    import tedd_oracle as orc  # (tests/conftest.py puts oracle/ on the path)
    consts = list(STRESS)
    x = np.linspace(0, 1, 2101) ** 1.07
    y = np.linspace(0, 1, 2051) ** 0.93
    ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=30.0,
                       mms_case=_ffi.MMS_EXPSIN)
    got = ctx.eval_forcing(2e-3)[0]
    ctx.close()
    model = orc.Model(consts, kind=2)
    mesh = orc.Mesh(x, y)
    want = orc.forcing_arrays(orc.CaseExpSin(model), model, mesh, 2e-3, "reg", 30.0)["fcp"]
    assert np.abs(got - want).max() < 1e-15


# ---- order in which the solve launches' tiles are dealt out
@pytest.mark.parametrize("panel", [1, 3, 8, 1000])
def test_tile_order_of_the_solve_launches_changes_nothing(panel):
    """Panels of tile rows walked down the columns first (ragged last panel, panels taller than the mesh)
    against the row-major order: the same tiles, so the same bits."""
    ref, _ = _run_solve(300, 280, _ffi.REACT_REGHCSTRIPLE, {"strip_panel": 0}, steps=2, sweeps=(6, 4, 3))
    got, _ = _run_solve(300, 280, _ffi.REACT_REGHCSTRIPLE, {"strip_panel": panel}, steps=2, sweeps=(6, 4, 3))
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("N,M", [(32, 32), (131, 200), (256, 256), (70, 300)])
def test_corrector_carried_by_the_cd_solve_is_bit_identical(N, M):
    """On small slabs the last launch of the cd chain also carries the cp corrector and the record pass of the
    cs corrector (RegHCsTriple): the same functions on the same values as the corrector's own launch, so states,
    records and the iteration count agree bit for bit -- also with 2 Newton x 2 predictor-corrector iterations
    (only the last Newton step of the first carries it) and with the global early exit of the cs iteration."""
    from test_gpu_persistent import NB_A, _early_exit_run

    def run(opts, **kw):
        x = np.linspace(0, 1, N + 1) ** 1.04
        y = np.linspace(0, 1, M + 1) ** 0.97
        ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_state_exact(0.0)
        dt = (1.0 / max(N, M)) ** 1.5
        series = ctx.advance(0.0, dt, 5, **kw)
        st = ctx.get_stats()
        out = ctx.get_state()
        ctx.close()
        return series, out, st

    for kw in ({}, {"num_pc_steps": 2, "num_newton_steps": 2}, {"cs_newton_iterations": 8}, {"cs_newton_iterations": 1}):
        ref = run({"fuse_corrector": 0}, **kw)
        got = run({}, **kw)
        np.testing.assert_array_equal(ref[0], got[0])
        for a, b in zip(ref[1], got[1]):
            np.testing.assert_array_equal(a, b, err_msg=str(kw))
        assert ref[2]["cs_iterations"] == got[2]["cs_iterations"] and ref[2]["last_update"] == got[2]["last_update"]
    if (N, M) == (32, 32):
        kept_ref, ref = _early_exit_run({"fuse_corrector": 0})
        kept, got = _early_exit_run({})
        assert kept == kept_ref and all(1 <= k < 8 for k in kept)
        for sa, sb in zip(ref, got):
            for a, b in zip(sa, sb):
                np.testing.assert_array_equal(a, b)
