"""The bench workload at its full sizes on the GPU (ExpSin, RegHCsTriple, notebook constants, dt = h^1.5 on the
4096^2 and 16384^2 synthetic meshes), checked through properties that do not need the CPU oracle, which
cannot run at these sizes:

  * the run tracks the exact solution (error norms after the steps stay at the level the second-order
    scheme gives on such a mesh);
  * every Jacobi solve reached its tolerance (recorded last update vs increment), and the Newton
    residuals of the last step are small (the bound of reference tests/test_newton_residuals.py:116-124);
  * the boundary rules of the step hold on every boundary node (T, cp, cs are 0 there;
    reference src/prob1base.py:3041, :2996, :3645);
  * implementations that must give the same numbers do: the solve chain with its automatic tile
    shapes, with the large tiles only, and one launch per Jacobi sweep; forcing arrays per time level
    and forcing evaluated inside the step kernels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    from tedd import _ffi

    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


def _run(N, nsteps, options, want_state, M=None, stretch=1.0):
    from tedd import _ffi
    x = np.linspace(0, 1, N + 1) ** stretch
    y = x if M is None else np.linspace(0, 1, M + 1) ** (1.0 / stretch)
    ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    for k, v in options.items():
        ctx.set_option(k, v)
    dt, t = (1.0 / N) ** 1.5, 0.0
    ctx.set_state_exact(0.0)
    for _ in range(nsteps):
        ctx.step_pc(t, dt)
        t += dt
    stats = ctx.get_stats()
    ctx.check_converged()      # the library's own verdict on the sweep counts it chose
    norms = np.asarray(ctx.error_norms(t))
    out = {"stats": stats, "norms": norms, "t": t}
    if want_state:
        out["state"] = ctx.get_state()
        out["residuals"] = ctx.get_residuals()
    ctx.close()
    return out


def _check_properties(N, r):
    h = 1.0 / N
    # Jacobi: last update below 10 x tolerance x increment for T, cl, cd
    for v in range(3):
        upd, inc = r["stats"]["last_update"][v], r["stats"]["increment"][v]
        assert upd <= 1e-12 * inc or upd == 0.0, (N, v, upd, inc)
    # tracks the exact solution: ||e||_H of the five unknowns after a few steps of size h^1.5
    err_H = float(np.sqrt(np.sum(r["norms"][:5])))
    assert np.isfinite(r["norms"]).all() and err_H < 50 * h * h * r["t"] + 1e-13, (N, err_H)
    if "state" in r:
        cp, T, cl, cd, cs = r["state"]
        for name, a in (("T", T), ("cp", cp), ("cs", cs)):
            assert not a[0, :].any() and not a[-1, :].any() and not a[:, 0].any() and not a[:, -1].any(), name
        for a in r["state"]:
            assert np.isfinite(a).all()
        w = h * h  # uniform mesh: ||r||_H^2 = sum_interior r^2 h k
        for name, res in zip(("T", "cl", "cd"), r["residuals"]):
            assert np.sqrt(np.sum(res[1:-1, 1:-1] ** 2) * w) < 1e-9, name


def test_solver_variants_agree_4096():
    """16.8 M nodes: all variants, full states compared."""
    N = 4096
    ref = _run(N, 3, {}, True)
    _check_properties(N, ref)
    scale = [np.max(np.abs(a)) for a in ref["state"]]
    for name, opts in (("large tiles", {"strip_shape": 0}), ("sweeps", {"strip_solve": 0}),
                       ("forcing in kernels", {"cache_forcing": 0}), ("forcing by predictor", {"fuse_forcing": 1})):
        r = _run(N, 3, opts, True)
        for a, b, s in zip(r["state"], ref["state"], scale):
            # tile shapes change nothing; the per-sweep solver and the in-kernel forcing round differently
            tol = 0.0 if name == "large tiles" else 1e-14 * max(s, 1.0)
            assert np.max(np.abs(a - b)) <= tol, name
        np.testing.assert_allclose(r["norms"], ref["norms"], rtol=1e-6, atol=1e-30, err_msg=name)


def test_floor_of_the_stopping_rule_4096():
    """On this mesh a step moves T by 8e-8 of its size, so jacobi_tol x increment asks for updates of 1e-20 in a
    field whose rounding unit is 2e-16.  With the floor of the stopping rule (updates below 2^-62 max|u| count as
    converged, option jacobi_floor) the warm-started T solve takes fewer sweeps than with the plain rule; the
    library's verdict passes either way and the states differ by rounding only (the reference solves directly,
    src/prob1base.py:2103, 2130: both runs are within 1e-15 of what it would give)."""
    N, nsteps = 4096, 12
    with_floor = _run(N, nsteps, {}, True)
    plain = _run(N, nsteps, {"jacobi_floor": 0.0}, True)
    sf, sp = with_floor["stats"], plain["stats"]
    assert sf["warm_started"][0] == 1 and sp["warm_started"][0] == 1
    assert sf["sweeps"][0] < sp["sweeps"][0], (sf["sweeps"], sp["sweeps"])
    # the plain rule, met by the plain run and knowingly not by the other
    assert sp["last_update"][0] <= 1e-12 * sp["increment"][0]
    assert sf["last_update"][0] <= 10 * 2.0 ** -62 * 1.0          # max|T| <= 1 for this case
    assert max(sf["worst_ratio"]) <= 10.0 and max(sp["worst_ratio"]) <= 10.0
    for a, b in zip(with_floor["state"], plain["state"]):
        assert np.max(np.abs(a - b)) <= 4.5e-16
    # (the squared error norms are 1e-24 ... 1e-35 here, sums of squared rounding errors of the O(1) fields: the
    # two runs agree in them as far as such sums can, to a few parts in a million)
    np.testing.assert_allclose(with_floor["norms"], plain["norms"], rtol=1e-4, atol=1e-30)


def test_bench_mesh_16384():
    """268 M nodes (the mesh the headline number is measured on): three steps, then the properties; the
    large-tile chain must reproduce the automatic one bit for bit (compared through the error norms, whose
    deterministic sums see every node, and through the downloaded state of the default run)."""
    N = 16384
    ref = _run(N, 3, {}, True)
    _check_properties(N, ref)
    del ref["state"], ref["residuals"]
    r = _run(N, 3, {"strip_shape": 0}, False)
    _check_properties(N, r)
    np.testing.assert_array_equal(r["norms"], ref["norms"])


def test_consecutive_contexts_of_the_bench_mesh_stream_alike():
    """The arena of a 16384^2 context is given one of two kinds of physical memory in turn when contexts are made
    one after the other (13 reads + 7 writes stream at 6.5 or at 5.9 TB/s; the predictor takes 7.5 or 8.2 ms); the
    library streams through the arena when it takes it and takes turns until it holds the faster kind.  Three
    contexts in a row: the placement note says what was measured, and the predictor launches agree to 3 %."""
    from tedd import _ffi
    N = 16384
    x = np.linspace(0, 1, N + 1)
    dt = (1.0 / N) ** 1.5
    times = []
    for _ in range(3):
        ctx = _ffi.Context(x, x, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
        assert "one arena of 32 buffers" in ctx.placement and "streamed at" in ctx.placement, ctx.placement
        ctx.set_state_exact(0.0)
        ctx.advance(0.0, dt, 3, want_errors=False)
        ctx.profile_enable(True)
        for k in range(3):
            ctx.step_pc((3 + k) * dt, dt)
        ctx.sync()
        prof = ctx.profile_get()
        ctx.close()
        times.append(prof["predict"]["total_ms"] / prof["predict"]["launches"])
    assert max(times) <= 1.03 * min(times), times


def test_odd_nonuniform_mesh_variants_agree():
    """A 1001 x 778-node non-uniform mesh (no dimension a multiple of the tile sizes): every tile shape and
    chain depth of the solve, the per-sweep solver and both forcing modes give the same state."""
    N, M = 1000, 777
    ref = _run(N, 3, {}, True, M=M, stretch=1.15)
    for a in ref["state"]:
        assert np.isfinite(a).all()
    scale = [max(np.max(np.abs(a)), 1.0) for a in ref["state"]]
    exact = ("shape 0", "shape 1", "shape 2", "shape 1 depth 3", "shape 2 depth 2", "shape 3", "shape 3 depth 2", "shape 4",
             "shape 4 depth 3")
    for name, opts in (("shape 0", {"strip_shape": 0}), ("shape 1", {"strip_shape": 1}), ("shape 2", {"strip_shape": 2}),
                       ("shape 1 depth 3", {"strip_shape": 1, "strip_hmax": 3}),
                       ("shape 2 depth 2", {"strip_shape": 2, "strip_hmax": 2}),
                       ("shape 3", {"strip_shape": 3}), ("shape 3 depth 2", {"strip_shape": 3, "strip_hmax": 2}),
                       ("shape 4", {"strip_shape": 4}), ("shape 4 depth 3", {"strip_shape": 4, "strip_hmax": 3}),
                       ("sweeps", {"strip_solve": 0}), ("forcing in kernels", {"cache_forcing": 0}),
                       ("forcing launch", {"fuse_forcing": 0})):
        r = _run(N, 3, opts, True, M=M, stretch=1.15)
        for a, b, s in zip(r["state"], ref["state"], scale):
            assert np.max(np.abs(a - b)) <= (0.0 if name in exact else 1e-14 * s), name


@pytest.mark.parametrize("N,M", [(63, 64), (131, 257), (300, 301)])
@pytest.mark.parametrize("react", [0, 1, 2])
def test_tile_shapes_and_depths_bit_identical(N, M, react):
    """Random non-uniform meshes whose sizes straddle the tile edges, large steps (increments 1e-3 of the
    fields, so that a last-bit difference of an increment shows in the unknown): every tile shape and chain
    depth of the solve gives the same bits, and the per-sweep solver the same values to 1e-14.  (This needs
    the build's -ffp-contract=on: with cross-statement contraction the instantiations of the kernel round
    the matrix rows differently.)"""
    from tedd import _ffi
    rng = np.random.default_rng(7 + N)
    x = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(N))])
    y = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(M))])
    x, y = x / x[-1], y / y[-1]

    def run(opts):
        ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=react, eta=50.0 if react == 1 else 0.0,
                           mms_case=_ffi.MMS_EXPSIN)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_state_exact(0.0)
        for k in range(2):
            ctx.step_pc(k * 2e-3, 2e-3, jacobi_sweeps=(23, 11, 7))
        out = ctx.get_state()
        ctx.close()
        return out

    ref = run({"strip_solve": 0})
    first = None
    for shape in (0, 1, 2, 3, 4):  # (3: two rows per thread, 16-row tiles, cl and cd; 4: eight rows per thread, 64-row tiles, T)
        for hmax in (1, 3, 10):
            out = run({"strip_shape": shape, "strip_hmax": hmax})
            first = out if first is None else first
            for a, b, c in zip(out, first, ref):
                np.testing.assert_array_equal(a, b, err_msg=f"shape {shape} depth {hmax}")
                assert np.max(np.abs(a - c)) <= 1e-14 * max(1.0, np.max(np.abs(c)))


def test_out_of_memory_is_an_error_and_leaves_the_library_usable():
    """A mesh whose fields cannot be allocated (100001^2 nodes: 80 GB per array) fails with RuntimeError at the
    first call that needs the arrays; contexts made before and after keep working."""
    from tedd import _ffi

    def small():
        x = np.linspace(0, 1, 33)
        return _ffi.Context(x, x, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                            mms_case=_ffi.MMS_EXPSIN)
    before = small()
    before.set_state_exact(0.0)
    x = np.linspace(0, 1, 100001)
    big = _ffi.Context(x, x, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    with pytest.raises(RuntimeError, match="memory|hipMalloc"):
        big.set_state_exact(0.0)
        big.step_pc(0.0, 1e-8)
    big.close()
    after = small()
    after.set_state_exact(0.0)
    for ctx in (before, after):
        ctx.step_pc(0.0, 1e-3)
        ctx.sync()
        assert np.sqrt(ctx.error_norms(1e-3)[:5].sum()) < 1e-4
        ctx.close()
