"""Row-slab decomposition + ghost-row exchange, world_size 2 and 3 over gloo on CPU.

Each rank drives the host twin of the device library on its slab (same context code, ghost-row
bookkeeping and C ABI as the HIP library) and exchanges ghost rows with tedd.distributed.  The
decomposed run must reproduce the single-slab run bit for bit on every owned row, because a
slab performs exactly the arithmetic of the corresponding rows of the whole mesh."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, HERE)

from tedd.distributed import HaloExchange, SlabStepper, partition_rows  # noqa: E402

NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]
STRESS = [8.3144621, 1.60217662e-19, 0.11, 0.23, 0.37, 0.19, 2e-2, 3e-2, 0.4, 0.03, 0.2, 1.5, 1e-2, 0.3, 50.0,
          5e-2, 300.0]


def _free_port():
    """A TCP port nobody listens on right now (for the rendezvous of the ranks a test starts)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_partition_covers_all_rows():
    for n_rows in (9, 33, 257, 16385):
        for world in (1, 2, 3, 8):
            if n_rows // world < 2:
                continue
            slabs = [partition_rows(n_rows, world, r, 2) for r in range(world)]
            assert slabs[0].row_lo == 0 and slabs[-1].row_hi == n_rows
            for a, b in zip(slabs[:-1], slabs[1:]):
                assert a.row_hi == b.row_lo
            sizes = [s.row_hi - s.row_lo for s in slabs]
            assert max(sizes) - min(sizes) <= 1
            assert slabs[0].ghost_lo == 0 and slabs[-1].ghost_hi == 0
            if world > 1:
                assert slabs[0].ghost_hi == 2 and slabs[-1].ghost_lo == 2
    with pytest.raises(ValueError):
        partition_rows(9, 4, 0, 3)   # slabs of 2 rows cannot feed 3 ghost rows
    with pytest.raises(ValueError):
        partition_rows(9, 2, 0, 0)


def _worker(rank, world, port, case, N, M, nsteps, kind, consts, sweeps, ghost, out_dir, options, step_kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_twin_cabi_parity import TWIN
        from tedd import _ffi
        _ffi.use_library(TWIN)
        x = np.linspace(0, 1, N + 1) ** 1.1
        y = np.linspace(0, 1, M + 1)
        slab = partition_rows(N + 1, world, rank, ghost)
        ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                           mms_case=case, row_lo=slab.row_lo, row_hi=slab.row_hi, ghost=slab.ghost)
        assert ctx.nrows == slab.local_rows and ctx.row_offset == slab.row_offset
        options = dict(options)
        overlap = bool(options.pop("_overlap", False))   # (not a library option: the stepper's mode)
        for name, value in options.items():
            ctx.set_option(name, value)
        stepper = SlabStepper(ctx, slab, torch.device("cpu"), overlap=overlap)
        ctx.set_state_exact(0.0)
        dt, t = 0.01, 0.0
        norms = []
        for k in range(nsteps):
            if kind == "pc":
                stepper.step_pc(t, dt, jacobi_sweeps=sweeps, **(step_kw or {}))
            else:
                stepper.step_feuler(t, dt)
            t += dt
            if overlap and k < nsteps - 1:
                # the ghost rows stay in flight until the next step asks for them; anything else is refused
                with pytest.raises(ValueError, match="in flight"):
                    ctx.error_norms(t)
                norms.append(np.full(8, np.nan))
                continue
            stepper.finish()
            local = torch.from_numpy(ctx.error_norms(t))
            dist.all_reduce(local)
            norms.append(local.numpy().copy())
        stepper.check_converged()
        state = ctx.get_state()
        a, b = slab.owned_local
        st = ctx.get_stats()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=slab.row_lo, hi=slab.row_hi,
                 norms=np.array(norms), cs_iterations=st["cs_iterations"], sweeps=np.array(st["sweeps"]),
                 warm=np.array(st["warm_started"]), increment=np.array(st["increment"]),
                 first_update=np.array(st["first_update"]),
                 **{f"f{k}": s[a:b] for k, s in enumerate(state)})
        ctx.close()
    finally:
        dist.destroy_process_group()


def _single(case, N, M, nsteps, kind, consts, sweeps, step_kw=None):
    from test_twin_cabi_parity import TWIN, build_twin
    from tedd import _ffi
    build_twin()
    _ffi.use_library(TWIN)
    x = np.linspace(0, 1, N + 1) ** 1.1
    y = np.linspace(0, 1, M + 1)
    ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=case)
    ctx.set_state_exact(0.0)
    dt, t = 0.01, 0.0
    norms = []
    for _ in range(nsteps):
        if kind == "pc":
            ctx.step_pc(t, dt, jacobi_sweeps=sweeps, **(step_kw or {}))
        else:
            ctx.step_feuler(t, dt)
        t += dt
        norms.append(ctx.error_norms(t))
    out = ctx.get_state()
    st = ctx.get_stats()
    _single.cs_iterations = st["cs_iterations"]
    _single.sweeps, _single.warm = list(st["sweeps"]), list(st["warm_started"])
    _single.increment, _single.first_update = list(st["increment"]), list(st["first_update"])
    ctx.close()
    return out, np.array(norms)


# "forcing_launch": the forcing arrays of a time level come from a launch of their own, which the slab
# stepper enqueues ahead of time while the ghost rows travel (as on slabs too large for the predictor
# launch to make them)
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind,case,consts,options", [("pc", 1, NB_A, {}), ("pc", 0, STRESS, {}), ("fe", 2, NB_A, {}),
                                                      ("pc", 1, NB_A, {"fuse_forcing": 0}),
                                                      ("pc", 1, NB_A, {"_overlap": True}),
                                                      ("pc", 0, STRESS, {"_overlap": True, "fuse_forcing": 0})],
                         ids=["pc-expsin", "pc-pol-stress", "fe-peaks", "pc-expsin-forcing_launch",
                              "pc-expsin-overlap", "pc-pol-stress-overlap-forcing_launch"])
def test_slab_run_equals_single_domain(world, kind, case, consts, options, tmp_path):
    N, M, nsteps = 47, 21, 4
    sweeps = [4, 3, 2]
    ghost = sum(sweeps) + 2 if kind == "pc" else 1
    ref_state, ref_norms = _single(case, N, M, nsteps, kind, consts, sweeps)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, N, M, nsteps, kind, consts, sweeps, ghost, str(tmp_path), options),
             nprocs=world, join=True)
    rows = 0
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        for k in range(5):
            np.testing.assert_array_equal(z[f"f{k}"], ref_state[k][lo:hi])   # bit for bit
        # the global norms are sums over ranks: equal up to the summation order (overlap mode takes them after
        # the last step only: the ghost rows are in flight in between)
        took = ~np.isnan(z["norms"][:, 0])
        assert took[-1]
        np.testing.assert_allclose(z["norms"][took], ref_norms[took], rtol=1e-12, atol=1e-300)
        rows += hi - lo
    assert rows == N + 1


# (on this mesh the T system needs ~25 sweeps from zero: the warm start pays, and the library changes its sweep
# counts in the course of the run from the convergence records)
WARM_T = list(NB_A)


@pytest.mark.parametrize("world", [2, 3])
def test_slab_run_with_library_chosen_sweeps_equals_single_domain(world, tmp_path):
    """Sweep counts and warm starts left to the library (jacobi_sweeps = None) over 12 steps: the slabs share the
    convergence record of every step (one MAX all-reduce), take the decisions of the single-domain run and
    reproduce its states bit for bit; ghost = sum of the a-priori counts + 3."""
    from bench import pick_sweeps
    N, M, nsteps = 127, 21, 12
    x = np.linspace(0, 1, N + 1) ** 1.1
    y = np.linspace(0, 1, M + 1)
    cold = pick_sweeps(N, 0.01, 1e-13, x=x, y=y, consts=WARM_T)
    assert cold[0] >= 15, cold
    ghost = sum(cold) + 3
    assert (N + 1) // world >= ghost, (cold, ghost)
    ref_state, ref_norms = _single(1, N, M, nsteps, "pc", WARM_T, None)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 1, N, M, nsteps, "pc", WARM_T, None, ghost, str(tmp_path), {}),
             nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        for k in range(5):
            np.testing.assert_array_equal(z[f"f{k}"], ref_state[k][lo:hi])   # bit for bit
        np.testing.assert_allclose(z["norms"], ref_norms, rtol=1e-12, atol=1e-300)
        # the counts the library ended up with are those of the single-domain run, warm start included
        assert list(z["sweeps"]) == _single.sweeps and list(z["warm"]) == _single.warm
        # ... and every rank holds the record of the whole mesh (maxima over all slabs), not that of its rows
        assert list(z["increment"]) == _single.increment and list(z["first_update"]) == _single.first_update
    assert _single.warm[0] == 1 and _single.sweeps[0] < cold[0]


def _positive_state(N, M):
    """A smooth state whose cs is positive on every node, boundary included (min |x| of the cs Newton iteration
    is then not the rounding noise of a zero boundary value) and grows along x, so that slabs see different
    minima."""
    i = np.arange(N + 1)[:, None] / N
    j = np.arange(M + 1)[None, :] / M
    cp = 0.3 + 0.2 * np.sin(2 * i + j)
    T = 0.5 + 0.1 * np.cos(i - 3 * j)
    cl = 0.2 * np.sin(3 * i * j)
    cd = 0.4 + 0.1 * i * j
    cs = 0.02 + 0.9 * i ** 2 + 0.05 * j
    return [np.ascontiguousarray(a) for a in (cp, T, cl, cd, cs)]


def _early_exit_run(ctx, stepper, rows, N, M, nsteps, kw):
    zero = np.zeros((ctx.nrows, M + 1))
    lo = ctx.row_offset
    ctx.set_state(*[a[lo:lo + ctx.nrows] for a in _positive_state(N, M)])
    kept = []
    for k in range(nsteps):
        ctx.set_forcing(0, zero, zero, zero, zero, zero)
        ctx.set_forcing(1, zero, zero, zero, zero, zero)
        if stepper is None:
            ctx.step_pc(0.05 * k, 0.05, **kw)
        else:
            stepper.step_pc(0.05 * k, 0.05, **kw)
        kept.append(ctx.get_stats()["cs_iterations"])
    return kept


def _early_exit_worker(rank, world, port, N, M, nsteps, kw, ghost, use_hook, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_twin_cabi_parity import TWIN
        from tedd import _ffi
        _ffi.use_library(TWIN)
        x = np.linspace(0, 1, N + 1)
        y = np.linspace(0, 1, M + 1)
        slab = partition_rows(N + 1, world, rank, ghost)
        ctx = _ffi.Context(x, y, STRESS, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=-1,
                           row_lo=slab.row_lo, row_hi=slab.row_hi, ghost=slab.ghost)
        stepper = SlabStepper(ctx, slab, torch.device("cpu"))
        if not use_hook:
            ctx.set_reduce_hook(None)
        kept = _early_exit_run(ctx, stepper, slab, N, M, nsteps, kw)
        a, b = slab.owned_local
        np.savez(os.path.join(out_dir, f"exit{rank}.npz"), lo=slab.row_lo, hi=slab.row_hi, kept=np.array(kept),
                 **{f"f{k}": s[a:b] for k, s in enumerate(ctx.get_state())})
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cs_corrector_early_exit_is_decided_over_all_slabs(world, tmp_path):
    """The reference stops the cs corrector's Newton iteration when max|dx| < rtol min|x| over ALL nodes
    (src/prob1base.py:3661).  On a state whose cs is positive everywhere the exit fires; every slab must then
    keep the iteration count the whole mesh decides (all-reduce of the per-iteration record between the two
    passes of the corrector) and reproduce the single-domain state bit for bit.  Without the all-reduce the
    slabs decide differently -- the last assertion shows that the case can tell."""
    from test_twin_cabi_parity import TWIN, build_twin
    from tedd import _ffi
    build_twin()
    _ffi.use_library(TWIN)
    N, M, nsteps = 47, 21, 3
    kw = dict(cs_newton_iterations=12, cs_consec_xs_rtol=2e-3, jacobi_sweeps=[4, 3, 2])
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, np.linspace(0, 1, M + 1), STRESS, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE,
                       eta=50.0, mms_case=-1)
    kept_ref = _early_exit_run(ctx, None, None, N, M, nsteps, kw)
    ref_state = ctx.get_state()
    ctx.close()
    assert all(1 <= k < 12 for k in kept_ref), kept_ref        # the early exit fired on the whole mesh
    seen = {}
    for use_hook in (True, False):
        out = tmp_path / f"hook{int(use_hook)}"
        out.mkdir()
        port = _free_port()
        mp.spawn(_early_exit_worker, args=(world, port, N, M, nsteps, kw, 11, use_hook, str(out)), nprocs=world, join=True)
        seen[use_hook] = [np.load(out / f"exit{r}.npz") for r in range(world)]
    for z in seen[True]:
        assert list(z["kept"]) == kept_ref
        lo, hi = int(z["lo"]), int(z["hi"])
        for k in range(5):
            np.testing.assert_array_equal(z[f"f{k}"], ref_state[k][lo:hi])
    assert any(list(z["kept"]) != kept_ref for z in seen[False])


def test_too_few_ghost_rows_is_an_error(tmp_path):
    """A step that applies more stencils than there are ghost rows must fail loudly."""
    from test_twin_cabi_parity import TWIN, build_twin
    from tedd import _ffi
    build_twin()
    _ffi.use_library(TWIN)
    x = np.linspace(0, 1, 33)
    ctx = _ffi.Context(x, x, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=1,
                       row_lo=10, row_hi=20, ghost=3)
    ctx.set_state_exact(0.0)
    with pytest.raises(ValueError, match="ghost"):
        ctx.step_pc(0.0, 0.01, jacobi_sweeps=[4, 3, 2])
    ctx.close()


# ---- convergence sweeps: independent levels sharded across ranks
def test_level_assignment_is_balanced_and_complete():
    from tedd.distributed import assign_levels
    costs = [2.0 ** (3.5 * k) for k in range(8)] + [300.0, 600.0, 1200.0, 2400.0]
    for world in (1, 2, 3, 8, 16):
        plan = assign_levels(costs, world)
        assert sorted(k for part in plan for k in part) == list(range(len(costs)))
        loads = [sum(costs[k] for k in part) for part in plan]
        assert max(loads) <= max(costs) + sum(costs) / world     # the LPT bound
    assert assign_levels([1, 1, 1, 1], 2) == [[0, 2], [1, 3]]
    assert assign_levels([], 3) == [[], [], []]


def _sweep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import contextlib
    import io
    from test_twin_cabi_parity import TWIN
    sys.argv = ["sweep_sharded.py", "--case", "expsin", "--backend", "gloo", "--library", TWIN, "--max-n", "32"]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep_sharded
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sweep_sharded.main()
    with open(os.path.join(out_dir, f"sweep{rank}.txt"), "w") as f:
        f.write(buf.getvalue())


def test_sharded_sweep_reproduces_the_notebook(tmp_path):
    """tools/sweep_sharded.py on two gloo ranks over the host twin: ExpSin levels up to N = 32 of the spatial
    sweep plus the N = 32 temporal study, against the notebook's printed errors."""
    import json
    from test_twin_cabi_parity import build_twin
    build_twin()
    port = _free_port()
    mp.spawn(_sweep_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = json.loads((tmp_path / "sweep0.txt").read_text().strip().splitlines()[-1])
    assert (tmp_path / "sweep1.txt").read_text().strip() == ""
    assert out["ranks"] == 2 and out["levels"] == 5 + 6
    assert out["worst_relative_difference_to_notebook"] < 2e-9
    assert out["spatial_rates_3pt"] == pytest.approx(out["notebook_rates_3pt"][:3], abs=2e-3)
