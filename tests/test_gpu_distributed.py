"""Two slabs on one MI355X (two processes sharing cuda:0; ghost rows staged through the host
over gloo, since two ranks cannot share one device under RCCL): the HIP slab path must
reproduce the single-slab HIP run bit for bit on every owned row."""
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

pytestmark = pytest.mark.gpu

NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]
N, M, NSTEPS, SWEEPS = 300, 130, 3, [5, 3, 2]


def _free_port():
    """A TCP port nobody listens on right now (for the rendezvous of the ranks a test starts)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(ctx, stepper=None):
    ctx.set_state_exact(0.0)
    dt, t = (1.0 / N) ** 1.5, 0.0
    for _ in range(NSTEPS):
        if stepper is None:
            ctx.step_pc(t, dt, jacobi_sweeps=SWEEPS)
        else:
            stepper.step_pc(t, dt, jacobi_sweeps=SWEEPS)
        t += dt
    if stepper is not None:
        stepper.finish()
    return ctx.get_state()


def _worker(rank, world, port, out_dir, options):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tedd import _ffi
        from tedd.distributed import partition_rows
        from gloo_staging import HostStagedSlabStepper as SlabStepper
        torch.cuda.set_device(0)
        x = np.linspace(0, 1, N + 1)
        y = np.linspace(0, 1, M + 1)
        slab = partition_rows(N + 1, world, rank, sum(SWEEPS) + 2)
        ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                           mms_case=_ffi.MMS_EXPSIN, row_lo=slab.row_lo, row_hi=slab.row_hi, ghost=slab.ghost)
        options = dict(options)
        overlap = bool(options.pop("_overlap", False))
        for name, value in options.items():
            ctx.set_option(name, value)
        stepper = SlabStepper(ctx, slab, torch.device("cuda", 0), overlap=overlap)
        state = _run(ctx, stepper)
        a, b = slab.owned_local
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=slab.row_lo, hi=slab.row_hi,
                 **{f"f{k}": s[a:b] for k, s in enumerate(state)})
        ctx.close()
    finally:
        dist.destroy_process_group()


# forcing_launch: the forcing arrays come from a launch of their own, enqueued by the slab stepper while the
# ghost rows travel (the configuration of slabs too large for the predictor launch to make them)
# overlap: the ghost rows of a step travel while the next step's predictor runs on the rows that do not need them
# march: the marching launchers that large slabs use, forced on this small mesh (the split predictor of the overlap mode
# then runs its interior part in the marching launcher and its bands one thread per node)
@pytest.mark.parametrize("options", [{}, {"fuse_forcing": 0}, {"_overlap": True}, {"_overlap": True, "fuse_forcing": 0},
                                     {"_overlap": True, "fuse_forcing": 0, "march": 1}],
                         ids=["default", "forcing_launch", "overlap", "overlap-forcing_launch", "overlap-marching"])
def test_two_slabs_equal_single_slab(tmp_path, options):
    from tedd import _ffi
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    x = np.linspace(0, 1, N + 1)
    y = np.linspace(0, 1, M + 1)
    ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ref = _run(ctx)
    ctx.close()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), options), nprocs=2, join=True)
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        for k in range(5):
            np.testing.assert_array_equal(z[f"f{k}"], ref[k][lo:hi])


def test_halo_pack_and_unpack_on_device_views():
    """The pack / unpack copies of the RCCL exchange path (multi-tensor copies between ghost-row slices of the
    library's own device arrays, wrapped through their raw addresses, and the message buffers)."""
    from tedd import _ffi
    from tedd.distributed import _copy_many, wrap_device_fields
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    x = np.linspace(0, 1, 41)
    y = np.linspace(0, 1, 24)
    ctx = _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.1)
    ref = ctx.get_state()
    dev = torch.device("cuda", 0)
    fields = wrap_device_fields(ctx.state_device_ptrs(), ctx.nrows, ctx.pitch, dev)
    send = torch.empty((5, 7, ctx.pitch), dtype=torch.float64, device=dev)
    _copy_many([send[k] for k in range(5)], [f[3:10] for f in fields])
    torch.cuda.synchronize()
    for k in range(5):
        np.testing.assert_array_equal(send[k].cpu().numpy()[:, :24], ref[k][3:10])
    _copy_many([f[30:37] for f in fields], [send[k] for k in range(5)])   # rows 3..9 -> rows 30..36
    torch.cuda.synchronize()
    ctx.ghosts_refreshed()
    out = ctx.get_state()
    for k in range(5):
        np.testing.assert_array_equal(out[k][30:37], ref[k][3:10])
        np.testing.assert_array_equal(out[k][:30], ref[k][:30])
    ctx.close()
