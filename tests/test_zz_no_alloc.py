"""tedd_error_norms_noalloc (the entry point that allocates nothing) against tedd_error_norms, through the
C ABI on the GPU: two calls in a row on the same GPU, the second on a smaller mesh, then the first context
again -- the static scratch array keeps the partial sums of the call before, and none of them may leak
into a later result.  The sums are the same kernels in the same order, so the comparison is bit for bit."""
import numpy as np
import pytest

NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]


def _context(N, M, case, params=()):
    from tedd import _ffi
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    x = np.linspace(0, 1, N + 1) ** 1.05
    y = np.linspace(0, 1, M + 1)
    return _ffi.Context(x, y, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=case,
                        mms_params=params)


@pytest.mark.gpu
def test_noalloc_norms_equal_the_allocating_entry_point():
    import torch
    from tedd import _ffi
    assert torch.cuda.is_available()
    out = torch.full((8,), float("nan"), dtype=torch.float64, device="cuda:0")
    # 1101 x 521 nodes are 276 x 9 = 2484 tiles of 4 x 64 nodes, more than the 2048 workgroups the reduction
    # launches: every slot of the static array is written and the workgroups stride over the tiles
    big = _context(1100, 520, _ffi.MMS_EXPSIN)
    # a handful of workgroups: leaves nearly all slots holding the sums of `big`
    small = _context(37, 21, _ffi.MMS_PEAKS, (1.0, 10.0))
    try:
        for ctx, steps in ((big, 3), (small, 2), (big, 1)):
            dt = 1e-3
            ctx.set_state_exact(0.0)
            ctx.advance(0.0, dt, steps, want_errors=False)
            t = steps * dt
            ref = ctx.error_norms(t)                     # existing entry point (context-owned partials)
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            ctx.error_norms_noalloc(t, out.data_ptr())
            ctx.sync()
            got = out.cpu().numpy()
            assert np.isfinite(ref).all() and (ref[:5] > 0).all()
            np.testing.assert_array_equal(got, ref)      # bit for bit
    finally:
        big.close()
        small.close()


@pytest.mark.gpu
def test_noalloc_norms_reject_what_they_cannot_serve():
    import torch
    from tedd import _ffi
    out = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    x = np.linspace(0, 1, 9)
    ctx = _ffi.Context(x, x, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_STREAM)
    try:
        z = np.zeros((9, 9))
        ctx.set_state(z, z, z, z, z)
        with pytest.raises(ValueError, match="closed-form"):
            ctx.error_norms_noalloc(0.0, out.data_ptr())
        with pytest.raises(ValueError, match="null"):
            ctx.error_norms_noalloc(0.0, 0)
    finally:
        ctx.close()
    xb = np.linspace(0, 1, 8193)                       # 8193^2 nodes > 2^26
    ctx = _ffi.Context(xb, xb, NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    try:
        ctx.set_state_exact(0.0)
        with pytest.raises(ValueError, match="2\\^26"):
            ctx.error_norms_noalloc(0.0, out.data_ptr())
    finally:
        ctx.close()


def test_noalloc_norms_on_the_host_twin():
    """Host logic of the entry point (limits, dispatch) over the host-compiled twin: `device` memory is host memory there."""
    from test_twin_cabi_parity import TWIN, build_twin
    from tedd import _ffi
    build_twin()
    _ffi.use_library(TWIN)
    x = np.linspace(0, 1, 24) ** 1.05
    ctx = _ffi.Context(x, x[:17], NB_A, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_POL)
    try:
        ctx.set_state_exact(0.0)
        ctx.advance(0.0, 1e-3, 2, want_errors=False)
        ref = ctx.error_norms(2e-3)
        out = np.full(8, np.nan)
        ctx.error_norms_noalloc(2e-3, out.ctypes.data)
        np.testing.assert_array_equal(out, ref)
    finally:
        ctx.close()
