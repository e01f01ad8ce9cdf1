"""GPU parity tests: the HIP path, called through the C ABI, against vectors produced by the
reference (tests/golden) -- the same fixtures that pin the CPU oracle.

Tolerances (float64, stated per check): the device code evaluates the same formulas as the
reference with fused multiply-adds, reciprocal metric tables and its own exp(); the Newton
systems are solved by Jacobi sweeps to a relative update of 1e-13 instead of SuperLU."""
import numpy as np
import pytest

import cabi_cases as cc

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


@pytest.mark.parametrize("tag", cc.tags("golden_steps.npz", "/t0"))
def test_integrator_steps(tag):
    err, res_err, stats = cc.run_steps(tag)
    # state after 1, 2 and 5 steps, relative to the largest field magnitude
    assert err < (2e-12 if "bigdt" in tag else 5e-14), (tag, err, stats)
    # Newton residuals: absolute (they are differences of O(1) quantities)
    assert res_err < 1e-13, (tag, res_err)
