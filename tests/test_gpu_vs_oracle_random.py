"""HIP path against the CPU oracle on seeded random inputs (tests/oracle_cases.py)."""
import pytest

import oracle_cases as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    from tedd import _ffi

    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


VARIANTS = {"default": None, "large_tiles": {"strip_shape": 0}, "eight_row_strips": {"strip_shape": 4}, "mid_tiles_h3": {"strip_shape": 2, "strip_hmax": 3},
            "sweeps": {"strip_solve": 0}}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("react", ["reg", "cs3", "hcs"])
@pytest.mark.parametrize("seed,N,M", [(1, 9, 7), (2, 33, 70), (3, 131, 67)])
def test_random_mesh_state_and_forcing(react, seed, N, M, variant):
    # Jacobi to 1e-14 against the oracle's direct solve; fused multiply-adds and exp() differ in the last bits
    assert oc.check_random_step(react, seed, N, M, VARIANTS[variant]) < 5e-13
