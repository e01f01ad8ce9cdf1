"""Host twin of the library against the CPU oracle on seeded random inputs (tests/oracle_cases.py)."""
import pytest

import oracle_cases as oc
from test_twin_cabi_parity import TWIN, build_twin
from tedd import _ffi


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)
    yield
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)


VARIANTS = {"default": None}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("react", ["reg", "cs3", "hcs"])
@pytest.mark.parametrize("seed,N,M", [(1, 9, 7), (2, 33, 70), (3, 131, 67)])
def test_random_mesh_state_and_forcing(react, seed, N, M, variant):
    # Jacobi to 1e-14 against the oracle's direct solve; fused multiply-adds and exp() differ in the last bits
    assert oc.check_random_step(react, seed, N, M, VARIANTS[variant]) < 5e-13
