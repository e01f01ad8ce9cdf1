"""The verdict on the Jacobi solves and the warm start of the Newton systems, on the host twin (the GPU tests
repeat them through libtedd_hip.so, tests/test_gpu_cabi_parity.py).  Reference: the direct solves of
src/prob1base.py:2088-2130 cannot stop short; sweeps can, and the library must say so."""
import numpy as np
import pytest

import cabi_cases as cc
from test_twin_cabi_parity import TWIN, build_twin
from tedd import _ffi


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)
    yield


def test_too_few_sweeps_through_advance_is_an_error():
    cc.check_under_converged_advance_raises()


def test_warm_start_needs_fewer_sweeps_for_the_same_tolerance():
    cc.check_warm_start_same_answer_fewer_sweeps()


def test_fixed_sweeps_are_not_judged_and_not_warm_started():
    cc.check_fixed_sweeps_are_plain_jacobi()


def test_null_output_pointers_are_refused():
    cc.check_null_output_pointers_are_refused()


def test_failed_create_leaves_the_library_usable():
    cc.check_failed_create_releases_what_it_took()
