"""CPU checks of the device source: the node functors and the step sequencing of csrc/,
compiled for the host with a loop backend (tests/cpu_twin), against the reference's golden
vectors.  This is how the arithmetic and the buffer bookkeeping are debugged without a GPU;
the GPU tests repeat the same checks through libtedd_hip.so."""
import os
import subprocess

import numpy as np
import pytest

import cabi_cases as cc
from tedd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN_DIR = os.path.join(ROOT, "tests", "cpu_twin")
# TEDD_TWIN_SANITIZE=1: every test that uses the twin runs on its AddressSanitizer + UndefinedBehaviorSanitizer
# build (the process must have been started with the sanitizer runtime preloaded: tests/test_twin_sanitized.py)
SANITIZE = os.environ.get("TEDD_TWIN_SANITIZE") == "1"
TWIN = os.path.join(TWIN_DIR, "libtedd_twin_asan.so" if SANITIZE else "libtedd_twin.so")
CSRC = os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd", "csrc")
SANITIZE_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def build_twin():
    srcs = [os.path.join(TWIN_DIR, "tedd_twin.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    srcs.append(os.path.join(ROOT, "include", "tedd.h"))
    if os.path.exists(TWIN) and all(os.path.getmtime(TWIN) >= os.path.getmtime(s) for s in srcs):
        return
    opt = SANITIZE_FLAGS if SANITIZE else ["-O2"]
    subprocess.check_call(["g++"] + opt + ["-fopenmp", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, "-o", TWIN,
                                           os.path.join(TWIN_DIR, "tedd_twin.cpp")])


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)
    yield


@pytest.mark.parametrize("tag", cc.tags("golden_ops.npz", "/consts"))
def test_field_and_jacobian_stencils(tag):
    assert cc.check_ops(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_mms.npz", "/consts"))
def test_mms_exact_and_forcing(tag):
    e_exact, e_forcing = cc.check_mms(tag)
    assert e_exact < 1e-13
    assert e_forcing < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_steps.npz", "/t0"))
def test_integrator_steps(tag):
    err, res_err, stats = cc.run_steps(tag)
    assert err < (2e-12 if "bigdt" in tag else 5e-14), (tag, err, stats)
    assert res_err < 1e-13, (tag, res_err)


@pytest.mark.parametrize("tag", cc.tags("golden_steps.npz", "/t0"))
@pytest.mark.parametrize("options", [{"cache_forcing": 0}, {"fuse_forcing": 0}], ids=["in_kernels", "own_launch"])
def test_integrator_steps_with_other_forcing_modes(tag, options):
    if not tag.startswith("pc"):
        pytest.skip("the forcing modes concern the predictor-corrector step")
    err, res_err, stats = cc.run_steps(tag, options=options)
    assert err < (2e-12 if "bigdt" in tag else 5e-14), (tag, err, stats)
    assert res_err < 1e-13, (tag, res_err)


# ---- HCsTriple (sharp Heaviside) field, forcing terms and predictor-corrector integrator
@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/hcs/Fcp"))
def test_hcs_field_and_jacobian_stencils(tag):
    assert cc.check_ops_hcs(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/t0/hcs/fcp"))
def test_hcs_forcing(tag):
    assert cc.check_mms_hcs(tag) < 1e-13


@pytest.mark.parametrize("tag", cc.tags("golden_hcs.npz", "/t0"))
def test_hcs_integrator_steps(tag):
    err, res_err, stats = cc.run_steps_hcs(tag)
    assert err < 5e-14, (tag, err, stats)
    assert res_err < 1e-13, (tag, res_err)


def test_hcs_corrector_reports_nonpositive_denominator():
    cc.check_hcs_denominator_error()


def test_prefetch_forcing_changes_nothing():
    cc.check_prefetch_forcing()


# ---- exact solutions given as expression programs (TEDD_MMS_PROGRAM)
@pytest.mark.parametrize("react", ["reg", "cs3", "hcs"])
def test_mms_program_equals_closed_form(react):
    cc.check_mms_program(react)


# ---- the pieces of a predictor-corrector step called on their own
@pytest.mark.parametrize("streamed", [False, True], ids=["closed_form", "streamed"])
@pytest.mark.parametrize("tag", cc.tags("golden_substeps.npz", "/t0"))
def test_substeps(tag, streamed):
    err, res_err = cc.check_substeps(tag, streamed)
    assert err < 5e-14, (tag, err)
    assert res_err < 1e-13, (tag, res_err)
