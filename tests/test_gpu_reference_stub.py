"""The reference-side binding over the HIP library: the trial of tests/test_reference_stub.py on the GPU."""
import numpy as np
import pytest

import test_reference_stub as trs
from tedd import _ffi, reference_stub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_library():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    _ffi.use_library(_ffi.DEFAULT_LIBRARY)
    yield


@pytest.mark.parametrize("mms_case", [reference_stub.MMS_STREAM, reference_stub.MMS_EXPSIN], ids=["streamed", "closed-form"])
def test_binding_reproduces_the_reference_trial_on_the_gpu(mms_case):
    cls = reference_stub.make_integrator_class(_ffi.DEFAULT_LIBRARY, mms_case=mms_case)
    tr = trs._trial(cls)
    s = tr.run_for_errors(Tf=trs.PIN["Tf"], dt=(1.0 / trs.PIN["N"]) ** 1.5)
    ref = trs.PIN["reference"]
    assert abs(s.overall_combined_error - ref["overall"]) <= 1e-9 * ref["overall"]
    for v, e in ref["per_variable"].items():
        assert abs(s.per_variable_sup_errors[v] - e) <= 1e-9 * e + 1e-15, v
    out = tr.integrator.step(tr.initial_state, t0=0.0, dt=(1.0 / trs.PIN["N"]) ** 1.5)
    sums = [float(np.abs(getattr(out, v)).sum()) for v in ("cp", "T", "cl", "cd", "cs")]
    np.testing.assert_allclose(sums, trs.PIN["one_step_state_checksums"], rtol=1e-13)


def test_last_residual_side_channel_on_the_gpu():
    trs.check_last_residual(_ffi.DEFAULT_LIBRARY)


def test_model_constants_are_read_at_every_step_on_the_gpu():
    trs.check_model_changed_in_place(_ffi.DEFAULT_LIBRARY)


def test_unconverged_solves_do_not_pass_silently_on_the_gpu():
    trs.check_unconverged_solves_do_not_pass_silently(_ffi.DEFAULT_LIBRARY)
