"""The reference-side binding (tedd/reference_stub.py: ctypes and NumPy only, the listing a maintainer of the
reference copies) over the host-compiled library.  tests/golden/make_integration_pin.py ran it inside the
reference's own MMSTrial, with the reference's classes, against the reference's integrator and stored the
reference's numbers; here the same trial is rebuilt from this package's mirror of those classes (the reference
does not travel) and must reproduce the reference's error summary."""
import json
import os

import numpy as np
import pytest

from test_twin_cabi_parity import TWIN, build_twin
from tedd import _ffi, reference_stub
from tedd import mms_trial_utils as mtu
from tedd import prob1_mms_cases as p1mc
from tedd import prob1base as p1

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = json.load(open(os.path.join(HERE, "golden", "golden_integration_pin.json")))


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)       # (the mirror's field and forcing-term classes evaluate through it)
    yield


def _trial(integrator_cls):
    consts = json.load(open(os.path.join(HERE, "golden", "model_sets.json")))[PIN["model_set"]]
    model = p1.DefaultModel02(p1.ModelConsts(**consts))
    grid = p1.make_uniform_grid(N=PIN["N"], M=PIN["N"])
    common = {"regularization_factor": PIN["eta"]}
    return mtu.MMSTrial(grid=grid, model=model, mms_case_cls=p1mc.MMSCaseExpSin,
                        field_cls=p1.SemiDiscreteField_RegHCsTriple, forcing_terms_cls=p1.ForcingTerms_RegHCsTriple,
                        integrator_cls=integrator_cls, integrator_params=dict(common),
                        forcing_terms_params=common, field_params=common)


@pytest.mark.parametrize("mms_case", [reference_stub.MMS_STREAM, reference_stub.MMS_EXPSIN], ids=["streamed", "closed-form"])
def test_binding_reproduces_the_reference_trial(mms_case):
    cls = reference_stub.make_integrator_class(TWIN, mms_case=mms_case)
    tr = _trial(cls)
    assert type(tr.integrator) is cls
    s = tr.run_for_errors(Tf=PIN["Tf"], dt=(1.0 / PIN["N"]) ** 1.5)
    ref = PIN["reference"]
    assert s.dt_used == ref["dt_used"]
    assert abs(s.overall_combined_error - ref["overall"]) <= 1e-9 * ref["overall"]
    for v, e in ref["per_variable"].items():
        # (the mirror's exact solution differs from the reference's lambdified one in the last bit of O(1)
        # values, which is all that is left of an error of 3e-11)
        assert abs(s.per_variable_sup_errors[v] - e) <= 1e-9 * e + 1e-15, v
    out = tr.integrator.step(tr.initial_state, t0=0.0, dt=(1.0 / PIN["N"]) ** 1.5)
    sums = [float(np.abs(getattr(out, v)).sum()) for v in ("cp", "T", "cl", "cd", "cs")]
    np.testing.assert_allclose(sums, PIN["one_step_state_checksums"], rtol=1e-13)


def test_binding_reports_library_errors():
    cls = reference_stub.make_integrator_class(TWIN)
    tr = _trial(cls)
    with pytest.raises(ValueError):
        tr.integrator.args = (1, 1, -3, 1e-6)      # a negative iteration count is the library's TEDD_ERR_INVALID
        tr.integrator.step(tr.initial_state, t0=0.0, dt=1e-3)


def test_listing_in_the_integration_notes_is_the_tested_file():
    """INTEGRATION.md prints the binding; what it prints is the file these tests run."""
    root = os.path.dirname(HERE)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    src = open(reference_stub.__file__).read()
    body = src[src.index("import ctypes as C"):]
    assert body.strip() in doc
