"""StateVars of the mirror (reference src/prob1base.py:1854-2085; the scenarios of the reference's
tests/test_statevars.py:93-336): stored fields, the 24 lazily cached derived arrays against arrays
produced by the reference's class (tests/golden/make_golden.py, gen_statevars), caching of prerequisites,
immutability.  Host only."""
import os

import numpy as np
import pytest

import tedd.prob1base as p1

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_statevars.npz"))
DERIVED = sorted({k.split("/")[1] for k in Z.files if k.startswith("kind1/")})
FIELDS = ("cp", "T", "cl", "cd", "cs")


def make(kind):
    consts = p1.ModelConsts(**dict(zip(p1.ModelConsts._fields, Z["consts"])))
    model = (p1.DefaultModel01 if kind == 1 else p1.DefaultModel02)(consts)
    grid = p1.Grid(Z["x"], Z["y"])
    return p1.StateVars(**{f: Z[f] for f in FIELDS}, model=model, hh=grid.hh, kk=grid.kk), model, grid


def test_the_reference_has_the_same_derived_names():
    assert len(DERIVED) == 24
    assert sorted(p1.StateVars._COMPUTED_PROPERTIES) == DERIVED


def test_stored_fields_and_parameters():
    S, model, grid = make(1)
    assert S.model is model
    np.testing.assert_array_equal(S.hh, Z["hh"])
    np.testing.assert_array_equal(S.kk, Z["kk"])
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(S, f), Z[f])
        assert getattr(S, f) is getattr(S, f"_{f}_data")


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("name", DERIVED)
def test_derived_array(name, kind):
    S, _model, _grid = make(kind)
    value = getattr(S, name)
    ref = Z[f"kind{kind}/{name}"]
    assert value.shape == ref.shape
    # first row / column of backward differences hold inf - inf or x / inf of the padded spacings
    np.testing.assert_allclose(value, ref, rtol=1e-13, atol=0, equal_nan=True)
    assert getattr(S, name) is value            # cached: same object on the second access
    assert hasattr(S, "_cache_" + name)
    with pytest.raises(AttributeError):
        setattr(S, name, np.zeros_like(value))


def test_prerequisites_are_cached_by_a_dependent_array():
    S, _model, _grid = make(2)
    assert not hasattr(S, "_cache_Mxcp") and not hasattr(S, "_cache_MxT")
    dd = S.Dd_MxcpT
    assert hasattr(S, "_cache_Mxcp") and hasattr(S, "_cache_MxT") and hasattr(S, "_cache_Dd_MxcpT")
    assert S.Dd_MxcpT is dd
    assert not hasattr(S, "_cache_Mycp")


def test_immutability():
    S, model, _grid = make(1)
    for name, value in (("hh", 0.99), ("kk", 0.88), ("model", model.copy()), ("cp", np.zeros_like(S.cp)), ("T", None),
                        ("brand_new", 1)):
        with pytest.raises(AttributeError):
            setattr(S, name, value)
    with pytest.raises(AttributeError):
        del S.cp
    with pytest.raises(AttributeError):
        S.no_such_array


def test_with_changes_copy_and_into_dict():
    S, model, _grid = make(1)
    S2 = S.with_changes(cl=S.cl + 1.0)
    assert S2 is not S and S2.model is model and S2.cp is S.cp
    np.testing.assert_array_equal(S2.cl, S.cl + 1.0)
    np.testing.assert_allclose(S2.Dmxcl[1:], S.Dmxcl[1:], rtol=0, atol=1e-12)   # a constant shift has no gradient
    with pytest.raises(ValueError):
        S.with_changes(hh=S.hh)
    S3 = S.copy()
    assert S3 is not S and all(getattr(S3, f) is getattr(S, f) for f in FIELDS)
    d = S.into_dict({}, which=["MxT", "cp"])
    assert set(d) == {"MxT", "cp"} and d["MxT"] is S.MxT
    assert set(S.into_dict({})) == set(DERIVED) | set(FIELDS)


def test_state_arrays_cannot_be_edited_in_place():
    """Device copies and field values are cached by the identity of the StateVars: an edit in place would
    go unseen (the reference re-reads the arrays on every call), so it must fail."""
    S, _model, _grid = make(1)
    for v in ("cp", "T", "cl", "cd", "cs"):
        with pytest.raises(ValueError):
            getattr(S, v)[1, 1] = 3.0
    S2 = S.with_changes(T=np.array(S.T))      # the way to change a state: a new one
    assert S2.T is not S.T and S2.cp is S.cp
