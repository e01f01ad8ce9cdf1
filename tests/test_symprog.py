"""Lowering of SymPy expressions to register programs (tedd/_symprog.py), checked on the host against
sympy.lambdify with the NumPy interpreter of the same program format the device runs."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sympy = pytest.importorskip("sympy")
from tedd import _symprog as sp  # noqa: E402

t, x, y = sympy.symbols("t x y", real=True)
XX, YY = np.meshgrid(np.linspace(0.05, 1.0, 9), np.linspace(0.1, 0.95, 7), indexing="ij")


def _check(exprs, tt=0.37, rtol=1e-14):
    code, consts, n_regs = sp.lower(exprs, t, x, y)
    assert n_regs <= sp.MAX_REGS and code.dtype == np.uint32
    out = sp.run_numpy(code, consts, n_regs, len(exprs), tt, XX, YY)
    for e, o in zip(exprs, out):
        ref = np.broadcast_to(sympy.lambdify((t, x, y), e, "numpy")(tt, XX, YY), XX.shape)
        np.testing.assert_allclose(o, ref, rtol=rtol, atol=1e-300)
    return code, consts, n_regs


def test_arithmetic_and_functions():
    W = x * (1 - x) * y * (1 - y)
    _check([W * sympy.exp(-3 * t), sympy.sin(sympy.pi * x) * sympy.cos(2 * y) / (1 + t) ** 2,
            sympy.sqrt(1 + x * y) + sympy.Abs(x - 0.3) ** 2.5, x ** -3 * sympy.tanh(y) + sympy.log(2 + t),
            sympy.S(0), x, -x - y - t, 1 / sympy.sqrt(x + 1), sympy.sign(x - 0.5) * y, sympy.cosh(x) - sympy.sinh(y),
            sympy.tan(x), sympy.Rational(1, 3) * x - 2 * y / 7, sympy.E ** (x * y), (x + y) ** 20])


def test_derivatives_of_a_library_style_case():
    """All 21 node outputs and the 4 cell outputs of a case with peaks and oscillation in time."""
    u = 3 * (x ** 2 + y ** 2) ** 3 * sympy.sin(sympy.pi * x) * sympy.sin(sympy.pi * y) * (1 + sympy.sin(7 * t)) \
        * sympy.exp(-t / 3)
    named = {}
    for k, name in enumerate(("cp", "T", "cl", "cd", "cs")):
        e = u * (k + 1) + k * x * y * t
        dx, dy = sympy.diff(e, x), sympy.diff(e, y)
        named.update({name: e, "dt_" + name: sympy.diff(e, t), "dx_" + name: dx, "dy_" + name: dy,
                      "lap_" + name: sympy.diff(dx, x) + sympy.diff(dy, y)})
    (code, consts, n_regs), (ccode, cconsts, cregs) = sp.lower_case(named, t, x, y)
    out = sp.run_numpy(code, consts, n_regs, len(sp.NODE_OUTPUTS), 0.21, XX, YY)
    for name, o in zip(sp.NODE_OUTPUTS, out):
        ref = sympy.lambdify((t, x, y), named[name], "numpy")(0.21, XX, YY)
        np.testing.assert_allclose(o, ref, rtol=1e-12, atol=1e-13, err_msg=name)
    out = sp.run_numpy(ccode, cconsts, cregs, len(sp.CELL_OUTPUTS), 0.21, XX, YY)
    for name, o in zip(sp.CELL_OUTPUTS, out):
        ref = sympy.lambdify((t, x, y), named[name], "numpy")(0.21, XX, YY)
        np.testing.assert_allclose(o, ref, rtol=1e-12, atol=1e-13, err_msg=name)


def test_program_words_are_well_formed():
    code, consts, n_regs = _check([x * y + t, sympy.exp(x) * sympy.exp(x) + 2.5])
    defined = {0, 1, 2}
    for w in code.tolist():
        op, d, a, b = w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, w >> 24
        assert op <= sp.OP_MOV
        if op == sp.OP_CONST:
            assert (a | (b << 8)) < len(consts) and d < n_regs
            defined.add(d)
            continue
        assert a in defined
        if op in (sp.OP_ADD, sp.OP_SUB, sp.OP_MUL, sp.OP_DIV, sp.OP_POW):
            assert b in defined
        if op != sp.OP_OUT:
            assert d < n_regs
            defined.add(d)


def test_registers_are_reused():
    # a long chain needs only a handful of registers
    e = x
    for k in range(40):
        e = sympy.sin(e) + k
    code, consts, n_regs = sp.lower([e], t, x, y)
    assert len(code) > 80 and n_regs <= 8


def test_unsupported_expressions_are_refused():
    for e in (sympy.erf(x), sympy.Piecewise((x, x > 0), (0, True)), sympy.Symbol("q") * x, sympy.Heaviside(x)):
        with pytest.raises(sp.Unsupported):
            sp.lower([e], t, x, y)
    # more live values than registers
    wide = sum(sympy.sin((k + 2) * x) * sympy.cos((k + 3) * y) for k in range(100))
    many = [sympy.diff(wide, x, k) for k in range(4)]
    with pytest.raises(sp.Unsupported):
        sp.lower(many, t, x, y, max_regs=6)
