"""Shared drivers for the C-ABI parity tests: they run one golden fixture through a C-ABI
library (the HIP library on a GPU box, or the host twin on CPU) and return error measures."""
import os

import numpy as np
import pytest

from tedd import _ffi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VARS = ("cp", "T", "cl", "cd", "cs")
CASE_ID = {"pol": _ffi.MMS_POL, "expsin": _ffi.MMS_EXPSIN, "peaks_fast1e1": _ffi.MMS_PEAKS,
           "nfs_h2_h3": _ffi.MMS_NFS}
CASE_PARAMS = {"pol": (), "expsin": (), "peaks_fast1e1": (1.0, 10.0),
               "nfs_h2_h3": (2.1, 3.1, 3.1, 3.1, 2.1, 1.0 / np.pi)}
JAC_OPS = ["delT_ab_FT_ij", "delT_ab_Fcl_ij", "delcl_ab_Fcl_ij", "delT_ab_Fcd_ij", "delcl_ab_Fcd_ij",
           "delcd_ab_Fcd_ij"]

_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name))
    return _cache[name]


def tags(name, marker):
    z = load(name)
    return sorted({k[: -len(marker)] for k in z.files if k.endswith(marker)})


def make_ctx(z, tag, react, mms_case, mms_params=()):
    return _ffi.Context(z[f"{tag}/x"], z[f"{tag}/y"], z[f"{tag}/consts"],
                        model_kind=int(z[f"{tag}/model_kind"]), reaction_kind=react,
                        eta=float(z[f"{tag}/eta"]), mms_case=mms_case, mms_params=mms_params)


def relerr(a, b, scale=None):
    s = np.max(np.abs(b)) if scale is None else scale
    d = np.max(np.abs(np.asarray(a) - np.asarray(b)))
    return d / s if s > 0 else d


def check_ops(tag):
    """F* and the six Jacobian stencils on a random state with zero forcing. Returns max error."""
    z = load("golden_ops.npz")
    worst = 0.0
    for rname, rk in (("reg", _ffi.REACT_REGHCSTRIPLE), ("cs3", _ffi.REACT_CSTRIPLE)):
        ctx = make_ctx(z, tag, rk, _ffi.MMS_STREAM)
        S = [z[f"{tag}/{v}"] for v in VARS]
        ctx.set_state(*S)
        zero = np.zeros_like(S[0])
        ctx.set_forcing(0, zero, zero, zero, zero, zero)
        F = ctx.eval_field(0.3)
        for v, a in zip(VARS, F):
            worst = max(worst, relerr(a, z[f"{tag}/{rname}/F{v}"]))
        for op, name in enumerate(JAC_OPS):
            J = ctx.eval_jacobian(op)
            g = z[f"{tag}/{rname}/{name}"]
            sc = np.max(np.abs(g))
            for n in range(5):
                worst = max(worst, relerr(J[n], g[n], sc))
        ctx.close()
    return worst


def check_mms(tag):
    """Exact solution and the five forcing arrays of a closed-form case at two times."""
    z = load("golden_mms.npz")
    cname = tag.split("/")[0]
    worst_exact, worst_forcing = 0.0, 0.0
    for rname, rk in (("reg", _ffi.REACT_REGHCSTRIPLE), ("cs3", _ffi.REACT_CSTRIPLE)):
        ctx = make_ctx(z, tag, rk, CASE_ID[cname], CASE_PARAMS[cname])
        for it in (0, 1):
            tt = f"{tag}/t{it}"
            t = float(z[f"{tt}/t"])
            for v, a in zip(VARS, ctx.eval_exact(t)):
                worst_exact = max(worst_exact, relerr(a, z[f"{tt}/{v}"]))
            fscale = max(1.0, np.max(np.abs(z[f"{tt}/cp"])))
            for v, a in zip(("fcp", "fT", "fcl", "fcd", "fcs"), ctx.eval_forcing(t)):
                worst_forcing = max(worst_forcing, relerr(a, z[f"{tt}/{rname}/{v}"], fscale))
        ctx.close()
    return worst_exact, worst_forcing


def run_steps(tag, options=None, **overrides):
    """Integrate a golden step fixture; returns (max state error relative to the field scale,
    max absolute residual error, stats of the last step)."""
    z = load("golden_steps.npz")
    kind = tag.split("/")[0]
    cname = str(z[f"{tag}/case"]) if f"{tag}/case" in z.files else tag.split("/")[1]
    rk = _ffi.REACT_REGHCSTRIPLE if kind.endswith("reg") else _ffi.REACT_CSTRIPLE
    ctx = make_ctx(z, tag, rk, CASE_ID[cname], CASE_PARAMS[cname])
    for name, value in (options or {}).items():
        ctx.set_option(name, value)
    ctx.set_state(*[z[f"{tag}/s0/{v}"] for v in VARS])
    t0, dt = float(z[f"{tag}/t0"]), float(z[f"{tag}/dt"])
    saved = sorted(int(k.split("/")[-2][1:]) for k in z.files
                   if k.startswith(tag + "/s") and k.endswith("/cp") and "/s0/" not in k)
    kw = {}
    is_pc = kind.startswith("pc")
    if is_pc:
        for name in ("num_pc_steps", "num_newton_steps"):
            if f"{tag}/{name}" in z.files:
                kw[name] = int(z[f"{tag}/{name}"])
        if rk == _ffi.REACT_REGHCSTRIPLE:
            kw["cs_newton_iterations"] = int(z[f"{tag}/num_newton_iterations"])
            kw["cs_consec_xs_rtol"] = float(z[f"{tag}/consec_xs_rtol"])
        kw.update(overrides)
    t, worst, wres = t0, 0.0, 0.0
    for k in range(1, max(saved) + 1):
        if is_pc:
            ctx.step_pc(t, dt, **kw)
        else:
            ctx.step_feuler(t, dt)
        t += dt
        if k in saved:
            out = ctx.get_state()
            scale = max(np.max(np.abs(z[f"{tag}/s{k}/{v}"])) for v in VARS)
            for v, a in zip(VARS, out):
                worst = max(worst, np.max(np.abs(a - z[f"{tag}/s{k}/{v}"])) / scale)
            if is_pc:
                for v, a in zip(("T", "cl", "cd"), ctx.get_residuals()):
                    wres = max(wres, np.max(np.abs(a - z[f"{tag}/s{k}/res_{v}"])))
    stats = ctx.get_stats() if is_pc else {}
    ctx.close()
    return worst, wres, stats


# ---------------------------------------------------------------------------------- HCsTriple
def check_ops_hcs(tag):
    """F* and the six Jacobian stencils of the sharp-Heaviside field on a random state (cs of both
    signs and an exact zero), zero forcing."""
    z = load("golden_hcs.npz")
    ctx = make_ctx(z, tag, _ffi.REACT_HCSTRIPLE, _ffi.MMS_STREAM)
    S = [z[f"{tag}/{v}"] for v in VARS]
    ctx.set_state(*S)
    zero = np.zeros_like(S[0])
    ctx.set_forcing(0, zero, zero, zero, zero, zero)
    worst = 0.0
    for v, a in zip(VARS, ctx.eval_field(0.3)):
        worst = max(worst, relerr(a, z[f"{tag}/hcs/F{v}"]))
    for op, name in enumerate(JAC_OPS):
        J = ctx.eval_jacobian(op)
        g = z[f"{tag}/hcs/{name}"]
        sc = np.max(np.abs(g))
        for n in range(5):
            worst = max(worst, relerr(J[n], g[n], sc))
    ctx.close()
    return worst


def check_mms_hcs(tag):
    z = load("golden_hcs.npz")
    cname = tag.split("/")[1]
    ctx = make_ctx(z, tag, _ffi.REACT_HCSTRIPLE, CASE_ID[cname], CASE_PARAMS[cname])
    worst = 0.0
    for it in (0, 1):
        t = float(z[f"{tag}/t{it}/t"])
        fscale = max(1.0, np.max(np.abs(z[f"{tag}/t{it}/cp"])))
        for v, a in zip(("fcp", "fT", "fcl", "fcd", "fcs"), ctx.eval_forcing(t)):
            worst = max(worst, relerr(a, z[f"{tag}/t{it}/hcs/{v}"], fscale))
    ctx.close()
    return worst


def run_steps_hcs(tag, options=None):
    """Integrate a HCsTriple step fixture; same return values as run_steps."""
    z = load("golden_hcs.npz")
    kind = tag.split("/")[1]
    cname = str(z[f"{tag}/case"]) if f"{tag}/case" in z.files else tag.split("/")[2]
    ctx = make_ctx(z, tag, _ffi.REACT_HCSTRIPLE, CASE_ID[cname], CASE_PARAMS[cname])
    for name, value in (options or {}).items():
        ctx.set_option(name, value)
    ctx.set_state(*[z[f"{tag}/s0/{v}"] for v in VARS])
    t, dt = float(z[f"{tag}/t0"]), float(z[f"{tag}/dt"])
    saved = sorted(int(k.split("/")[-2][1:]) for k in z.files
                   if k.startswith(tag + "/s") and k.endswith("/cp") and "/s0/" not in k)
    is_pc = kind.startswith("pc")
    kw = {name: int(z[f"{tag}/{name}"]) for name in ("num_pc_steps", "num_newton_steps") if f"{tag}/{name}" in z.files}
    worst, wres = 0.0, 0.0
    for k in range(1, max(saved) + 1):
        if is_pc:
            ctx.step_pc(t, dt, cs_newton_iterations=0, **kw)
        else:
            ctx.step_feuler(t, dt)
        t += dt
        if k in saved:
            out = ctx.get_state()
            scale = max(np.max(np.abs(z[f"{tag}/s{k}/{v}"])) for v in VARS)
            for v, a in zip(VARS, out):
                worst = max(worst, np.max(np.abs(a - z[f"{tag}/s{k}/{v}"])) / scale)
            if is_pc:
                for v, a in zip(("T", "cl", "cd"), ctx.get_residuals()):
                    wres = max(wres, np.max(np.abs(a - z[f"{tag}/s{k}/res_{v}"])))
    stats = ctx.get_stats() if is_pc else {}
    ctx.close()
    return worst, wres, stats


def check_hcs_denominator_error():
    """A step whose corrector denominator 2 - dt Kd (Sd - cd1)(1 + cl1) is not positive: the library
    reports ValueError at the first call that synchronises (the reference raises it inside step())."""
    z = load("golden_hcs.npz")
    tag = "steps/pc_hcs/pol/N5"
    consts = np.array(z[f"{tag}/consts"])
    consts[_ffi.CONST_NAMES.index("Kd")] = 1e4
    ctx = _ffi.Context(z[f"{tag}/x"], z[f"{tag}/y"], consts, model_kind=2, reaction_kind=_ffi.REACT_HCSTRIPLE,
                       eta=0.0, mms_case=_ffi.MMS_POL)
    ctx.set_state_exact(0.0)
    ctx.step_pc(0.0, 0.1, cs_newton_iterations=0, jacobi_sweeps=(50, 50, 50))
    try:
        ctx.get_stats()
    except ValueError as e:
        assert "positiveness" in str(e)
    else:
        raise AssertionError("no ValueError")
    ctx.get_stats()  # the condition is reported once
    ctx.close()


def check_prefetch_forcing():
    """tedd_prefetch_forcing between steps (what the slab stepper does while ghost rows travel) leaves
    every result as it is, whichever way the forcing reaches the step kernels."""
    z = load("golden_steps.npz")
    tag = "pc_reg/expsin/N8"
    x, y, consts = z[f"{tag}/x"], z[f"{tag}/y"], z[f"{tag}/consts"]
    dt = float(z[f"{tag}/dt"])
    out = {}
    for name, opts, prefetch in (("plain", {}, False), ("fused+prefetch", {}, True), ("launch", {"fuse_forcing": 0}, False),
                                 ("launch+prefetch", {"fuse_forcing": 0}, True),
                                 ("in_kernels+prefetch", {"cache_forcing": 0}, True)):
        ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=float(z[f"{tag}/eta"]),
                           mms_case=_ffi.MMS_EXPSIN)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_state_exact(0.0)
        t = 0.0
        for _ in range(4):
            ctx.step_pc(t, dt, jacobi_sweeps=(30, 20, 20))
            t1 = t + dt
            if prefetch:
                ctx.prefetch_forcing(t1 + dt, t1)
                ctx.prefetch_forcing(t1 + dt, t1)  # a second call finds the level in place
            t = t1
        out[name] = ctx.get_state()
        ctx.close()
    for name in out:
        for a, b in zip(out[name], out["plain"]):
            np.testing.assert_array_equal(a, b, err_msg=name)


# ---------------------------------------------------------------------------------- expression programs
def expsin_programs():
    """Mesh, model constants and the node / cell programs lowered from the expressions of the ExpSin case
    (reference src/prob1_mms_cases.py:280-337, the time integrals written out) on a non-uniform mesh with
    exaggerated model constants."""
    sympy = pytest.importorskip("sympy")
    from tedd import _symprog as sp
    z = load("golden_mms.npz")
    tag = "expsin/nu5x7/stress2"   # a non-uniform mesh, exaggerated model constants
    consts = np.array(z[f"{tag}/consts"])
    x, y = z[f"{tag}/x"], z[f"{tag}/y"]
    c = dict(zip(_ffi.CONST_NAMES, consts))
    t, X, Y = sympy.symbols("t x y", real=True)
    W = sympy.sin(sympy.pi * X) * sympy.sin(sympy.pi * Y)
    a = 2 * sympy.pi ** 2 * sympy.Float(c["DT"])
    T = sympy.exp(-a * t) * W
    cl = -sympy.exp(-t) * W
    cd = sympy.exp(-t) * W
    s_ = sympy.Symbol("s", real=True)
    cp = W * sympy.exp(sympy.integrate((-c["K1"] * (1 + cl) - c["K2"] * T).subs(t, s_), (s_, 0, t)))
    cs = c["r_sp"] * W * sympy.exp(sympy.integrate((-c["Kd"] * (c["Sd"] - cd) * (1 + cl)).subs(t, s_), (s_, 0, t)))
    named = {}
    for name, e in {"cp": cp, "T": T, "cl": cl, "cd": cd, "cs": cs}.items():
        dx, dy = sympy.diff(e, X), sympy.diff(e, Y)
        named.update({name: e, "dt_" + name: sympy.diff(e, t), "dx_" + name: dx, "dy_" + name: dy,
                      "lap_" + name: sympy.diff(dx, X) + sympy.diff(dy, Y)})
    node, cell = sp.lower_case(named, t, X, Y)
    return x, y, consts, node, cell


def check_mms_program(react):
    """TEDD_MMS_PROGRAM through the C ABI: the exact solution, its derivatives, the five forcing arrays
    and a few steps from programs lowered from the expressions of the ExpSin case (reference
    src/prob1_mms_cases.py:280-337, here with the time integrals written out) equal the library's
    closed-form ExpSin case; malformed programs are refused."""
    from tedd import _symprog as sp
    x, y, consts, node, cell = expsin_programs()
    kw = dict(model_kind=2, reaction_kind=REACT_OF[react], eta=50.0 if react == "reg" else 0.0)
    prog = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_PROGRAM, **kw)
    ref = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_EXPSIN, **kw)
    with pytest.raises(ValueError):
        prog.eval_exact(0.1)  # no program yet
    prog.set_mms_program(0, *node)
    prog.set_mms_program(1, *cell)
    worst = 0.0
    for tt in (0.0, 0.37):
        for what in range(5):
            for p_, r_ in zip(prog.eval_mms(tt, what), ref.eval_mms(tt, what)):
                np.testing.assert_allclose(p_, r_, rtol=0, atol=2e-13, equal_nan=True)
        for p_, r_ in zip(prog.eval_forcing(tt), ref.eval_forcing(tt)):
            worst = max(worst, float(np.max(np.abs(p_ - r_))))
    assert worst < 1e-13, worst
    # steps, device-resident advance with norms, residuals
    for ctx in (prog, ref):
        ctx.set_state_exact(0.0)
    dt = 0.01
    ea = prog.advance(0.0, dt, 3, integrator=1)
    eb = ref.advance(0.0, dt, 3, integrator=1)
    np.testing.assert_allclose(ea, eb, rtol=1e-8, atol=1e-28)
    for p_, r_ in zip(prog.get_state(), ref.get_state()):
        np.testing.assert_allclose(p_, r_, rtol=0, atol=1e-14)
    for p_, r_ in zip(prog.get_residuals(), ref.get_residuals()):
        np.testing.assert_allclose(p_, r_, rtol=0, atol=1e-14)
    for ctx in (prog, ref):
        ctx.step_feuler(3 * dt, 1e-4)
    for p_, r_ in zip(prog.get_state(), ref.get_state()):
        np.testing.assert_allclose(p_, r_, rtol=0, atol=1e-14)
    # malformed programs
    code, cst, nreg = node
    bad = code.copy()
    bad[0] = (sp.OP_ADD) | (3 << 8) | (200 << 16) | (1 << 24)          # operand register out of range
    for args in ((0, bad, cst, nreg), (0, code, cst, 2), (0, code, cst, 200), (0, code, cst[:1], nreg),
                 (2, code, cst, nreg), (0, code[:-1], cst, nreg), (1, code, cst, nreg)):
        with pytest.raises(ValueError):
            prog.set_mms_program(*args)
    with pytest.raises(ValueError):
        ref.set_mms_program(0, *node)  # not a program context
    prog.close()
    ref.close()


def all_ops_programs():
    """Programs whose expressions use every operation of the program format (powers with integer, negative and
    real exponents, exp, sin, cos, sqrt, log, abs, sign, tanh, division)."""
    sympy = pytest.importorskip("sympy")
    from tedd import _symprog as sp
    t, X, Y = sympy.symbols("t x y", real=True)
    cp = sympy.tanh(X * Y) + sympy.sqrt(1 + t + X) / (2 + Y)
    # (abs only of the time: its time derivative is a sign, and no derivative of a sign is asked for)
    T = sympy.log(2 + X * Y + t) * sympy.cos(3 * X) - sympy.Abs(t - sympy.Rational(1, 3)) * X * Y
    cl = (1 + X) ** sympy.Rational(5, 2) * sympy.exp(-t) + (Y - sympy.Rational(2, 5)) ** 4
    cd = (1 + X * X + Y) ** (-2) + sympy.sin(t + Y) ** 5
    cs = sympy.Rational(1, 10) + (X * (1 - X) * Y * (1 - Y)) ** 2 / (1 + t) ** 3
    named = {}
    for name, e in {"cp": cp, "T": T, "cl": cl, "cd": cd, "cs": cs}.items():
        dx, dy = sympy.diff(e, X), sympy.diff(e, Y)
        named.update({name: e, "dt_" + name: sympy.diff(e, t), "dx_" + name: dx, "dy_" + name: dy,
                      "lap_" + name: sympy.diff(dx, X) + sympy.diff(dy, Y)})
    node, cell = sp.lower_case(named, t, X, Y)
    used = {int(w) & 0xff for w in node[0]} | {int(w) & 0xff for w in cell[0]}
    for op in ("OP_ADD", "OP_SUB", "OP_MUL", "OP_DIV", "OP_POWI", "OP_POW", "OP_EXP", "OP_SIN", "OP_COS", "OP_SQRT",
               "OP_LOG", "OP_ABS", "OP_SIGN", "OP_TANH"):
        assert getattr(sp, op) in used, op
    return node, cell


def check_generated_kernels_equal_interpreter(react, which):
    """The expression programs compiled at run time (option prog_codegen = 1: generated kernels or an error)
    produce, bit for bit, the arrays of the interpreter (prog_codegen = 0): exact solution and derivatives,
    forcing arrays, states and error norms after a few steps."""
    if which == "expsin":
        x, y, consts, node, cell = expsin_programs()
    else:
        x, y, consts, _, _ = expsin_programs()
        node, cell = all_ops_programs()
    kw = dict(model_kind=2, reaction_kind=REACT_OF[react], eta=50.0 if react == "reg" else 0.0)
    ctxs = []
    for codegen in (1, 0):
        c = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_PROGRAM, **kw)
        c.set_option("prog_codegen", codegen)
        c.set_mms_program(0, *node)
        c.set_mms_program(1, *cell)
        ctxs.append(c)
    gen, interp = ctxs
    for tt in (0.0, 0.37):
        for what in range(5):
            for p_, r_ in zip(gen.eval_mms(tt, what), interp.eval_mms(tt, what)):
                np.testing.assert_array_equal(p_, r_)
        for p_, r_ in zip(gen.eval_forcing(tt), interp.eval_forcing(tt)):
            assert np.all(np.isfinite(r_))
            np.testing.assert_array_equal(p_, r_)
    for c in ctxs:
        c.set_state_exact(0.0)
    ea = gen.advance(0.0, 0.01, 3, integrator=1)
    eb = interp.advance(0.0, 0.01, 3, integrator=1)
    np.testing.assert_array_equal(ea, eb)
    for p_, r_ in zip(gen.get_state(), interp.get_state()):
        np.testing.assert_array_equal(p_, r_)
    # new programs replace the compiled ones
    x2, y2, consts2, node2, cell2 = expsin_programs()
    gen.set_mms_program(0, *node2)
    gen.set_mms_program(1, *cell2)
    interp.set_mms_program(0, *node2)
    interp.set_mms_program(1, *cell2)
    for p_, r_ in zip(gen.eval_forcing(0.2), interp.eval_forcing(0.2)):
        np.testing.assert_array_equal(p_, r_)
    for c in ctxs:
        c.close()


def check_residuals_survive_other_calls():
    """A step whose predictor launch makes b_T itself does not store Y_T (nothing in the step reads it);
    tedd_get_residuals remakes it from the old cp and T, which stay out of the buffer pool until the next step.
    Calls in between that take buffers from the pool (field evaluation, norms, forcing arrays) must not disturb
    that: the residuals asked for twice are the same arrays, equal to those of the per-sweep solver (which stores
    Y_T) to the rounding of the two solvers, and small (reference tests/test_newton_residuals.py:116-124)."""
    N = 40
    x = np.linspace(0, 1, N + 1) ** 1.1
    y = np.linspace(0, 1, 34)
    consts = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
              1.60217662e-19 / 8.3144621, 5e-2, 300.0]
    out = {}
    for name, opts in (("chain", {}), ("sweeps", {"strip_solve": 0})):
        ctx = _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                           mms_case=_ffi.MMS_EXPSIN)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_state_exact(0.0)
        dt = 1e-3
        for k in range(3):
            ctx.step_pc(k * dt, dt)
        first = ctx.get_residuals()
        ctx.eval_field(3 * dt)
        ctx.error_norms(3 * dt)
        ctx.eval_forcing(0.5)
        ctx.eval_jacobian(5)
        second = ctx.get_residuals()
        for a, b in zip(first, second):
            np.testing.assert_array_equal(a, b)
        out[name] = first
        ctx.step_pc(3 * dt, dt)            # ... and the next step finds its buffers
        assert all(np.isfinite(a).all() for a in ctx.get_residuals())
        ctx.close()
    for a, b in zip(out["chain"], out["sweeps"]):
        assert np.max(np.abs(a - b)) < 1e-14
        assert np.max(np.abs(a[1:-1, 1:-1])) < 1e-9


def check_skewed_header_tree_is_refused(tmp_dir):
    """The generated kernels are compiled at run time against the headers beside the library (or under
    $TEDD_CSRC_DIR) and take their argument structures by value from the prebuilt library.  A header tree that is not
    the library's own -- here a copy with another contract version -- must fail to compile: prog_codegen = 1 reports
    it, prog_codegen = -1 keeps the interpreter and its arrays."""
    import shutil
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(_ffi.__file__))), "csrc")
    skew = os.path.join(str(tmp_dir), "csrc")
    shutil.copytree(csrc, skew)
    path = os.path.join(skew, "tedd_prog.hpp")
    text = open(path).read()
    assert "constexpr int kProgContractVersion = " in text
    import re
    text = re.sub(r"constexpr int kProgContractVersion = (\d+);",
                  lambda m: f"constexpr int kProgContractVersion = {int(m.group(1)) + 1000};", text)
    open(path, "w").write(text)
    x, y, consts, node, cell = expsin_programs()
    kw = dict(model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0)
    ref = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_PROGRAM, **kw)
    ref.set_option("prog_codegen", 0)
    ref.set_mms_program(0, *node)
    ref.set_mms_program(1, *cell)
    want = ref.eval_forcing(0.3)
    old = os.environ.get("TEDD_CSRC_DIR")
    os.environ["TEDD_CSRC_DIR"] = skew
    try:
        strict = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_PROGRAM, **kw)
        strict.set_option("prog_codegen", 1)
        strict.set_mms_program(0, *node)
        strict.set_mms_program(1, *cell)
        with pytest.raises(ValueError, match="not the ones it was built from"):
            strict.eval_forcing(0.3)
        strict.close()
        lenient = _ffi.Context(x, y, consts, mms_case=_ffi.MMS_PROGRAM, **kw)
        lenient.set_option("prog_codegen", -1)
        lenient.set_mms_program(0, *node)
        lenient.set_mms_program(1, *cell)
        for p_, r_ in zip(lenient.eval_forcing(0.3), want):
            np.testing.assert_array_equal(p_, r_)
        lenient.close()
    finally:
        if old is None:
            os.environ.pop("TEDD_CSRC_DIR", None)
        else:
            os.environ["TEDD_CSRC_DIR"] = old
    ref.close()


# ---------------------------------------------------------------------------------- sub-steps
REACT_OF = {"reg": _ffi.REACT_REGHCSTRIPLE, "cs3": _ffi.REACT_CSTRIPLE, "hcs": _ffi.REACT_HCSTRIPLE}


def check_substeps(tag, streamed=False, options=None):
    """Every piece of one predictor-corrector step called on its own through the C ABI, with the inputs the
    reference passed between them.  Returns (max error of a result relative to the field scale, max absolute
    error of a Newton residual).  streamed: forcing uploaded as arrays instead of evaluated in closed form."""
    z = load("golden_substeps.npz")
    kind = tag.split("/")[0]
    cname = str(z[f"{tag}/case"])
    rk = REACT_OF[kind]
    ctx = make_ctx(z, tag, rk, CASE_ID[cname], CASE_PARAMS[cname])
    for name, value in (options or {}).items():
        ctx.set_option(name, value)
    t0, dt = float(z[f"{tag}/t0"]), float(z[f"{tag}/dt"])
    S0 = [z[f"{tag}/s0/{v}"] for v in VARS]
    scale = max(np.max(np.abs(a)) for a in S0)
    forcing = None
    if streamed:
        forcing = [ctx.eval_forcing(t0), ctx.eval_forcing(t0 + dt)]
        ctx.close()
        ctx = make_ctx(z, tag, rk, _ffi.MMS_STREAM)

    def prepare(state):
        ctx.set_state(*state)
        if streamed:
            ctx.set_forcing(0, *forcing[0])
            ctx.set_forcing(1, *forcing[1])

    worst, wres = 0.0, 0.0

    def cmp(a, name):
        nonlocal worst
        worst = max(worst, np.max(np.abs(a - z[f"{tag}/{name}"])) / scale)

    prepare(S0)
    cp_pred, cs_pred = ctx.predict_cp_cs(t0, dt)
    cmp(cp_pred, "cp_pred")
    cmp(cs_pred, "cs_pred")
    lag = [z[f"{tag}/cp_pred"], S0[1], S0[2], S0[3], z[f"{tag}/cs_pred"]]
    prepare(lag)
    T1, res = ctx.newton_step(1, t0, dt, z[f"{tag}/YT"])
    cmp(T1, "T1")
    wres = max(wres, np.max(np.abs(res - z[f"{tag}/res_T"])))
    prepare(lag)
    cl1, res = ctx.newton_step(2, t0, dt, z[f"{tag}/Ycl"], z[f"{tag}/T1"])
    cmp(cl1, "cl1")
    wres = max(wres, np.max(np.abs(res - z[f"{tag}/res_cl"])))
    prepare(lag)
    cd1, res = ctx.newton_step(3, t0, dt, z[f"{tag}/Ycd"], z[f"{tag}/T1"], z[f"{tag}/cl1"])
    cmp(cd1, "cd1")
    wres = max(wres, np.max(np.abs(res - z[f"{tag}/res_cd"])))
    prepare(S0)
    cp_c, cs_c = ctx.correct_cp_cs(t0, dt, z[f"{tag}/T1"], z[f"{tag}/cl1"], z[f"{tag}/cd1"])
    cmp(cp_c, "cp_corr")
    cmp(cs_c, "cs_corr")
    ctx.close()
    return worst, wres


# ---------------------------------------------------------------------------------- verdict on the Jacobi solves
NB_A = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5,
        1.60217662e-19 / 8.3144621, 5e-2, 300.0]
# diffusivities large enough that a handful of sweeps is far from converged at dt = 1e-2 on a 40 x 40 mesh
STIFF = [8.3144621, 1.60217662e-19, 1e-3, 1e-3, 1e-3, 1e-3, 2e-2, 3e-3, 1e-5, 1e-9, 1e-2, 1.0, 1e-3, 1e-5,
         1.60217662e-19 / 8.3144621, 5e-2, 300.0]


def _ctx(N, M, consts, case=_ffi.MMS_EXPSIN, **kw):
    x = np.linspace(0, 1, N + 1) ** 1.05
    y = np.linspace(0, 1, M + 1)
    return _ffi.Context(x, y, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=case, **kw)


def check_under_converged_advance_raises():
    """Sweep counts left to the library but capped below what the tolerance needs (jacobi_max = 2): the steps
    run, the device keeps the worst last_update / (tol x increment), and tedd_advance reports it as
    TEDD_ERR_NOT_CONVERGED (RuntimeError) when it returns -- once; the check also clears the record."""
    ctx = _ctx(40, 33, STIFF)
    try:
        ctx.set_state_exact(0.0)
        with pytest.raises(RuntimeError, match="missed its tolerance"):
            ctx.advance(0.0, 1e-2, 6, jacobi_max=2)
        st = ctx.get_stats()
        assert st["sweeps"] == [2, 2, 2]
        assert max(st["worst_ratio"]) == 0.0          # cleared by the failed check
        ctx.check_converged()                         # nothing new to report
        # the same run with the cap lifted converges and says nothing
        ctx.set_state_exact(0.0)
        series = ctx.advance(0.0, 1e-2, 6)
        assert np.isfinite(series).all()
        st = ctx.get_stats()
        assert all(0.0 <= r <= 1.0 for r in st["worst_ratio"]), st
        # stepping one call at a time: the verdict waits for tedd_check_converged
        ctx.set_state_exact(0.0)
        for k in range(3):
            ctx.step_pc(k * 1e-2, 1e-2, jacobi_max=2)
        assert max(ctx.get_stats()["worst_ratio"]) > 10.0
        with pytest.raises(RuntimeError, match="missed its tolerance"):
            ctx.check_converged()
    finally:
        ctx.close()


# ... and a heat conductivity for which the T system needs ~30 sweeps from zero
STIFF_T = list(STIFF)
STIFF_T[6] = 6e-2


def check_warm_start_same_answer_fewer_sweeps(options=None):
    """A run of steps with the sweep counts left to the library: from the second step on the Jacobi iteration
    starts from the last increment; once the record of a warm step is in (four steps later) a system keeps the
    warm start where it saves at least three sweeps (here: T) and returns to zero iterates where it does not
    (it reads two more arrays).  Every step meets the tolerance either way, and the states agree with those of
    a run from zero iterates to the tolerance of the solves."""
    out = {}
    for warm in (1, 0):
        ctx = _ctx(48, 40, STIFF_T)
        for name, value in (options or {}).items():
            ctx.set_option(name, value)
        ctx.set_option("warm_start", warm)
        ctx.set_state_exact(0.0)
        dt, nsteps = 2e-3, 10
        sweeps, started = [], []
        for k in range(nsteps):
            ctx.step_pc(k * dt, dt)
            st = ctx.get_stats()
            sweeps.append(st["sweeps"])
            started.append(st["warm_started"])
            assert all(r <= 1.0 for r in st["worst_ratio"]), (warm, k, st)
            for v in range(3):
                assert st["last_update"][v] <= 1e-13 * st["increment"][v], (warm, k, v, st)
        ctx.check_converged()
        out[warm] = (ctx.get_state(), sweeps, started)
        ctx.close()
    (Sw, sw_w, st_w), (Sc, sw_c, st_c) = out[1], out[0]
    assert st_c == [[False] * 3] * 10 and st_w[0] == [False] * 3 and all(s == [True] * 3 for s in st_w[1:5])
    assert sw_c[0] == sw_w[0]                                    # the first step has nothing to start from
    assert all(a <= b + 1 for sa, sb in zip(sw_w, sw_c) for a, b in zip(sa, sb))
    assert st_w[-1][0] and sw_w[-1][0] <= sw_c[-1][0] - 4        # T: warm start kept, at least 3 sweeps saved
    for v in (1, 2):                                             # cl, cd: whichever start, never more sweeps than from zero
        assert sw_w[-1][v] <= sw_c[-1][v]
    for a, b in zip(Sw, Sc):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-14)


def check_fixed_sweeps_are_plain_jacobi():
    """Sweep counts the caller fixes: the iteration starts from zero whatever came before, and the solve is not
    judged (no verdict, whatever the count)."""
    ctx = _ctx(24, 20, STIFF)
    try:
        ctx.set_state_exact(0.0)
        for k in range(3):
            ctx.step_pc(k * 1e-2, 1e-2, jacobi_sweeps=(3, 2, 2))
        st = ctx.get_stats()
        assert st["warm_started"] == [False] * 3 and st["sweeps"] == [3, 2, 2] and max(st["worst_ratio"]) == 0.0
        ctx.check_converged()
        a = ctx.get_state()
        ctx.set_option("warm_start", 0)
        ctx.set_state_exact(0.0)
        for k in range(3):
            ctx.step_pc(k * 1e-2, 1e-2, jacobi_sweeps=(3, 2, 2))
        for u, v in zip(a, ctx.get_state()):
            np.testing.assert_array_equal(u, v)
    finally:
        ctx.close()


def check_null_output_pointers_are_refused():
    """Entry points that write through caller pointers refuse null ones with TEDD_ERR_INVALID (-1)."""
    ctx = _ctx(12, 10, STIFF)
    ctx.set_state_exact(0.0)
    lib, h = ctx._lib, ctx._h
    assert lib.tedd_state_device_ptrs(h, None) == -1
    assert b"null" in lib.tedd_last_error(h)
    assert lib.tedd_profile_get(h, 4, None, None, None) == -1
    assert lib.tedd_profile_get(h, 0, None, None, None) == 0      # (room for nothing: nothing written)
    ctx.close()


def check_failed_create_releases_what_it_took():
    """A context whose construction fails reports through tedd_last_error(NULL) and leaves the library usable."""
    x = np.linspace(0, 1, 9)
    for _ in range(3):
        with pytest.raises(ValueError):
            _ffi.Context(x[::-1].copy(), x, STIFF, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                         mms_case=_ffi.MMS_EXPSIN)
    ctx = _ctx(12, 10, STIFF)
    ctx.set_state_exact(0.0)
    ctx.close()
