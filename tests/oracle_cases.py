"""The library (through the C ABI) against the CPU oracle on seeded random inputs that no fixture holds:
random non-uniform meshes of odd shapes, random states, random forcing arrays streamed in.  Run on the
host twin (CPU suite) and on the GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tedd_oracle as orc  # noqa: E402

from tedd import _ffi  # noqa: E402

STRESS = [8.3144621, 1.60217662e-19, 0.11, 0.23, 0.37, 0.19, 2e-2, 3e-2, 0.4, 0.03, 0.2, 1.5, 1e-2, 0.3, 50.0, 5e-2, 300.0]
REACTS = {"reg": _ffi.REACT_REGHCSTRIPLE, "cs3": _ffi.REACT_CSTRIPLE, "hcs": _ffi.REACT_HCSTRIPLE}


def _axis(rng, n):
    w = 0.5 + rng.random(n)
    return np.concatenate([[0.0], np.cumsum(w) / np.sum(w)])


def check_random_step(react, seed, N, M, options=None):
    """Field evaluation, one forward-Euler step and two predictor-corrector steps (2 Newton steps, 2 PC
    iterations in the second) with random streamed forcing; returns the worst deviation relative to the
    field scale."""
    rng = np.random.default_rng(seed)
    x, y = _axis(rng, N), _axis(rng, M)
    shp = (N + 1, M + 1)
    eta = 7.5 if react == "reg" else 0.0
    S = {"cp": rng.random(shp) + 0.1, "T": rng.random(shp), "cl": rng.random(shp), "cd": rng.random(shp),
         "cs": rng.random(shp) * (1.0 if react == "cs3" else 0.4) - (0.0 if react == "cs3" else 0.1)}
    f = [{k: 0.3 * (rng.random(shp) - 0.5) for k in ("fcp", "fT", "fcl", "fcd", "fcs")} for _ in range(3)]
    for fk in f:
        fk["fcp"][0, :] = fk["fcp"][-1, :] = 0.0   # cell averages are 0 on the boundary
        fk["fcp"][:, 0] = fk["fcp"][:, -1] = 0.0
    mesh, model = orc.Mesh(x, y), orc.Model(STRESS, 2)
    pb = orc.Problem(mesh, model, None, react, eta)
    ctx = _ffi.Context(x, y, STRESS, model_kind=2, reaction_kind=REACTS[react], eta=eta, mms_case=_ffi.MMS_STREAM)
    for k, v in (options or {}).items():
        ctx.set_option(k, v)
    order = ("fcp", "fT", "fcl", "fcd", "fcs")
    worst = 0.0

    def compare(dev, ref):
        nonlocal worst
        for a, v in zip(dev, orc.VARS):
            worst = max(worst, float(np.max(np.abs(a - ref[v])) / max(1.0, np.max(np.abs(ref[v])))))

    ctx.set_state(*[S[v] for v in orc.VARS])
    ctx.set_forcing(0, *[f[0][k] for k in order])
    compare(ctx.eval_field(0.0), orc.field_all(mesh, model, S, f[0], react, eta))
    dt = 0.004
    ctx.step_feuler(0.0, dt)
    compare(ctx.get_state(), orc.step_forward_euler(pb, S, 0.0, dt, forcing0=f[0]))
    # predictor-corrector from the same random state
    ctx.set_state(*[S[v] for v in orc.VARS])
    ctx.set_forcing(0, *[f[0][k] for k in order])
    ctx.set_forcing(1, *[f[1][k] for k in order])
    ctx.step_pc(0.0, dt, jacobi_tol=1e-14)
    S1, res = orc.step_pc(pb, S, 0.0, dt, forcing0=f[0], forcing1=f[1])
    compare(ctx.get_state(), S1)
    for a, v in zip(ctx.get_residuals(), ("T", "cl", "cd")):
        worst = max(worst, float(np.max(np.abs(a - res[v]))))
    ctx.set_state(*[S1[v] for v in orc.VARS])
    ctx.set_forcing(0, *[f[1][k] for k in order])
    ctx.set_forcing(1, *[f[2][k] for k in order])
    ctx.step_pc(dt, dt, num_pc_steps=2, num_newton_steps=2, jacobi_tol=1e-14)
    S2, _ = orc.step_pc(pb, S1, dt, dt, num_pc_steps=2, num_newton_steps=2, forcing0=f[1], forcing1=f[2])
    compare(ctx.get_state(), S2)
    ctx.close()
    return worst
