"""What the Jacobi tolerance buys: states after a few steps with jacobi_tol = 1e-13 (default, relative to the
solution increment) and looser values, compared entry by entry, and ms per step.
    python tools/jacobi_tol_probe.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5


def run(tol, steps=5):
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 2, want_errors=False, jacobi_tol=tol)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.advance(2 * dt, dt, steps, want_errors=False, jacobi_tol=tol)
    ctx.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = ctx.get_stats()
    out = ctx.get_state()
    ctx.close()
    return ms, st, out


ms0, st0, ref = run(1e-13)
print(f"N={N} tol 1e-13: {ms0:.3f} ms/step sweeps {st0['sweeps']} increments {st0['increment']}")
for tol in (1e-10, 1e-8, 1e-6):
    ms, st, out = run(tol)
    worst = max(float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) for a, b in zip(out, ref))
    print(f"N={N} tol {tol:g}: {ms:.3f} ms/step sweeps {st['sweeps']}  max relative difference of the state to the 1e-13 run: {worst:.2e}")
