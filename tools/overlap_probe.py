"""Does a forcing launch on a second HIP stream overlap with the step of another context?
Context A steps normally; context B (own stream) evaluates one forcing set per step of A.
    python tools/overlap_probe.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = np.linspace(0, 1, N + 1)
mk = lambda: _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE,
                          eta=50.0, mms_case=_ffi.MMS_EXPSIN)
A, B = mk(), mk()
for c in (A, B):
    c.set_option("fuse_forcing", 0)
dt = (1.0 / N) ** 1.5
A.set_state_exact(0.0)


def run(steps, with_b, t):
    A.sync(); B.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        if with_b:
            B.prefetch_forcing(t + 2 * dt + 1e-3 * (k + 1), t + 1e-3 * k + 2 * dt if k else -1.0)
        A.step_pc(t, dt)
        t += dt
    A.sync(); B.sync()
    return (time.perf_counter() - t0) / steps * 1e3, t


_, t = run(3, True, 0.0)
a, t = run(10, False, t)
b, t = run(10, True, t)
c, t = run(10, False, t)
print(f"N={N}: A alone {a:.2f} ms/step, A + forcing launch of B on its own stream {b:.2f} ms/step, A alone {c:.2f}")
