import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tedd import _ffi
rng = np.random.default_rng(7)
sizes = [(5, 5), (31, 33), (43, 45), (44, 44), (45, 43), (51, 53), (52, 56), (63, 64), (64, 65), (65, 63), (87, 89),
         (88, 104), (127, 129), (128, 128), (131, 257), (200, 37), (37, 200), (300, 301)]
for N, M in sizes:
    x = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(N))]); x /= x[-1]
    y = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(M))]); y /= y[-1]
react = _ffi.REACT_CSTRIPLE
def run(opts, nsteps=2, sweeps=(23, 11, 7)):
    ctx = _ffi.Context(x, y, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=react, eta=0.0, mms_case=_ffi.MMS_EXPSIN)
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_state_exact(0.0)
    dt = 2e-3
    for k in range(nsteps): ctx.step_pc(k * dt, dt, jacobi_sweeps=sweeps)
    out = ctx.get_state(); ctx.close(); return out
for nsteps in (1, 2):
    a = run({"strip_shape": 0, "strip_hmax": 1}, nsteps); b = run({"strip_shape": 1, "strip_hmax": 1}, nsteps)
    for name, u, v in zip(("cp", "T", "cl", "cd", "cs"), a, b):
        idx = np.argwhere(u != v)
        print(nsteps, name, len(idx), idx[:6].tolist(), [(u[tuple(i)], v[tuple(i)]) for i in idx[:2]])
