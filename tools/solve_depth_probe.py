"""Time of the single-launch Newton solves as a function of their depth (fixed sweep counts, cold start).

    python3 tools/solve_depth_probe.py [N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = np.linspace(0, 1, N + 1)
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
dt = (1.0 / N) ** 1.5
ctx.set_state_exact(0.0)
for s in (1, 2, 3, 4, 6, 8, 10):
    ctx.set_state_exact(0.0)
    ctx.step_pc(0.0, dt, jacobi_sweeps=(s, s, s))
    ctx.profile_enable(True)
    for k in range(4):
        ctx.step_pc((k + 1) * dt, dt, jacobi_sweeps=(s, s, s))
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    print(s, " ".join("%s=%.3f" % (k, v["total_ms"] / v["launches"]) for k, v in prof.items() if k.startswith("solve")), flush=True)
ctx.close()

# the record of the warm-started solves of an ordinary run (sweep counts chosen by the library)
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
ctx.set_state_exact(0.0)
for k in range(14):
    ctx.step_pc(k * dt, dt)
    st = ctx.get_stats()
    print(k, "sweeps", st["sweeps"], "warm", st["warm_started"], "first/inc",
          ["%.2e" % (f / i if i else 0) for f, i in zip(st["first_update"], st["increment"])], "last/inc",
          ["%.2e" % (u / i if i else 0) for u, i in zip(st["last_update"], st["increment"])], flush=True)
ctx.close()
