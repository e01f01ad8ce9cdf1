"""Per-step time of the predictor launch at 16384^2 (HIP events around every launch of the family) in several
contexts created one after the other in ONE process: the launch runs at one speed for the life of a context
and at another in the next process on the same machine; does a context that was created, stepped and destroyed
before change the speed of the next one?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
for r in range(rounds):
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 4, want_errors=False)
    ctx.sync()
    out = []
    for k in range(steps):
        ctx.profile_enable(True)
        ctx.step_pc((4 + k) * dt, dt)
        ctx.sync()
        prof = ctx.profile_get()
        ctx.profile_enable(False)
        out.append((prof.get("predict", {"total_ms": 0.0})["total_ms"], sum(v["total_ms"] for v in prof.values())))
    print(f"context {r}: predict ms per step:", " ".join(f"{a:.2f}" for a, _ in out), flush=True)
    print(f"context {r}: all kernels ms per step:", " ".join(f"{b:.1f}" for _, b in out), flush=True)
    ctx.close()
