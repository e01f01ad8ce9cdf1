"""Per-kernel times of the 16384^2 step with the field buffers carved from one allocation at different distances
(TEDD_FIELD_GAP_MB: bytes between the end of one field and the start of the next), contexts created one after the
other in one process.  usage: placement_gap_series.py [N] [steps] gap_MB ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gaps = sys.argv[3:] or ["none", "0", "1024", "2048"]
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
fams = ["predict", "forcing", "solve_T", "rhs_cl", "solve_cl", "rhs_cd", "solve_cd", "cs_newton"]
print("gap_MB   step_ms  " + " ".join(f"{f:>9s}" for f in fams), flush=True)
for gap in gaps:
    if gap == "none":
        os.environ.pop("TEDD_FIELD_GAP_MB", None)
    else:
        os.environ["TEDD_FIELD_GAP_MB"] = gap
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 8, want_errors=False)
    ctx.sync()
    ctx.profile_enable(True)
    for k in range(steps):
        ctx.step_pc((8 + k) * dt, dt)
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    tot = sum(v["total_ms"] for v in prof.values()) / steps
    print(f"{gap:>6s}  {tot:8.2f}  " + " ".join(f"{prof.get(f, {'total_ms': 0.0})['total_ms'] / steps:9.3f}" for f in fams), flush=True)
    ctx.close()
