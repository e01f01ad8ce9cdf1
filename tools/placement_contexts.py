"""Which device memory makes the predictor launch slow?  Contexts at 16384^2 (69 GB each) created in ONE process,
priming off, some kept alive while the next is created so that the next one cannot be given their memory:
  A  first allocation of the process
  B  created while A is alive (memory A never had)
  C  created after A was destroyed, B still alive (A's memory can come back)
  D  created after B and C were destroyed
Prints the predictor's and the whole step's kernel milliseconds per context (HIP events around every launch)."""
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
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5


def make():
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 4, want_errors=False)
    ctx.sync()
    return ctx


def measure(tag, ctx):
    out = []
    for k in range(steps):
        ctx.profile_enable(True)
        ctx.step_pc((4 + k) * dt, dt)
        ctx.sync()
        prof = ctx.profile_get()
        ctx.profile_enable(False)
        out.append((prof.get("predict", {"total_ms": 0.0})["total_ms"], sum(v["total_ms"] for v in prof.values())))
    print(f"{tag}: predict ms", " ".join(f"{a:.2f}" for a, _ in out), "| step ms", " ".join(f"{b:.1f}" for _, b in out),
          flush=True)


A = make()
measure("A (first allocation)", A)
B = make()
measure("B (while A alive)", B)
measure("A again (B alive)", A)
A.close()
C = make()
measure("C (after A freed, B alive)", C)
measure("B again", B)
B.close()
C.close()
D = make()
measure("D (after all freed)", D)
D.close()
