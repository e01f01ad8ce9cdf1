"""Per-kernel times of the 16384^2 step in contexts created one after the other in one process: which state does a
context land in -- predictor at 7.6 or at 8.2 ms -- and what does that follow?  Arguments, one context each: a
number (TEDD_FIELD_SKEW_KB for that context: only a PROBE build of the library reads it -- in arena_setup, allocate
512 MiB more and start the arena that many KB into the allocation, and print the allocation's address in the
placement note; the shipped library ignores it) or the word `stream` (create one more HIP stream and keep it).
Found with it: the state alternates with every context, whatever the shift and however many streams were made in
between, at the same virtual address of the arena.
usage: [AB_LIBRARY=<probe library>] placement_skew_series.py 0 0 0 0 stream 0 0 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

if os.environ.get("AB_LIBRARY"):
    _ffi.use_library(os.environ["AB_LIBRARY"])
N = 16384
steps = 5
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
fams = ["predict", "forcing", "solve_T", "rhs_cl", "solve_cl", "rhs_cd", "solve_cd", "cs_newton"]
print("skew_KB   step_ms  " + " ".join(f"{f:>9s}" for f in fams), flush=True)
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
dummies = []
for skew in sys.argv[1:]:
    if skew == "stream":  # an extra stream, kept: does the state follow the number of streams the process has made?
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        dummies.append(st)
        print("  (one more stream)", flush=True)
        continue
    os.environ["TEDD_FIELD_SKEW_KB"] = skew
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
    print(f"{skew:>8s}  {tot:8.2f}  " + " ".join(f"{prof.get(f, {'total_ms': 0.0})['total_ms'] / steps:9.3f}" for f in fams)
          + "  " + ctx.placement.split("base")[-1], flush=True)
    ctx.close()
