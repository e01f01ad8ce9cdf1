"""A/B of library options inside ONE context (the same memory, the same process): per-family kernel times at
16384^2 for option sets given as  name=value,name=value  arguments, visited in turn, several rounds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

if os.environ.get("AB_LIBRARY"):  # (a probe build of the library, e.g. one whose predictor body is stripped)
    _ffi.use_library(os.environ["AB_LIBRARY"])
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
N = int(os.environ.get("AB_MESH", "16384"))
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
ctx.set_state_exact(0.0)
ctx.advance(0.0, dt, 8, want_errors=False)
ctx.sync()
k = 8
defaults = {}
for s in sets:
    for name in s:
        defaults[name] = -1.0
for rnd in range(3):
    for s in sets:
        for name, v in {**defaults, **s}.items():
            ctx.set_option(name, v)
        def step():
            try:
                ctx.step_pc(k * dt, dt)
            except Exception as e:  # (a probe library computes nonsense: the launches are timed all the same)
                if not os.environ.get("AB_LIBRARY"):
                    raise
                print("step failed:", str(e)[:80], flush=True)
        step()
        k += 1
        ctx.profile_enable(True)
        for _ in range(4):
            step()
            k += 1
        ctx.sync()
        prof = ctx.profile_get()
        ctx.profile_enable(False)
        line = " ".join(f"{f}:{v['total_ms'] / v['launches']:.2f}" for f, v in prof.items())
        print(f"round {rnd} {s}: {line} | sum {sum(v['total_ms'] for v in prof.values()) / 4:.2f}", flush=True)
ctx.close()
