"""Predictor time at 16384^2 on the context's own stream, on the null stream, and with torch initialised in
the process (what bench.py and the slab stepper do)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
stream = None
if "torch" in mode:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    if "tstream" in mode:
        stream = torch.cuda.current_stream().cuda_stream
    if "side" in mode:
        side = torch.cuda.Stream()
        stream = side.cuda_stream
if "null" in mode:
    stream = 0
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
if stream is not None:
    ctx.set_stream(stream)
ctx.set_state_exact(0.0)
ctx.advance(0.0, dt, 8, want_errors=False)
ctx.sync()
pred, tot = [], []
for k in range(6):
    ctx.profile_enable(True)
    ctx.step_pc((8 + k) * dt, dt)
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    pred.append(prof.get("predict", {"total_ms": 0.0})["total_ms"])
    tot.append(sum(v["total_ms"] for v in prof.values()))
print(f"mode {mode} (stream {stream}): predict ms per step:", " ".join(f"{a:.2f}" for a in pred), "| all kernels:",
      " ".join(f"{b:.1f}" for b in tot), flush=True)
ctx.close()
