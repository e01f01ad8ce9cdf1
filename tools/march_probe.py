"""Marching launcher against the plain one at one mesh size: bit-identity of a step and the time of the
predictor launch under several settings (march_occ, march_rows)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402
from tedd import _ffi  # noqa: E402


def run(N, opts, steps=6, get=False):
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    ctx.set_option("fuse_forcing", 0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0)
    sw = bench.pick_sweeps(N, dt, 1e-13)
    t = 0.0
    for _ in range(2):
        ctx.step_pc(t, dt, jacobi_sweeps=sw)
        t += dt
    ctx.profile_enable(True)
    for _ in range(steps):
        ctx.step_pc(t, dt, jacobi_sweeps=sw)
        t += dt
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    out = {k: round(v["total_ms"] / v["launches"], 3) for k, v in prof.items()}
    state = ctx.get_state() if get else None
    ctx.close()
    return out, state


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ref, s0 = run(N, {"march": 0}, get=N <= 4096)
    print("plain", json.dumps(ref), flush=True)
    for opts in ({"march": 1}, {"march": 1, "march_occ": 3, "march_ahead": 0}, {"march": 1, "march_occ": 4, "march_ahead": 1},
                 {"march": 1, "march_occ": 4, "march_ahead": 0}, {"march": 1, "march_rows": 128},
                 {"march": 1, "march_rows": 256}):
        out, s1 = run(N, opts, get=N <= 4096)
        same = None if s0 is None else all(np.array_equal(a, b) for a, b in zip(s0, s1))
        print(json.dumps(opts), {k: out.get(k) for k in ("predict", "rhs_cl", "rhs_cd")}, "bit-identical" if same else same, flush=True)
