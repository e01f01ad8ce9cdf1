"""Device-resident steps/s on small meshes for sets of library options:
    python tools/study_mesh_variants.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi


def run(opts, N=256, steps=2000):
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE,
                       eta=50.0, mms_case=_ffi.MMS_EXPSIN)
    for k, v in opts.items():
        ctx.set_option(k, v)
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 50, want_errors=False)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.advance(50 * dt, dt, steps, want_errors=False)
    ctx.sync()
    el = time.perf_counter() - t0
    ctx.close()
    return steps / el


if __name__ == "__main__":
    meshes = [int(a) for a in sys.argv[1:]] or [256, 1024]
    for N in meshes:
        for o in ({}, {"strip_shape": 0}, {"strip_shape": 1}, {"strip_shape": 2}, {"fuse_forcing": 0}, {"cache_forcing": 0},
                  {"strip_solve": 0}):
            print(N, o, round(run(o, N, max(100, min(2000, 2000 * 256 * 256 // (N * N) * 8))), 1), "steps/s", flush=True)
