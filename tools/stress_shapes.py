"""One-off stress of the solve chain's tiling: many mesh sizes around the tile edges, every tile shape and
several chain depths, against the one-launch-per-sweep solver on the same inputs (bitwise between shapes,
1e-14 of the field scale against the per-sweep solver).  python tools/stress_shapes.py"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

sizes = [(5, 5), (31, 33), (43, 45), (44, 44), (45, 43), (51, 53), (52, 56), (63, 64), (64, 65), (65, 63), (87, 89),
         (88, 104), (127, 129), (128, 128), (131, 257), (200, 37), (37, 200), (300, 301)]
variants = [{"strip_shape": s, "strip_hmax": h} for s, h in itertools.product((0, 1, 2), (1, 2, 3, 5, 10))]
rng = np.random.default_rng(7)
bad = 0
for N, M in sizes:
    x = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(N))]); x /= x[-1]
    y = np.concatenate([[0.0], np.cumsum(0.5 + rng.random(M))]); y /= y[-1]
    for react in (_ffi.REACT_REGHCSTRIPLE, _ffi.REACT_CSTRIPLE, _ffi.REACT_HCSTRIPLE):
        def run(opts):
            ctx = _ffi.Context(x, y, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=react,
                               eta=50.0 if react == _ffi.REACT_REGHCSTRIPLE else 0.0, mms_case=_ffi.MMS_EXPSIN)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_state_exact(0.0)
            dt = 2e-3
            for k in range(2):
                ctx.step_pc(k * dt, dt, jacobi_sweeps=(23, 11, 7))
            out = ctx.get_state()
            ctx.close()
            return out
        ref = run({"strip_solve": 0})
        first = None
        for v in variants:
            out = run(v)
            if first is None:
                first = out
            d_ref = max(float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) for a, b in zip(out, ref))
            same = all(np.array_equal(a, b) for a, b in zip(out, first))
            if d_ref > 1e-14 or not same:
                bad += 1
                print("MISMATCH", N, M, react, v, d_ref, same, flush=True)
    print(f"{N}x{M} ok", flush=True)
print("mismatches:", bad)
