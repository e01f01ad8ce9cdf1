"""Device-resident steps with the fused error norms (tedd_advance with a series) on a small mesh, for
kernel traces:  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/trace_small_norms.py [N] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x = np.linspace(0, 1, N + 1)
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
dt = (1.0 / N) ** 1.5
ctx.set_state_exact(0.0)
ctx.advance(0.0, dt, 20, integrator=1)
t0 = time.perf_counter()
ctx.advance(0.0, dt, steps, integrator=1)
el = time.perf_counter() - t0
print(f"N={N}: {steps / el:.1f} steps/s with norms")
ctx.close()
