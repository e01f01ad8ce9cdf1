"""Device-resident steps on a small study mesh, for kernel traces (rocprofv3 --kernel-trace):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/trace_small.py [N] [steps]
tools/trace_gaps.py summarises the trace (per-kernel duration, idle time between kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x = np.linspace(0, 1, N + 1)
ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                   mms_case=_ffi.MMS_EXPSIN)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, float(v))
dt = (1.0 / N) ** 1.5
ctx.set_state_exact(0.0)
ctx.advance(0.0, dt, steps, want_errors=False)
ctx.sync()
ctx.close()
