"""ms per predictor-corrector step of the three reaction kinds and the four closed-form cases on one mesh:
    python tools/react_kinds_speed.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from tedd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = np.linspace(0, 1, N + 1)
dt = (1.0 / N) ** 1.5
params = {_ffi.MMS_PEAKS: (1.0, 10.0), _ffi.MMS_NFS: (2.1, 3.1, 3.1, 3.1, 2.1, 1 / np.pi)}
for case, cname in ((_ffi.MMS_EXPSIN, "expsin"), (_ffi.MMS_POL, "pol"), (_ffi.MMS_PEAKS, "peaks"), (_ffi.MMS_NFS, "nfs")):
    for react, rname in ((_ffi.REACT_REGHCSTRIPLE, "reg"), (_ffi.REACT_CSTRIPLE, "cs3"), (_ffi.REACT_HCSTRIPLE, "hcs")):
        ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=react,
                           eta=50.0 if rname == "reg" else 0.0, mms_case=case, mms_params=params.get(case, ()))
        ctx.set_state_exact(0.0)
        ctx.advance(0.0, dt, 3, want_errors=False)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.advance(3 * dt, dt, 10, want_errors=False)
        ctx.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        st = ctx.get_stats()
        print(f"N={N} {cname:6s} {rname}: {ms:7.3f} ms/step  sweeps {st['sweeps']}", flush=True)
        ctx.close()
