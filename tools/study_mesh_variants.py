import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tedd import _ffi
import numpy as np, time
def run(opts, N=256, steps=2000):
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, x, bench.consts_vec(bench.NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
    for k,v in opts.items(): ctx.set_option(k,v)
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0); ctx.advance(0.0, dt, 50, want_errors=False); ctx.sync()
    t0=time.perf_counter(); ctx.advance(50*dt, dt, steps, want_errors=False); ctx.sync(); el=time.perf_counter()-t0
    ctx.close(); return steps/el
for N in (256, 1024):
    for o in ({}, {"fuse_forcing":0}, {"cache_forcing":0}, {"strip_solve":0}):
        print(N, o, round(run(o, N, 2000 if N==256 else 300),1), "steps/s")
