"""Times integrator.step of the reference itself (imported from /root/reference/src, so only where that tree
exists: the development container, not the GPU box) on the ExpSin N = 256 mesh of the bench's CPU baseline,
beside the oracle port on the same machine.  PYTHONPATH=/root/reference/src python tools/time_reference_here.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import prob1base as p1  # noqa: E402  (the reference's module)
import prob1_mms_cases as p1mc  # noqa: E402

import bench  # noqa: E402

N, steps = 256, 5
model = p1.DefaultModel02(mc=p1.ModelConsts(**bench.NB_A))
grid = p1.make_uniform_grid(N, N)
case = p1mc.MMSCaseExpSin(grid=grid, model=model)
forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=50.0)
field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing, regularization_factor=50.0)
integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(semi_discrete_field=field, regularization_factor=50.0)
dt = (1.0 / N) ** 1.5
S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
S = integ.step(S, t0=0.0, dt=dt)
t0 = time.perf_counter()
for k in range(steps):
    S = integ.step(S, t0=(k + 1) * dt, dt=dt)
ref = steps / (time.perf_counter() - t0)
print(f"reference integrator.step, N={N}: {ref:.3f} steps/s", flush=True)
base = bench.cpu_baseline_run()
print("oracle port (bench cpu_baseline):", base)
print(f"oracle / reference = {base['value'] / ref:.2f}")
