"""One predictor-corrector step on the smallest meshes (1 x 1: no interior node; 1 x 3, 3 x 1, 2 x 2) through the
C ABI on the GPU, checked against the exact boundary values.  Run before tests that use such meshes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import api_cases as ac  # noqa: E402
import tedd.prob1base as p1  # noqa: E402

model = ac.model_of("nb_b")
for N, M in ((1, 1), (1, 3), (3, 1), (2, 2), (1, 2)):
    grid = p1.make_uniform_grid(N, M)
    for name in ("peaks_fast1e1", "expsin"):
        cls = ac.CASES[name][0]
        case = cls(grid=grid, model=model)
        forcing = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=50.0)
        field = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=forcing, regularization_factor=50.0)
        integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(field, regularization_factor=50.0)
        S = integ.step(p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid), t0=0.0, dt=0.01)
        E = p1.state_from_mms_when(mms_case=case, t=0.01, grid=grid)
        err = max(float(np.max(np.abs(getattr(S, v) - getattr(E, v)))) for v in ("cp", "T", "cl", "cd", "cs"))
        print(f"{N}x{M} {name}: max |state - exact| = {err:.3e}", flush=True)
        assert np.isfinite(err) and err < 1e-2
print("ok")
