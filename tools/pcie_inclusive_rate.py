import sys, os, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import tedd.prob1_mms_cases as p1mc, tedd.prob1base as p1
from api_cases import model_of
for N, steps in ((256, 200), (4096, 5), (16384, 3)):
    model = model_of("nb_a"); grid = p1.make_uniform_grid(N, N)
    case = p1mc.MMSCaseExpSin(grid=grid, model=model)
    F = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=50.0)
    fld = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=F, regularization_factor=50.0)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(fld, regularization_factor=50.0)
    S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    dt = (1.0 / N) ** 1.5
    S = integ.step(S, t0=0.0, dt=dt); t = dt
    t0 = time.perf_counter()
    for _ in range(steps):
        S = integ.step(S, t0=t, dt=dt); t += dt
    el = (time.perf_counter() - t0) / steps
    print(f"N={N}: integrator.step() with StateVars in/out: {el*1e3:.2f} ms/step = {1/el:.2f} steps/s", flush=True)
    del S, integ, fld, F, case
