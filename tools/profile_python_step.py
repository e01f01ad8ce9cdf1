"""cProfile of integrator.step() called one by one through the mirrored classes (host-side cost of the
reference-style loop):  python tools/profile_python_step.py [N] [closed|program]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tedd.prob1_mms_cases as p1mc
import tedd.prob1base as p1
from api_cases import model_of

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "closed"
model = model_of("nb_b")
grid = p1.make_uniform_grid(N, N)
if kind == "closed":
    case = p1mc.MMSCasePol(grid=grid, model=model)
else:
    u = p1.x_sym * (1 - p1.x_sym) * p1.y_sym * (1 - p1.y_sym) / (1 + p1.t_sym)
    case = p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=u, T_sym_expr=u, cl_sym_expr=u, cd_sym_expr=u, cs_sym_expr=u)
F = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=50.0)
fld = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=F, regularization_factor=50.0)
integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(fld, regularization_factor=50.0)
S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
dt = (1.0 / N) ** 1.5
S = integ.step(S, t0=0.0, dt=dt)
pr = cProfile.Profile()
pr.enable()
t = dt
for _ in range(100):
    S = integ.step(S, t0=t, dt=dt)
    t += dt
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
