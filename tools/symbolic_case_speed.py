"""steps/s of a symbolic MMS case (expression programs on the device), of the same case with host-evaluated
streamed forcing, and of the built-in closed form with the same expressions:  python tools/symbolic_case_speed.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import tedd.prob1_mms_cases as p1mc
import tedd.prob1base as p1
from api_cases import model_of

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# second argument: library option prog_codegen for every program context (1 kernels compiled at run time or an
# error, 0 interpreter, default -1: compiled where possible)
if len(sys.argv) > 2:
    from tedd import _ffi
    _mode, _orig = float(sys.argv[2]), _ffi.Context.set_mms_program

    def _set_program(self, which, *a):
        self.set_option("prog_codegen", _mode)
        return _orig(self, which, *a)
    _ffi.Context.set_mms_program = _set_program
    print("prog_codegen =", _mode)
model = model_of("nb_b")
grid = p1.make_uniform_grid(N, N)
u = p1.x_sym * (1 - p1.x_sym) * p1.y_sym * (1 - p1.y_sym) / (1 + p1.t_sym)


def run(case, steps, fused):
    F = p1.ForcingTerms_RegHCsTriple(mms_case=case, model=model, regularization_factor=50.0)
    fld = p1.SemiDiscreteField_RegHCsTriple(grid=grid, model=model, forcing_terms=F, regularization_factor=50.0)
    integ = p1.P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(fld, regularization_factor=50.0)
    S = p1.state_from_mms_when(mms_case=case, t=0.0, grid=grid)
    dt = (1.0 / N) ** 1.5
    if fused:
        integ.advance_with_errors(S, t0=0.0, dt=dt, num_steps=5)
        t0 = time.perf_counter()
        integ.advance_with_errors(S, t0=0.0, dt=dt, num_steps=steps)
        return steps / (time.perf_counter() - t0)
    S = integ.step(S, t0=0.0, dt=dt)
    t0 = time.perf_counter()
    t = dt
    for _ in range(steps):
        S = integ.step(S, t0=t, dt=dt)
        t += dt
    return steps / (time.perf_counter() - t0)


sym = lambda dp: p1.MMSCaseSymbolic(grid=grid, model=model, cp_sym_expr=u, T_sym_expr=u, cl_sym_expr=u, cd_sym_expr=u,
                                    cs_sym_expr=u, device_program=dp)
print(f"N={N}")
print("closed form (MMSCasePol), device-resident run with norms :", round(run(p1mc.MMSCasePol(grid=grid, model=model), 400, True), 1), "steps/s")
print("expression programs,       device-resident run with norms :", round(run(sym(True), 400, True), 1), "steps/s")
print("expression programs,       integrator.step() one by one   :", round(run(sym(True), 50, False), 1), "steps/s")
print("host-evaluated forcing,    integrator.step() one by one   :", round(run(sym(False), 20, False), 1), "steps/s")
osc = p1mc.MMSCasePolWithOscilatingTime(grid=grid, model=model, ampl=1.0, speed=10.0)
print("PolWithOscilatingTime (programs), device-resident run     :", round(run(osc, 400, True), 1), "steps/s")
