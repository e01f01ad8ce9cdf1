"""The Python mirror of the reference API, exercised on the host twin of the device library
(CPU suite; the GPU suite repeats these on libtedd_hip.so with the reference's full sizes)."""
import numpy as np
import pytest

import api_cases as ac
from test_twin_cabi_parity import TWIN, build_twin
from tedd import _ffi


@pytest.fixture(scope="module", autouse=True)
def twin_library():
    build_twin()
    _ffi.use_library(TWIN)
    yield


def test_overridden_field_methods_are_refused():
    ac.check_overridden_field_methods_are_refused()


def test_classes_instantiation():
    ac.check_classes_instantiation()


@pytest.mark.parametrize("case,levels,rel", [("expsin", (2, 4, 8, 16, 32), 2e-9), ("pol", (2, 4, 8, 16), 5e-6),
                                             ("peaks_fast1e1", (2, 4, 8), 2e-9), ("nfs_h2_h3", (2, 4, 8), 2e-9)])
def test_published_notebook_errors(case, levels, rel):
    ac.check_published_errors(case, levels, rel)


@pytest.mark.parametrize("case,max_n", [("expsin", 32), ("pol", 16), ("peaks_fast1e1", 8), ("nfs_h2_h3", 8)])
def test_error_norms_of_reference_runs(case, max_n):
    ac.check_reference_run_errors(case, max_n, 2e-9)


@pytest.mark.parametrize("case", ["expsin", "pol", "peaks_fast1e1", "nfs_h2_h3"])
def test_fused_loop_equals_stepwise_loop(case):
    ac.check_fused_equals_stepwise(case, 8, 6)


def test_streamed_forcing_equals_closed_form():
    ac.check_streamed_equals_closed_form()


def test_field_correctness():
    ac.check_field_correctness([8, 16, 32])


def test_spatial_convergence():
    ac.check_spatial_convergence([8, 16, 32])


def test_pc_beats_forward_euler():
    ac.check_pc_beats_forward_euler()


@pytest.mark.parametrize("grid_n", [5, 15])
def test_forward_euler_consistency(grid_n):
    ac.check_forward_euler_order(grid_n, 0.05)


@pytest.mark.parametrize("case", ["expsin", "pol"])
def test_newton_residuals(case):
    ac.check_newton_residuals(case)


def test_error_behaviour():
    ac.check_error_behaviour()


# ---- HCsTriple (sharp Heaviside) through the mirrored classes
@pytest.mark.parametrize("case", ["pol", "expsin", "peaks_fast1e1"] + ac.SYMBOLIC_CASES)
def test_hcs_newton_residuals(case):
    if case not in ac.CASES:
        pytest.skip("needs sympy")
    ac.check_hcs_newton_residuals(case)


@pytest.mark.parametrize("kname", ["reg", "cst", "hcs"])
@pytest.mark.parametrize("cname", ac.SYMBOLIC_CASES + ["from_analytic"])
def test_symbolic_library_case_golden(cname, kname):
    ac.check_symbolic_library_case_golden(cname, kname)


@pytest.mark.parametrize("case", ac.SYMBOLIC_CASES)
def test_newton_residuals_symbolic_cases(case):
    if case not in ac.CASES:
        pytest.skip("needs sympy")
    ac.check_newton_residuals(case)


@pytest.mark.parametrize("cs_initial,cs_slope", [(1.0, -0.5), (0.1, -10.0), (0.1, -15.0), (-0.2, -0.5), (0.0, -5.0)])
def test_hcs_single_step_with_cs_ramp(cs_initial, cs_slope):
    ac.check_hcs_cs_ramp_single_step(cs_initial, cs_slope)


def test_hcs_corrector_value_error():
    ac.check_hcs_denominator_value_error()


@pytest.mark.parametrize("kind", ["reg", "cs3", "hcs"])
def test_integrator_pieces_called_on_their_own(kind):
    ac.check_integrator_pieces(kind)


# ---- cases given as SymPy expressions: run on the device as expression programs, or evaluated on the
# host and streamed
@pytest.mark.parametrize("device_program", [True, False])
@pytest.mark.parametrize("kind", ["reg", "cs3", "hcs"])
def test_symbolic_case_equals_builtin_case(kind, device_program):
    ac.check_symbolic_case_equals_builtin(kind, device_program)


@pytest.mark.parametrize("cs_initial,cs_slope", [(1.0, -0.5), (0.1, -15.0), (-0.2, -0.5)])
def test_symbolic_cs_ramp_single_step(cs_initial, cs_slope):
    ac.check_symbolic_cs_ramp(cs_initial, cs_slope)


@pytest.mark.parametrize("fused", [True, False], ids=["device_loop", "step_by_step"])
@pytest.mark.parametrize("label", list(ac.STUDIES["studies"]))
def test_legacy_study_runner_against_reference_report(label, fused):
    ac.check_run_convergence_studies(label, rel=1e-9, fused=fused)


@pytest.mark.parametrize("which", ["FT", "Fcl", "Fcd"])
def test_isolated_operator_truncation_order(which):
    ac.check_isolated_operator_truncation(which)


@pytest.mark.parametrize("which", ["T_cl", "T_cl_cd"])
def test_coupled_subsystem_spatial_convergence(which):
    ac.check_coupled_subsystem_convergence(which)


def test_isolated_correctors_temporal_order():
    ac.check_isolated_correctors_temporal_order()


def test_gradient_T_one_step_spatial_order():
    ac.check_gradient_T_one_step()


def test_forward_euler_T_cl_spatial_order():
    ac.check_forward_euler_T_cl()


@pytest.mark.parametrize("eta", [10.0, 50.0, 100.0])
def test_reghcs_scenarios_against_reference_numbers(eta):
    ac.check_reghcs_scenarios(eta)


def test_cases_and_forcing_terms_evaluate_on_foreign_meshes():
    ac.check_foreign_mesh_evaluation()


@pytest.mark.parametrize("kind", ["reg", "cs3", "hcs"])
def test_interaction_law_accessors(kind):
    ac.check_interaction_law_accessors(kind)


def test_hcs_cs_predictor_known_answers():
    ac.check_hcs_cs_predictor_table()


def test_hcs_cs_corrector_known_answers_and_errors():
    ac.check_hcs_cs_corrector_table()


def test_notebook_helpers(tmp_path):
    ac.check_notebook_helpers(tmp_path)


@pytest.mark.parametrize("case,max_n", [("expsin", 32), ("peaks_fast1e1", 16)])
def test_published_temporal_studies_coarse_levels(case, max_n):
    ac.check_published_temporal_study(case, 2e-9, max_n=max_n)


@pytest.mark.parametrize("case", ["nfs_h1_h2", "nfs_h2_h2"])
def test_published_errors_of_the_rougher_variants(case):
    ac.check_published_errors(case, (2, 4, 8, 16), 2e-9)


def test_meshes_down_to_one_cell():
    ac.check_tiny_meshes()
