"""The N = 256 study-mesh run of the bench (ExpSin, 400 device-resident predictor-corrector steps) on its own,
as a target for rocprofv3 --kernel-trace --stats."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402

print(json.dumps(bench.study_mesh_run()))
