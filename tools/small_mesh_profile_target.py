"""The device-resident step loop of the bench on a mesh of N x N cells (argument, default 32), as a target for
rocprofv3 --kernel-trace --stats: per-kernel durations of the small meshes of the convergence studies."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print(json.dumps(bench.study_mesh_run(N=N)))
