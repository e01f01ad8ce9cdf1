"""Per-kernel means of rocprofv3 --pmc passes (counter_collection CSVs) -> JSON for profiles/.

usage: pmc_summary.py out.json mesh_N solver sweeps_per_launch forcing_mode pass1.csv [pass2.csv ...]
(forcing_mode: the string bench.py prints in config.forcing)
The summary is stamped with the hash of the library the passes ran (and of its sources, and with $TEDD_GIT_REV
if set): bench.py quotes a summary's traffic only for that same library.
The second launch of the cs-Newton kernel in a step returns at once (no early exit): its near-zero
samples are dropped."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (library_stamp)

out, mesh, solver, hmax, forcing = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sys.argv[6:]:
    for r in csv.DictReader(open(fn)):
        if "tedd" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for k, v in agg.items():
    kernels[k] = {}
    for c, x in v.items():
        xs = [y for y in x if y > 0.05 * max(x)] if ("cs_newton" in k and max(x) > 0) else x
        kernels[k][c] = {"mean": sum(xs) / len(xs), "n": len(xs)}
json.dump({"config": {"mesh_N": mesh, "solver": solver, "sweeps_per_launch": hmax,
                      "forcing": forcing},
           "stamp": dict(bench.library_stamp(), git_rev=os.environ.get("TEDD_GIT_REV")),
           "units": {"FETCH_SIZE": "KB (x2 on gfx950 for coalesced streaming reads)", "WRITE_SIZE": "KB"},
           "kernels": kernels}, open(out, "w"), indent=1)
