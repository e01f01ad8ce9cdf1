"""Counter passes of tools/placement_pmc.sh -> per-context means for the launches of one kernel family.

usage: placement_pmc_summary.py out.json pass1.csv [pass2.csv ...]
Every pass ran two contexts one after the other in one process; the dispatches of a kernel are split in
dispatch order into the first and the second half (context 0, context 1).  Reported per counter: mean per
launch in either context, their ratio, and the launch duration the profiler saw."""
import collections
import csv
import json
import sys

FAMILIES = {"predict": "PredictFn", "rhs_cl": "NewtonRhsFn<2", "rhs_cd": "NewtonRhsFn<3"}
out = sys.argv[1]
res = collections.defaultdict(dict)
for fn in sys.argv[2:]:
    rows = list(csv.DictReader(open(fn)))
    for fam, pat in FAMILIES.items():
        per = collections.defaultdict(dict)          # dispatch id -> counter -> value
        dur = {}
        for r in rows:
            if pat in r["Kernel_Name"]:
                d = int(r["Dispatch_Id"])
                per[d][r["Counter_Name"]] = float(r["Counter_Value"])
                if "Start_Timestamp" in r and r["Start_Timestamp"]:
                    dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        ids = sorted(per)
        if len(ids) < 4:
            continue
        half = len(ids) // 2
        groups = (ids[1:half], ids[half + 1:])           # first launch of a context left out
        for c in sorted({c for d in per.values() for c in d}):
            m = [sum(per[d].get(c, 0.0) for d in g) / len(g) for g in groups]
            res[fam][c] = {"context0": m[0], "context1": m[1], "ratio": (m[0] / m[1]) if m[1] else None}
        if dur:
            m = [sum(dur[d] for d in g) / len(g) for g in groups]
            res[fam].setdefault("launch_ms_under_profiler", []).append({"pass": fn.split("/")[-3] if fn.count("/") > 2 else fn,
                                                                        "context0": m[0], "context1": m[1]})
json.dump(res, open(out, "w"), indent=1)
for fam, d in res.items():
    print(fam)
    for c, v in d.items():
        if c == "launch_ms_under_profiler":
            for p in v:
                print(f"   ms under profiler [{p['pass']}]: {p['context0']:.3f} {p['context1']:.3f}")
        else:
            print(f"   {c:52s} {v['context0']:.6g} {v['context1']:.6g} ratio {v['ratio'] if v['ratio'] is None else round(v['ratio'], 4)}")
