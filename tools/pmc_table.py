"""Per-kernel means of the counters in rocprofv3 --pmc counter_collection CSVs.

    python3 tools/pmc_table.py file.csv [file.csv ...] [--grep substring]
"""
import collections
import csv
import sys

files = [a for a in sys.argv[1:] if not a.startswith("--")]
grep = None
if "--grep" in sys.argv:
    grep = sys.argv[sys.argv.index("--grep") + 1]
    files = [f for f in files if f != grep]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "tedd" not in k or (grep and grep not in k):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k[:150])
    for c, x in sorted(v.items()):
        xs = [y for y in x if y > 0.05 * max(x)] if max(x) > 0 else x
        print(f"    {c:28s} mean {sum(xs) / len(xs):16.1f}  n {len(xs)}")
