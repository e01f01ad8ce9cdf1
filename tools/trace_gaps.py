"""Summary of a rocprofv3 kernel trace (csv): per kernel name the count, mean duration and the mean
idle time before it (start minus the previous kernel's end), plus totals.
    python tools/trace_gaps.py <dir-or-kernel_trace.csv> [skip_first_n]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[skip:]
agg = defaultdict(lambda: [0, 0, 0])
prev_end = None
busy = idle = 0
for s, e, name in rows:
    short = re.sub(r"<.*", "", name.replace("void ", "").replace("tedd::", ""))
    a = agg[short]
    a[0] += 1
    a[1] += e - s
    busy += e - s
    if prev_end is not None:
        a[2] += max(0, s - prev_end)
        idle += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
print(f"{len(rows)} kernels, busy {busy/1e3:.1f} us, idle {idle/1e3:.1f} us, span {(rows[-1][1]-rows[0][0])/1e3:.1f} us")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} n={n:6d}  mean {d/n/1e3:7.2f} us   gap before {g/n/1e3:6.2f} us")
