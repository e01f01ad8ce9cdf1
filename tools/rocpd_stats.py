"""Per-kernel summary (calls, total/avg/min/max duration) of a rocprofv3 rocpd SQLite result."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                  "max(accum_vgpr_count), max(scratch_size), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print("name,calls,total_ms,avg_ms,min_ms,max_ms,pct,vgpr,agpr,scratch,lds")
for n, c, s, a, lo, hi, v, ag, sc, lds in rows:
    print(f"\"{n[:110]}\",{c},{s/1e6:.3f},{a/1e6:.4f},{lo/1e6:.4f},{hi/1e6:.4f},{100*s/total:.1f},{v},{ag},{sc},{lds}")
if len(sys.argv) > 2:  # list every dispatch of kernels whose name contains the pattern
    for n, st, d, gx in db.execute("select name, start, duration, grid_x from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)):
        print(f"{d/1e6:.3f} ms grid={gx} {n[:100]}")
