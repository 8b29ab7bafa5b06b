#!/usr/bin/env python3
"""Print the kernel timeline of the last profiled step from a rocprofv3 --kernel-trace sqlite database
(usage: python tools/trace_step.py <results.db> [marker-kernel-substring, default k_row_degree] [marker launches per step, default 1])."""
import sqlite3, sys
db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_row_degree"
rows = list(sqlite3.connect(db).execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s, e = idx[-1 - per], idx[-1]
t0, prev = rows[s][1], None
tot = {}
for r in rows[s:e]:
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    name = r[0].split("(")[0].replace("void ", "").replace("gm::", "")[:48]
    print(f"{(r[1]-t0)/1e3:9.1f}us dur {(r[2]-r[1])/1e3:8.1f}us gap {gap:7.1f}us blocks {r[3]//max(r[4],1):7d} {name}")
    prev = r[2]
    tot[name] = tot.get(name, 0.0) + (r[2] - r[1]) / 1e3
print("---- per-kernel totals (us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v:9.1f} {k}")
print(f"step span {(rows[e][1]-t0)/1e3:.1f} us, kernel time {sum(tot.values()):.1f} us")
