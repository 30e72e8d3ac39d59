#!/usr/bin/env python3
"""Kernel time of ONE training step grouped by (kernel, grid): count, total, average, min, max — where the time of a step with a
thousand dispatches goes (cfg4 / cfg5).  Usage: python tools/step_groups.py X_results.db [step index, default -1] [top N]"""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = c.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                 f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if ("adam_kernel" in r[0] or "adam4_kernel" in r[0])]
if which < 0:
    which += len(ends)
step = rows[ends[which - 1] + 1:ends[which] + 1]
g = {}
for name, st, en, gx, gy, gz, wx in step:
    nm = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    k = (nm, gx // wx, gy, gz)
    d = (en - st) / 1e3
    a = g.setdefault(k, [0, 0.0, 1e30, 0.0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in g.values())
span = (step[-1][2] - step[0][1]) / 1e3
print(f"# {len(step)} dispatches, kernel time {tot:.0f} us, span {span:.0f} us\n| kernel | grid | n | total us | avg | min | max | % |\n|---|---|---:|---:|---:|---:|---:|---:|")
for (nm, gx, gy, gz), a in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| {nm} | ({gx},{gy},{gz}) | {a[0]} | {a[1]:.0f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / tot:.1f} |")
