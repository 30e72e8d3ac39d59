#!/usr/bin/env python3
"""Print the kernel timeline of one training step found in a rocprofv3 rocpd database (step boundary = the adam_kernel
dispatch).  Usage: python tools/step_timeline.py X_results.db [min_us] [step index, default -1 = the last complete one]"""
import re
import sqlite3
import sys

db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f", d.{qcol}" if qcol else ", 0"
rows = c.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, "
                 f"d.workgroup_size_x{sel} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if ("adam_kernel" in r[0] or "adam4_kernel" in r[0])]
if which < 0:
    which += len(ends)
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0 = step[0][1]
tot = 0.0
for name, st, en, gx, gy, gz, wx, q in step:
    nm = name.replace("(anonymous namespace)::", "").replace("void ", "")
    nm = re.sub(r"\(.*", "", nm)[:56]
    d = (en - st) / 1e3
    tot += d
    if d >= min_us:
        print(f"{(st - t0) / 1e3:9.1f} {d:8.1f}us  q{q} grid=({gx // wx},{gy},{gz})  {nm}")
print(f"# {len(step)} dispatches, kernel time {tot:.1f} us, span {(step[-1][2] - t0) / 1e3:.1f} us")
