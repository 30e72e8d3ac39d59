#!/usr/bin/env python3
"""Last training step of a rocprofv3 kernel trace with queue ids (which HIP stream ran what).  usage: stream_timeline.py DB [min_us]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = c.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = step[0][1]
busy = {}
for name, st, en, gx, gy, gz, wx, q in step:
    busy[q] = busy.get(q, 0) + (en - st) / 1e3
    if (en - st) / 1e3 < min_us: continue
    nm = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:50]
    print(f"{(st-t0)/1e3:8.1f} {(en-st)/1e3:7.1f}us  q{q}  ({gx//wx},{gy},{gz}) {nm}")
print(f"# span {(step[-1][2]-t0)/1e3:.1f} us; busy per queue: " + ", ".join(f"q{q}: {v:.0f} us" for q, v in sorted(busy.items())))
