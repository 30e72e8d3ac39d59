#!/usr/bin/env python3
"""Per-dispatch PMC table of the last training step.  usage: pmc_table.py DB [min_us]"""
import collections, re, sqlite3, sys
db = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
pe = next(t for t in tabs if t.startswith("rocpd_pmc_event")); pi = next(t for t in tabs if t.startswith("rocpd_info_pmc"))
names = {r[0]: r[1] for r in c.execute(f"select id, name from {pi}")}
ev = collections.defaultdict(dict)
for eid, pid, val in c.execute(f"select event_id, pmc_id, value from {pe}"):
    ev[eid][names[pid]] = ev[eid].get(names[pid], 0) + val
rows = c.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.event_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
step = rows[ends[-2] + 1:ends[-1] + 1]
cols = sorted({k for e in ev.values() for k in e})
print("dur_us grid " + " ".join(cols) + " kernel")
for name, st, en, gx, gy, gz, wx, eid in step:
    d = (en - st) / 1e3
    if d < min_us: continue
    e = ev.get(eid, {})
    nm = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
    print(f"{d:7.1f} ({gx//wx},{gy},{gz}) " + " ".join(f"{e.get(k,0)/1e6:.2f}M" for k in cols) + " " + nm)
