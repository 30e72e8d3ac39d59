#!/usr/bin/env python3
"""Mean PMC values per (kernel, grid) over all dispatches in a rocprofv3 rocpd database.  usage: pmc_agg.py DB [substr]"""
import collections, re, sqlite3, sys
db = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
pe = next(t for t in tabs if t.startswith("rocpd_pmc_event")); pi = next(t for t in tabs if t.startswith("rocpd_info_pmc"))
names = {r[0]: r[1] for r in c.execute(f"select id, name from {pi}")}
ev = collections.defaultdict(dict)
for eid, pid, val in c.execute(f"select event_id, pmc_id, value from {pe}"):
    ev[eid][names[pid]] = ev[eid].get(names[pid], 0) + val
rows = c.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.event_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
agg = collections.OrderedDict()
for name, st, en, gx, gy, gz, wx, eid in rows:
    if sub not in name: continue
    nm = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    key = (nm, gx // wx, gy, gz)
    a = agg.setdefault(key, {"n": 0, "dur": 0.0, "c": collections.Counter()})
    a["n"] += 1; a["dur"] += (en - st) / 1e3
    for k, v in ev.get(eid, {}).items(): a["c"][k] += v
cols = sorted({k for a in agg.values() for k in a["c"]})
print("n dur_us grid | " + " | ".join(cols))
for (nm, gx, gy, gz), a in agg.items():
    n = a["n"]
    print(f"{n:3d} {a['dur']/n:8.1f} ({gx},{gy},{gz}) " + " ".join(f"{a['c'][k]/n/1e6:11.6f}M" for k in cols) + "  " + nm)
