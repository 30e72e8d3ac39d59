#!/bin/bash
# r06 u: are the 50-70 us holes in the K = 1 configurations' replayed steps copies / memsets (not in a kernel trace)?
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06u; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/cfg2 -o trace -- python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r06u/cfg2/*_results.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
print([t for t in tabs if "copy" in t or "memory" in t])
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
mc = next((t for t in tabs if t.startswith("rocpd_memory_copy")), None)
rows = [(s, e, "K " + n[:60]) for n, s, e in c.execute(f"select s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id")]
if mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({mc})")]
    print(cols)
    for r in c.execute(f"select start, end, size from {mc}"):
        rows.append((r[0], r[1], f"COPY {r[2]} bytes"))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam4_kernel" in r[2]]
lo, hi = ends[13] + 1, ends[14] + 1
t0 = rows[lo][0]
prev = t0
for s, e, n in rows[lo:hi]:
    gap = (s - prev) / 1e3
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}us  gap {gap:6.1f}  {n}")
    prev = max(prev, e)
PY
rm -rf $OUT/cfg2
