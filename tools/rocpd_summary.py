#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --stats -o X` -> X_results.db) into a
per-kernel table (calls, total, average, min, max, % of GPU kernel time).  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db [--skip-first N] > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = c.execute(q).fetchall()
    stats = {}
    for name, st, en in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        d = (en - st) / 1e3
        s = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        s[0] += 1
        s[1] += d
        s[2] = min(s[2], d)
        s[3] = max(s[3], d)
    total = sum(s[1] for s in stats.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0
    print(f"# {db}: {len(rows)} kernel dispatches, {total/1e3:.2f} ms of kernel time, span {span/1e3:.2f} ms\n")
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"| {name[:90]} | {s[0]} | {s[1]:.1f} | {s[1]/s[0]:.2f} | {s[2]:.2f} | {s[3]:.2f} | {100*s[1]/total:.1f} |")


if __name__ == "__main__":
    main()
