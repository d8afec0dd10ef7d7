#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) rocpd SQLite result (`*_results.db`, written by --kernel-trace --stats) into the small text
summary committed under profiles/.   usage: python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/...md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | AGPR | SGPR | LDS B | scratch B/lane | grid | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
tot = cur.execute("select sum(duration) from kernels").fetchone()[0]
q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
     "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc limit 8")
for n, c, s, a, mn, mx, v, ag, sg, lds, scr, g, wg in cur.execute(q):
    short = n.split("(")[0][:60]
    print(f"| {short} | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.1f} | {v} | {ag} | {sg} | {lds} | {scr} | {g} | {wg} |")
