#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories (kernel stats + one directory per --pmc pass) for one kernel.
usage: python tools/pmc_summary.py gpurun_out/prof3 k_step"""
import collections
import csv
import glob
import os
import sys

root, kernel = sys.argv[1], sys.argv[2]


def base(name):
    """'void k_step<false>(qs_config const*, ...)' -> 'k_step<false>'"""
    n = name.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


def wanted(name):   # "k_step" selects k_step and its template instances, not k_step_dense
    b = base(name)
    return b == kernel or b.startswith(kernel + "<")

for f in glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True):
    print("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        n = base(r["Name"])
        if n.startswith("k_") or float(r["Percentage"]) > 1:
            print(f"| {n[:50]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {r['Percentage']} |")
print()
print("| counter | per-launch avg | min | max | launches |\n|---|---|---|---|---|")
meta = None
for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if wanted(r["Kernel_Name"]):
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for c, v in d.items():
        print(f"| {c} | {sum(v) / len(v):.6g} | {min(v):.6g} | {max(v):.6g} | {len(v)} |")
if meta:
    print(f"\n{kernel}: grid {meta['Grid_Size']}, workgroup {meta['Workgroup_Size']}, LDS {meta['LDS_Block_Size']} B, scratch {meta['Scratch_Size']} B/lane, "
          f"VGPR {meta['VGPR_Count']}, AGPR {meta['Accum_VGPR_Count']}, SGPR {meta['SGPR_Count']}")
