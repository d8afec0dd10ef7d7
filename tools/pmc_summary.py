#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories (kernel stats + one directory per --pmc pass) for one kernel.
usage: python tools/pmc_summary.py gpurun_out/prof_x k_step [bench.json pmc.json] [--last N]
--last N: counters averaged over the LAST N launches of the kernel only -- with N = the profiled command's --steps that is its timed
region (the launches before it are preparation: staggered resets, pre-roll, warmup, whose settle load is not the steady state's).
With bench.json pmc.json the per-launch averages are also written as the pmc.json that bench.py reads (roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

argv = list(sys.argv)
last = 0
if "--last" in argv:
    i = argv.index("--last"); last = int(argv[i + 1]); del argv[i:i + 2]
root, kernel = argv[1], argv[2]
bench_json, out_json = (argv[3], argv[4]) if len(argv) > 4 else (None, None)


def base(name):
    """'void k_step<false, true>(qs_config const*, ...)' -> 'k_step<false, true>'"""
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
# per-dispatch durations of the kernel trace: the same average restricted to the last N launches (the timed region of the profiled command)
trace_avg_us = None
# (the trace of the pass WITHOUT counters -- profile_round.sh's trace/ directory -- if it is there: under --pmc the launches run ~6 % slower)
traces = sorted(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True), key=lambda f: (os.sep + "trace" + os.sep not in f, f))
for f in traces:
    rows = [r for r in csv.DictReader(open(f)) if wanted(r.get("Kernel_Name", ""))]
    if rows and "Start_Timestamp" in rows[0]:
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        sel = d[-last:] if last else d
        trace_avg_us = sum(sel) / len(sel)
        srt = sorted(sel)
        print(f"\n{kernel}: {len(d)} launches in the kernel trace; " + (f"the last {len(sel)} (the timed region)" if last else "all of them") +
              f": average {trace_avg_us:.2f} us, min {srt[0]:.2f}, median {srt[len(srt) // 2]:.2f}, 99th percentile {srt[int(len(srt) * 0.99)]:.2f}, max {srt[-1]:.2f} us")
    break
print()
print("| counter | per-launch avg | min | max | launches |\n|---|---|---|---|---|")
meta, avg, names = None, {}, collections.Counter()
for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if wanted(r["Kernel_Name"]):
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names[base(r["Kernel_Name"])] += 1
            meta = r
    for c, v in d.items():
        if last:
            v = v[-last:]          # rows are in dispatch order
        avg[c] = (sum(v) / len(v), len(v))
        print(f"| {c} | {sum(v) / len(v):.6g} | {min(v):.6g} | {max(v):.6g} | {len(v)} |")
if meta:
    print(f"\n{kernel}: grid {meta['Grid_Size']}, workgroup {meta['Workgroup_Size']}, LDS {meta['LDS_Block_Size']} B, scratch {meta['Scratch_Size']} B/lane, "
          f"VGPR {meta['VGPR_Count']}, AGPR {meta['Accum_VGPR_Count']}, SGPR {meta['SGPR_Count']}")
if out_json and avg:
    import importlib.util
    _spec = importlib.util.spec_from_file_location("qs_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quadruped-springs_amd", "build.py"))
    build = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(build)
    b = json.load(open(bench_json))
    c = b["config"]
    j = {"kernel": names.most_common(1)[0][0], "workload": c["workload"], "envs_per_gpu": c["envs_per_gpu"],
         "reset_lookahead": c["reset_lookahead"],
         "friction_model": c["friction_model"], "solver_residual_threshold": c["solver_residual_threshold"],
         "body_contacts": "true" if c.get("body_contacts") is True else str(c.get("body_contacts", "auto")).lower(),
         "source_sha256": build.source_fingerprint(), "library_sha256": build.fingerprint(os.environ.get("QS_LIB_PATH") or None),
         "fetch_size_kb": avg["FETCH_SIZE"][0], "write_size_kb": avg["WRITE_SIZE"][0], "fetch_correction": 2.0,
         "sq_insts_valu": avg.get("SQ_INSTS_VALU", (None, 0))[0], "sq_waves": avg.get("SQ_WAVES", (None, 0))[0],
         "launches": avg["FETCH_SIZE"][1], "bench_value": b["value"], "bench_kernel_ms": b["roofline"]["kernel_ms"], "rocprof_kernel_us_timed_region": trace_avg_us,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_*, separate passes, per-launch averages over " + (f"the last {last} launches (the timed region)" if last else "ALL launches") + " of the step kernel in "
                 "the profiled command (same table as the *_kernel_trace_pmc.md next to this file); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(16-B-per-lane streaming reads are tallied at half their bytes on gfx950)"}
    json.dump(j, open(out_json, "w"), indent=1)
