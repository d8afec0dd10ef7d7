#!/usr/bin/env python3
"""Static instruction counts per phase of the substep: compiles k_step with -DQS_COUNT_PHASES (scheduling barriers + assembly
comments at the phase boundaries) and counts the instructions between the marks in the main substep loop of the ISA."""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "base rotation, velocities", 1: "leg kinematics", 2: "link inertias", 3: "RNEA bias", 4: "CRBA (B, D, K)", 5: "Schur + Cholesky",
         6: "accelerations, v*", 7: "collision", 8: "contact rows + Delassus", 9: "PGS (loop body x sweeps not expanded)", 10: "delta v", 11: "integrate positions"}
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm",
                           "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-split-spill-mode=size", "-DQS_COUNT_PHASES", "-I" + os.path.join(REPO, "include"), "-save-temps", "-o", "t.so",
                           os.path.join(REPO, "quadruped-springs_amd", "csrc", "qs_hip.hip")], cwd=d, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "qs_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z6k_step")][0]
cur, counts, seen0, armed = None, collections.Counter(), 0, False
for l in lines[start:]:
    if l.startswith(".Lfunc_end"):
        break
    m = re.search(r"QS_PHASE_MARK (\d+)", l)
    if m:
        k = int(m.group(1))
        if k == 13:
            armed = True                          # E::step of the main path starts here (marks before it belong to the kernel entry)
        if k == 0 and armed:
            seen0 += 1
        cur = k if (armed and seen0 == 1 and k < 13) else None   # the first inlined copy of the substep after that is the main loop's
        continue
    t = l.strip()
    if cur is not None and re.match(r"(v_|s_|ds_|global_|scratch_|buffer_)", t) and not t.startswith("s_nop"):
        kind = "mfma" if t.startswith("v_mfma") else ("valu" if t.startswith("v_") else "other")
        counts[(cur, kind)] += 1
tot = 0
print(f"{'phase (instructions up to the next mark)':44s}  VALU  MFMA  other")
for k in range(0, 12):
    v, m, o = counts[(k, "valu")], counts[(k, "mfma")], counts[(k, "other")]
    tot += v + m + o
    print(f"{NAMES.get(k, str(k)):44s} {v:5d} {m:5d} {o:6d}")
print("total", tot)
