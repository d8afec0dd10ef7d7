#!/usr/bin/env python3
"""The projected Gauss-Seidel sweep loops of the hot build as compiled: finds, in the ISA of each k_step instance, the innermost loops whose body
holds the DPP-fused row updates (v_sub_f32_dpp ... quad_perm) and no MFMA, and prints their instruction counts (and the body of one).
usage: python tools/sweep_isa.py [--print cone|pyramid]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize", "-ffinite-math-only",
         "-fno-signed-zeros", "-fno-trapping-math", "-ffp-contract=on", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-split-spill-mode=size", "-mllvm", "-amdgpu-mfma-vgpr-form"]
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["hipcc"] + FLAGS + ["-I" + os.path.join(REPO, "include"), "-save-temps", "-o", "t.so",
                                               os.path.join(REPO, "quadruped-springs_amd", "csrc", "qs_hip.hip")], cwd=d, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "qs_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
isins = lambda l: re.match(r"\s+(v_|s_|ds_|global_|scratch_|buffer_)", l)
want = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--print" else None
for name, label in (("_Z6k_stepILb0ELb0EE", "pyramid"), ("_Z6k_stepILb1ELb0EE", "cone")):
    st = [i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].rstrip().endswith(":")][0]
    en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    f = lines[st:en]
    lab = {m.group(1): i for i, l in enumerate(f) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(f):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            body = [x.strip() for x in f[lab[m.group(1)]:i + 1] if isins(x)]
            dpp = sum(1 for x in body if "quad_perm" in x and x.startswith("v_sub_f32"))
            if dpp >= 12 and not any(x.startswith("v_mfma") for x in body) and len(body) < 400:
                loops.append((lab[m.group(1)], len(body), dpp, sum(1 for x in body if x.startswith("scratch_")), body))
    # the hot build's loops come first in the function (the full build's copy follows)
    loops.sort()
    print(f"k_step<{'true' if label == 'cone' else 'false'}, false>  ({label}):")
    for k, (pos, n, dpp, scr, body) in enumerate(loops[:4]):
        c = collections.Counter(x.split()[0] for x in body)
        kind = ""
        print(f"  sweep loop {k}: {n} instructions, {dpp} DPP row broadcasts, {scr} scratch")
    if want == label and loops:
        print("\n```")
        print("\n".join("    " + x for x in loops[int(os.environ.get("QS_SWEEP_LOOP", "0"))][4]))
        print("```")
