#!/usr/bin/env python3
"""Where the step kernels touch scratch memory: compiles qs_hip.hip with build.py's flags and prints, per k_step / k_step_dense instance, the
number of scratch instructions in each twentieth of the kernel's ISA and the position of the first one.  The common-path build comes first
in k_step; it must hold none (round 2 found the action rows there: loops with a runtime trip count had turned them into memory arrays).
usage: python tools/scratch_map.py"""
import collections
import os
import re
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize", "-ffinite-math-only",
         "-fno-signed-zeros", "-fno-trapping-math", "-ffp-contract=on", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-split-spill-mode=size", "-mllvm", "-amdgpu-mfma-vgpr-form"]
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["hipcc"] + FLAGS + ["-I" + os.path.join(REPO, "include"), "-save-temps", "-o", "t.so",
                                               os.path.join(REPO, "quadruped-springs_amd", "csrc", "qs_hip.hip")], cwd=d, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "qs_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
for name in ("_Z6k_stepILb1ELb0EE", "_Z6k_stepILb1ELb1EE", "_Z12k_step_denseILb1ELb0EE"):   # <CONE, SOFT>
    st = [i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].rstrip().endswith(":")][0]
    en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    f = lines[st:en]
    scr = [i for i, l in enumerate(f) if re.match(r"\s+scratch_", l)]
    h = collections.Counter(i * 20 // len(f) for i in scr)
    print(f"{name}: {len(f)} ISA lines, {len(scr)} scratch instructions ({sum(1 for i in scr if 'store' in f[i])} stores), first at line {scr[0] if scr else '-'}")
    print("   per twentieth of the kernel: " + " ".join(f"{h.get(k, 0):3d}" for k in range(20)))
