#!/usr/bin/env python3
"""ISA of the headline step kernel alone (k_step<true, false>; --dense: k_step_dense<true, false>): compiles qs_hip.hip's device code with
build.py's flags and -DQS_ISA_ONLY (about half a minute instead of the library's two), writes the .s file and prints the loops
(tools/hot_loop_isa.py) and the scratch map.  usage: python tools/isa_headline.py [--dense] [-o out.s] [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize", "-ffinite-math-only",
         "-fno-signed-zeros", "-fno-trapping-math", "-ffp-contract=on", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]
args = sys.argv[1:]
dense = "--dense" in args
if dense:
    args.remove("--dense")
out = "/tmp/qs_headline%s.s" % ("_dense" if dense else "")
if "-o" in args:
    i = args.index("-o"); out = args[i + 1]; del args[i:i + 2]
cmd = ["hipcc"] + FLAGS + ["-DQS_ISA_ONLY"] + (["-DQS_ISA_DENSE"] if dense else []) + args + \
      ["-I" + os.path.join(REPO, "include"), "--cuda-device-only", "-S", "-o", out, os.path.join(REPO, "quadruped-springs_amd", "csrc", "qs_hip.hip")]
subprocess.check_call(cmd)
name = "_Z12k_step_denseILb1ELb0EE" if dense else "_Z6k_stepILb1ELb0EE"
lines = open(out).read().split("\n")
st = [i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].rstrip().endswith(":")][0]
en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
f = lines[st:en]
scr = [i for i, l in enumerate(f) if re.match(r"\s+scratch_", l)]
h = collections.Counter(i * 20 // len(f) for i in scr)
print(f"{name}: {len(f)} ISA lines, {len(scr)} scratch instructions ({sum(1 for i in scr if 'store' in f[i])} stores), first at line {scr[0] if scr else '-'}")
print("   per twentieth of the kernel: " + " ".join(f"{h.get(k, 0):3d}" for k in range(20)))
meta = [l.strip() for l in lines if re.search(r"\.(sgpr|vgpr|agpr)_count|\.private_segment_fixed_size|vgpr_spill_count", l)]
print("   " + "  ".join(meta[-5:]))   # (the step kernel is the file's last)
sys.stdout.flush()
subprocess.call([sys.executable, os.path.join(REPO, "tools", "hot_loop_isa.py"), out, name])
