#!/usr/bin/env python3
"""The loops of a kernel in a saved gfx950 assembly file (hipcc -save-temps), largest first: instruction, scratch, AGPR-copy, MFMA, LDS counts.
usage: python tools/hot_loop_isa.py file.s mangled_kernel_prefix [...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
DS = re.compile(r"\s+ds_")


def kern(name):
    st = [i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].rstrip().endswith(":")][0]
    en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[st:en]


def isinstr(l):
    l = l.strip()
    return bool(l) and not l.startswith((";", ".", "//")) and not l.endswith(":")


for name in sys.argv[2:]:
    f = kern(name)
    labels = {}
    for i, l in enumerate(f):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(f):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    loops.sort(key=lambda x: x[0] - x[1])
    print(name, "instructions:", sum(isinstr(l) for l in f))
    for a, b in [x for x in loops if x[1] - x[0] > 2500][:int(__import__("os").environ.get("QS_LOOPS", "16"))]:
        body = f[a:b + 1]
        cnt = lambda pred: sum(1 for l in body if pred(l))
        print(f"  lines {a}-{b}: {cnt(isinstr)} instr, scratch {cnt(lambda l: 'scratch_' in l)}, accvgpr {cnt(lambda l: 'v_accvgpr' in l)}, "
              f"mfma {cnt(lambda l: 'v_mfma' in l)}, waitcnt {cnt(lambda l: 's_waitcnt' in l)}, lds {cnt(lambda l: bool(DS.match(l)))}, "
              f"nop {cnt(lambda l: 's_nop' in l)}")
