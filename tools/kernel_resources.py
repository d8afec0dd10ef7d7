#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in the built library's gfx950 code object (the note records of the ELF).
usage: python tools/kernel_resources.py [path/to/libqs_hip.so] [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(REPO, "quadruped-springs_amd", "qs_amd", "libqs_hip.so")
flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    assert blob[:24] == b"__CLANG_OFFLOAD_BUNDLE__"
    off = 24
    n, = struct.unpack_from("<Q", blob, off); off += 8
    co = None
    for _ in range(n):
        o, sz, il = struct.unpack_from("<QQQ", blob, off); off += 24
        name = blob[off:off + il].decode(); off += il
        if "gfx950" in name:
            co = os.path.join(d, "k.co"); open(co, "wb").write(blob[o:o + sz])
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    sizes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True).stdout
code = {m.group(2): int(m.group(1)) for m in re.finditer(r"\s+\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)", sizes)}
print(f"{'kernel':60s} {'vgpr':>5s} {'agpr':>5s} {'spill':>6s} {'scratch B':>9s} {'code KB':>8s}")
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk).group(1)
    name = g("name")
    if flt and flt not in name:
        continue
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    print(f"{short[:60]:60s} {g('vgpr_count'):>5s} {blk.split()[0]:>5s} {g('vgpr_spill_count'):>6s} {g('private_segment_fixed_size'):>9s} {code.get(name, 0) / 1024:8.1f}")
