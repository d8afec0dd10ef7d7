#!/usr/bin/env python3
"""The verdict of tools/gate.sh: parses the logs of one gate run, appends one JSON line to gpurun_out/validated_libraries.jsonl (to be committed as a
line of profiles/validated_libraries.jsonl) and exits 0 only if everything ran and nothing deviated.   usage: python tools/gate_record.py <dir> [full|quick]"""
import datetime
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "quadruped-springs_amd"))


def pytest_counts(path):
    try:      # pytest's summary line (not the log's last line: the one-rank RCCL test prints its banner behind it)
        tail = [l for l in open(path).read().splitlines() if re.search(r"\d+ (passed|failed|error)", l)][-1]
    except (OSError, IndexError):
        return dict(ran=False)
    g = lambda w: int(m.group(1)) if (m := re.search(rf"(\d+) {w}", tail)) else 0
    return dict(ran=True, passed=g("passed"), failed=g("failed") + g("error"), line=tail.strip("= "))


def fuzz_counts(path):
    try:
        m = re.search(r"(\d+) configurations ran, (\d+) deviated", open(path).read())
    except OSError:
        m = None
    return dict(ran=bool(m), configurations=int(m.group(1)) if m else 0, deviated=int(m.group(2)) if m else None)


def main(out, mode="full", jsonl_dir=None):
    import importlib.util
    from qs_amd import lib
    spec = importlib.util.spec_from_file_location("qs_build", os.path.join(REPO, "quadruped-springs_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rec = dict(date=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), mode=mode, library=os.path.relpath(lib.LIB_PATH, REPO),
               library_sha256=b.fingerprint(lib.LIB_PATH), source_sha256=lib.source_sha(), tree_sha256=b.source_fingerprint(), version=lib.load().qs_version().decode(),
               pytest_gpu=pytest_counts(os.path.join(out, "pytest_default.log")), pytest_gpu_dense_kernel=pytest_counts(os.path.join(out, "pytest_dense.log")),
               fuzz={k: fuzz_counts(os.path.join(out, f"fuzz_{k}.log")) for k in ("plain", "fallen", "lookahead", "fallen_dense")})
    try:
        soak = open(os.path.join(out, "soak.log")).read().strip().splitlines()
        rec["soak"] = dict(ok=soak[-1] == "ok", last=soak[-2] if len(soak) > 1 else "")
    except (OSError, IndexError):
        rec["soak"] = dict(ok=False, last="")
    tests_ok = all(rec[k]["ran"] and rec[k]["failed"] == 0 and rec[k]["passed"] > 100 for k in ("pytest_gpu", "pytest_gpu_dense_kernel"))
    fuzz_ok = all(v["ran"] and v["deviated"] == 0 and v["configurations"] > 50 for v in rec["fuzz"].values())
    rec["deviated"] = sum((v["deviated"] or 0) for v in rec["fuzz"].values()) + rec["pytest_gpu"].get("failed", 0) + rec["pytest_gpu_dense_kernel"].get("failed", 0)
    rec["accepted"] = bool(tests_ok and fuzz_ok and rec["soak"]["ok"] and rec["source_sha256"] == rec["tree_sha256"])
    line = json.dumps(rec)
    for path in (os.path.join(out, "verdict.json"), os.path.join(jsonl_dir or os.path.join(REPO, "gpurun_out"), "validated_libraries.jsonl")):
        with open(path, "a" if path.endswith(".jsonl") else "w") as f:
            f.write(line + "\n")
    print("GATE", "ACCEPTED" if rec["accepted"] else "REFUSED", line)
    return 0 if rec["accepted"] else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "full"))
