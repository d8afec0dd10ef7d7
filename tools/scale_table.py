#!/usr/bin/env python3
"""Table of a tools/scale.sh run: per mode (weak / strong / sharded) and GPU count the whole-job env-steps/s, ms per step, the efficiency
against the 1-GPU line of the same mode (weak: value / (n x value_1); strong: the same with the total work fixed), the ranks RCCL saw and the
per-rank spread.  usage: python tools/scale_table.py gpurun_out/scale/scale.jsonl"""
import json
import sys


def parse(path):
    rows = []
    for raw in open(path):
        raw = raw.strip()
        if raw:
            rows.append(json.loads(raw))
    return rows


def table(rows):
    base = {r["mode"]: r["line"]["value"] for r in rows if r.get("line") and r["n_gpus"] == 1}
    out = []
    for r in rows:
        ln = r.get("line")
        if not ln:
            out.append(dict(mode=r["mode"], n_gpus=r["n_gpus"], value=None, error=r.get("error")))
            continue
        assert ln["n_gpus"] == r["n_gpus"], "a line reports another GPU count than it was run with"
        c = ln["config"]
        eff = ln["value"] / (r["n_gpus"] * base[r["mode"]]) if r["mode"] in base else None
        out.append(dict(mode=r["mode"], n_gpus=r["n_gpus"], value=ln["value"], ms_per_step=ln["ms_per_step"], scaling=ln["scaling"], efficiency=eff,
                        rccl_ranks=c.get("rccl_ranks"), rank_ms_min_max=c.get("rank_ms_per_step_min_max"), exchange_us=c.get("exchange_us"),
                        envs_per_gpu=c["envs_per_gpu"], stalls=c.get("stalls")))
    return out


if __name__ == "__main__":
    t = table(parse(sys.argv[1]))
    print("| mode | GPUs | envs/GPU | env-steps/s | ms/step | efficiency | RCCL ranks | rank ms/step min..max | exchange us | stalls |\n|---|---|---|---|---|---|---|---|---|---|")
    for r in t:
        if r["value"] is None:
            print(f"| {r['mode']} | {r['n_gpus']} | failed: {r['error']} |")
            continue
        mm = r["rank_ms_min_max"] or [None, None]
        print(f"| {r['mode']} | {r['n_gpus']} | {r['envs_per_gpu']} | {r['value'] / 1e6:.1f} M | {r['ms_per_step']:.4f} | "
              f"{'-' if r['efficiency'] is None else format(r['efficiency'], '.2f')} | {r['rccl_ranks']} | {mm[0]:.4f}..{mm[1]:.4f} | "
              f"{'-' if r['exchange_us'] is None else format(r['exchange_us'], '.1f')} | {r['stalls']} |")
