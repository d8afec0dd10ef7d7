#!/usr/bin/env python3
"""Copy what a GPU call of tools/r04_gpu_profiles.sh (r03_gpu_profiles.sh) left under gpurun_out/ into profiles/ (tracked), named per round and tag.
usage: python tools/collect_profiles.py gpurun_out/r03p gpurun_out/prof_r03a r03_a"""
import os
import shutil
import sys

run, prof, tag = sys.argv[1:4]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rnd = tag.split("_")[0]
for f in sorted(os.listdir(run)):
    if f.endswith(".json") and os.path.getsize(os.path.join(run, f)) > 10:
        shutil.copy(os.path.join(run, f), os.path.join(dst, f"{tag}_{f[:-5]}" + (".json" if f.startswith("numpy_path") else "_bench.json")))
for src, name in (("summary.md", "kernel_trace_pmc.md"), ("pmc.json", "pmc.json"), ("kernel_stats.csv", "kernel_stats.csv"), ("bench.json", "profiled_command_bench.json")):
    if os.path.exists(os.path.join(prof, src)):
        shutil.copy(os.path.join(prof, src), os.path.join(dst, f"{tag}_{name}"))
# the same PMC passes with the streaming refill off (tools/r02_gpu_profiles.sh), phase cycles and rare-path timings of the same call
static = prof.rstrip("/") + "_static"
for src, name in (("summary.md", "static_pool_kernel_trace_pmc.md"), ("pmc.json", "static_pool_pmc.json")):
    if os.path.exists(os.path.join(static, src)):
        shutil.copy(os.path.join(static, src), os.path.join(dst, f"{tag}_{name}"))
for src, name, title in (("phase_cycles.txt", "phase_cycles.md", "Cycles per phase of one physics substep (tools/phase_profile.py, -DQS_PROFILE_PHASES build of the same source)"),
                         ("rare_path.txt", "rare_path.md", "Step time when waves take the many-rows solver (tools/time_rare_path.py: NO_TASK, raw torques, N = 8192)"),
                         ("falling_policy.txt", "falling_policy.md", "Look-ahead resets under a policy that throws every robot down every ~38 steps (tools/falling_policy_rate.py, N = 8192)"),
                         ("gym_env_rate.txt", "gym_env_rate.md", "One environment through QuadrupedGymEnv.step (tools/gym_env_rate.py): the latency of one launch")):
    if os.path.exists(os.path.join(run, src)):
        body = [l for l in open(os.path.join(run, src)).read().splitlines() if "amdgpu.ids" not in l]
        with open(os.path.join(dst, f"{tag}_{name}"), "w") as f:
            f.write(f"# {title}\n\n```\n" + "\n".join(body) + "\n```\n")
if os.path.exists(os.path.join(run, "split_probe.jsonl")):
    shutil.copy(os.path.join(run, "split_probe.jsonl"), os.path.join(dst, f"{tag}_split_probe.jsonl"))
print(sorted(x for x in os.listdir(dst) if x.startswith(tag)))
