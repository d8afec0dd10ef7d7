#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Go1 + PEA step (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one QuadrupedGymEnv.step() of every environment (10 physics substeps x 30 solver sweeps + task / reward /
observation epilogue) with actions already resident in HBM.  Environments shard over ranks with no data-path collective
(weak scaling: 8192 environments per GPU); the only collective is the max-over-ranks of the elapsed time."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "quadruped-springs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

ALGO_BYTES_PER_ENV_STEP = 1112  # SURVEY.md 8(d), d = 6, o = 28
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s


def workload(name):
    base = dict(enable_springs=True, enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD")
    if name == "jump_in_place_8192":   # the configuration BASELINE.json's metric is quoted on
        return 8192, dict(base, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", env_randomizer_mode="GROUND_RANDOMIZER",
                          time_step=0.001, action_repeat=10)
    if name == "config2_4096":         # BASELINE.json configs[1]: dt = 1/500 s, flat ground
        return 4096, dict(base, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", env_randomizer_mode="NONE",
                          time_step=0.002, action_repeat=5)
    if name == "config3_8192":         # configs[2]
        return 8192, dict(base, task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD",
                          env_randomizer_mode="SPRING_RANDOMIZER", time_step=0.001, action_repeat=10)
    if name == "config5_8192":         # configs[4]: backflip task, Hopf CPG action layer, masses + payload + springs + friction randomised
        return 8192, dict(base, task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", env_randomizer_mode="TEST_RANDOMIZER",
                          action_space_mode="CPG", time_step=0.001, action_repeat=10)
    if name == "config4_sharded":      # configs[3]: 8192 envs per GPU, JUMPING_FORWARD, actions broadcast + ONE all-gather of [n, o+2]
        return 8192, dict(base, task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC", env_randomizer_mode="GROUND_RANDOMIZER",
                          time_step=0.001, action_repeat=10)   # per rank and step, rank 0 fills an SB3-PPO-shaped rollout buffer
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(cfg_kwargs, budget_s=12.0):
    """The oracle (CPU restatement, float64, scalar C, OpenMP over environments) timed on this box's host cores on a bounded sample."""
    import numpy as np
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)          # read by libgomp when the oracle library is loaded
    os.environ["OMP_WAIT_POLICY"] = "passive"
    from oracle.qso import Oracle
    from qs_amd.config import build_config
    n = 8 * cores
    cfg, _ = build_config(n_envs=n, auto_reset=True, seed=1234, **cfg_kwargs)
    o = Oracle(cfg)
    o.reset()
    rng = np.random.default_rng(0)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(20):
            o.step(rng.uniform(-1, 1, size=(n, cfg.action_dim)).astype(np.float32))
        steps += 20
    dt = time.perf_counter() - t0
    return dict(value=n * steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{n} envs x {steps} env-steps of the same workload on {cores} host threads (OpenMP over environments), auto-reset incl. 2500-substep settles ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="jump_in_place_8192")
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--total-envs", type=int, default=0, help="strong scaling: this many environments split over the ranks (SURVEY 8e: 65536)")
    ap.add_argument("--reset-pool", type=int, default=4096, help="pre-settled reset states per GPU (0 = settle inside the step)")
    ap.add_argument("--no-pool-streaming", action="store_true",
                    help="do not re-settle the reset pool in the background while stepping (the pool is then filled once, before the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of the cpu_baseline leg
    ap.add_argument("--friction-model", default="pyramid", choices=["pyramid", "cone"],
                    help="pyramid with Bullet's skip rule (default) or PyBullet's implicit cone (enableConeFriction)")
    ap.add_argument("--solver-residual-threshold", type=float, default=0.0,
                    help="PyBullet solverResidualThreshold (its default is 1e-7); 0 = always int(300/action_repeat) sweeps")
    args = ap.parse_args()
    if args.cpu_baseline_only:   # runs in a child process that never touches the GPU or torch: its OpenMP runtime starts with the settings below
        _, kw = workload(args.workload)
        kw["solver_residual_threshold"] = args.solver_residual_threshold
        kw["friction_model"] = args.friction_model
        print(json.dumps(cpu_baseline(kw)))
        return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = args.workload == "config4_sharded"
    if world > 1 or sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the simulation step has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from qs_amd.vec_env import QuadrupedVecEnv
    n_default, kw = workload(args.workload)
    kw["solver_residual_threshold"] = args.solver_residual_threshold
    kw["friction_model"] = args.friction_model
    n = args.envs_per_gpu or n_default
    if args.total_envs:
        assert args.total_envs % (16 * world) == 0, "--total-envs must split into whole waves (16 environments) per rank"
        n = args.total_envs // world
    env = QuadrupedVecEnv(num_envs=n, device=local_rank, auto_reset=True, reset_pool=args.reset_pool, env_id_offset=n * rank,
                          seed=1234, **kw)   # Philox streams keyed by the global environment id: one job of n x world environments
    env.reset_tensor()
    d = env.action_dim
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = 64  # a ring of pre-generated U(-1,1) action batches, resident in HBM
    acts = torch.rand((n_act, n, d), generator=gen, device=dev) * 2 - 1
    local_step = step_fn = env.step_tensor
    if sharded:
        # the centralised-learner exchange of SURVEY.md 8e on top of the same local step: rank 0 owns the global action batch
        # and an SB3-PPO-shaped rollout buffer (n_steps = 128) that every gathered step is written into
        from qs_amd.sharded import ShardedVecEnv
        shard = ShardedVecEnv(env, learner_rank=0)
        n_glob, o_dim = n * world, env.obs_dim
        g_acts = (torch.rand((8, n_glob, d), generator=gen, device=dev) * 2 - 1) if rank == 0 else None
        if rank == 0:
            buf = dict(obs=torch.zeros((128, n_glob, o_dim), device=dev), act=torch.zeros((128, n_glob, d), device=dev),
                       rew=torch.zeros((128, n_glob), device=dev), start=torch.zeros((128, n_glob), device=dev))
        state = dict(t=0)

        def sharded_step(_unused):
            t = state["t"]
            a = g_acts[t % 8] if rank == 0 else None
            obs, rew, done, trunc = shard.step(a)
            if rank == 0:
                k = t % 128
                buf["obs"][k].copy_(obs); buf["act"][k].copy_(a); buf["rew"][k].copy_(rew); buf["start"][k].copy_(done)
            state["t"] = t + 1

        step_fn = sharded_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Auto-reset draws pre-settled states from a pool; with streaming on, as many entries as were consumed are re-settled by
    # extra workgroups of the step kernel (2500 substeps per state, new randomizer draws, action_repeat substeps per launch),
    # so the settle work of the resets consumed in the timed region is done in the timed region, next to the stepping.
    streaming = bool(args.reset_pool) and not args.no_pool_streaming
    if streaming:
        env.pool_streaming(True)
    # Untimed preparation: put the environments at evenly spread episode phases, as in the steady state of a training run
    # (they would otherwise all hit the 1000-step limit of gym_env.py:35 in the same step: one burst of N resets).
    groups = 16
    ids = torch.arange(n, device=dev)
    for gidx in range(groups):
        env.reset_tensor((ids % groups == gidx).to(torch.uint8))
        for i in range(1000 // groups):
            local_step(acts[i % n_act])
        torch.cuda.current_stream().synchronize()
    for i in range(args.warmup):
        step_fn(acts[i % n_act])
    kernel_ms = []
    barrier()
    stats0 = env.stats()
    limit0 = env.counter("limit_path_substeps")
    refills0 = env.pool_streaming(True) if streaming else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_fn(acts[i % n_act])
    barrier()
    elapsed = time.perf_counter() - t0
    stats1 = env.stats()
    limit1 = env.counter("limit_path_substeps")
    refills1 = env.pool_streaming(True) if streaming else 0
    env.enable_timing(True)
    # per-launch duration of the step kernel from HIP events on the kernel's own stream (separate short loop so that
    # the event synchronisation does not sit inside the timed region)
    for i in range(min(args.steps, 50)):
        local_step(acts[i % n_act])
        kernel_ms.append(env.last_step_kernel_ms())
    env.enable_timing(False)
    if streaming:
        env.pool_streaming(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    total_steps = n * world * args.steps
    if rank == 0:
        kavg = sum(kernel_ms) / len(kernel_ms) * 1e-3
        achieved = n * ALGO_BYTES_PER_ENV_STEP / kavg / 1e9
        # HBM bytes per launch from the committed PMC passes of this very configuration (rocprofv3 cannot run inside this
        # process); null when the run differs from the profiled one
        traffic = valu = pmc = None
        try:
            import glob
            # the latest committed PMC passes of this very configuration
            pmcs = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")), reverse=True)]
            pmc = next((p for p in pmcs if (p["workload"], p["envs_per_gpu"], p["reset_pool"], p["settle_lanes"], p.get("friction_model", "pyramid")) ==
                        (args.workload, n, args.reset_pool, streaming, args.friction_model)), None)
            if pmc is not None:
                traffic = (pmc["fetch_correction"] * pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024 / kavg / 1e9
                if "sq_insts_valu" in pmc:
                    # the roof that does bound this kernel: one wave64 fp32 VALU instruction per SIMD every 4 cycles (16 lanes per SIMD)
                    prop = torch.cuda.get_device_properties(dev)
                    peak = prop.multi_processor_count * 4 * getattr(prop, "clock_rate", 2.4e6) * 1e3 / 4 / 1e9
                    valu = {"achieved": pmc["sq_insts_valu"] / kavg / 1e9, "peak": peak, "unit": "G wave-instructions/s",
                            "frac": pmc["sq_insts_valu"] / kavg / 1e9 / peak,
                            "note": "SQ_INSTS_VALU per launch (same PMC passes) / step-kernel duration, against SIMDs x clock / 4; " +
                                    (f"at N = {n} only {(n // 16) / (prop.multi_processor_count * 4):.0%} of the SIMDs hold a stepping wave"
                                     if n // 16 < prop.multi_processor_count * 4 else f"at N = {n} every SIMD holds {(n // 16) / (prop.multi_processor_count * 4):.0f} stepping waves")}
        except (OSError, KeyError, ValueError):
            pmc = None
        out = {
            "metric": "env-steps/sec (whole node), Go1+PEA jump-in-place, N=8192 envs",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.total_envs else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "envs_per_gpu": n, "substeps_per_env_step": kw["action_repeat"],
                       "solver_sweeps": int(300 / kw["action_repeat"]), "solver_residual_threshold": args.solver_residual_threshold, "friction_model": args.friction_model, "dt": kw["time_step"], "actions": "U(-1,1), resident in HBM",
                       "auto_reset": True,
                       "reset": ((f"pool of {args.reset_pool} pre-settled states per GPU, " +
                                  ("consumed entries re-settled by extra workgroups of the step kernel (settle lanes)" if streaming else "filled once before the timed region"))
                                 if args.reset_pool else "2500-substep settle inside the step"),
                       "resets_in_timed_region": int(stats1["resets"] - stats0["resets"]),
                       "pool_states_settled_in_timed_region": int(refills1 - refills0),
                       "settle_substeps_in_timed_region": int(stats1["settle_substeps"] - stats0["settle_substeps"]),
                       "joint_limit_path_wave_substeps": int(limit1 - limit0),
                       "parallelism": (f"env-sharded x{world}, actions broadcast + one all-gather of [n, o+2] per step, rollout buffer on rank 0" if sharded
                                       else f"env-sharded x{world}, no data-path collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "valu_issue": valu, "kernel": (pmc or {}).get("kernel", "k_step"), "kernel_ms": kavg * 1e3,
                         "note": "achieved = N x 1112 algorithmic bytes per env-step (SURVEY 8d) / k_step duration; traffic = PMC bytes per launch (latest profiles/r*_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE) / the same duration, GB/s; the step is ~50 k dependent fp32 VALU instructions per wave per env-step (80 % VALU-active): latency/issue-bound, not HBM-bound"},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is a property of the box, reported with the single-GPU line only
            import subprocess
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
                                    "--solver-residual-threshold", str(args.solver_residual_threshold), "--friction-model", args.friction_model], capture_output=True, text=True, timeout=300)
            if child.returncode != 0:
                raise SystemExit("cpu_baseline child failed:\n" + child.stderr[-2000:])
            out["cpu_baseline"] = json.loads(child.stdout.strip().splitlines()[-1])
    env.close()
    if world > 1 or sharded:
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio, which (piped) would otherwise
        # be flushed at process exit, after anything printed here
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
