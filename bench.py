#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Go1 + PEA step (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python bench.py --gpus 8 --steps 1000 --warmup 50                       # starts the 8 ranks itself (one fresh child per GPU)
    python bench.py --gpus 8 --total-envs 65536 --steps 1000 --warmup 50    # strong scaling: SURVEY.md 8e's N = 65536 split over the ranks
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


def usable_cores():
    """Host threads this process can actually run at once: the affinity mask, cut down to the CPU-time quota of its cgroup when there is
    one (cgroup v2 cpu.max, v1 cpu.cfs_quota_us).  Round 2's box showed 256 CPUs in the mask and gave ~8 CPUs' worth of time: 256 busy
    threads then run at 3 % each."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    visible, quota = n, None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(math.floor(quota + 1e-9))))
    return n, visible, quota


def cpu_baseline(cfg_kwargs, budget_s=14.0):
    """The oracle (CPU restatement, float64, scalar C) timed on this box's host cores on a bounded sample of the same workload: 64
    environments per host thread, every thread stepping its own share without a barrier between steps (qso_rollout: independent workers
    are the CPU's best case -- a reset's in-place 2500-substep settle then delays only its own thread), once on ONE thread and once on
    all of them.  Also tries the reference's own PyBullet path."""
    import numpy as np
    cores, visible, quota = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)          # read by libgomp when the oracle library is loaded
    os.environ["OMP_WAIT_POLICY"] = "passive"
    from oracle.qso import Oracle
    from qs_amd.config import build_config
    per_thread, rng = 64, np.random.default_rng(0)

    def leg(threads, budget):
        n = per_thread * threads
        cfg, _ = build_config(n_envs=n, auto_reset=True, seed=1234, **cfg_kwargs)
        o = Oracle(cfg)
        o.set_threads(threads)
        o.reset()
        ring = rng.uniform(-1, 1, size=(64, n, cfg.action_dim)).astype(np.float32)
        t0 = time.perf_counter(); o.rollout(ring, 4); probe = (time.perf_counter() - t0) / 4      # seconds per step of all n environments
        steps = max(8, int(budget / max(probe, 1e-6)))
        t0 = time.perf_counter(); resets = o.rollout(ring, steps); dt = time.perf_counter() - t0
        o.close()
        return dict(threads=threads, envs=n, env_steps=n * steps, seconds=dt, rate=n * steps / dt, resets=resets,
                    settle_share=resets * cfg.settle_steps / (resets * cfg.settle_steps + n * steps * cfg.action_repeat))

    one = leg(1, 0.3 * budget_s)
    allt = leg(cores, 0.7 * budget_s) if cores > 1 else one
    try:
        import pybullet  # noqa: F401
        ref = "pybullet importable, but the reference's QuadrupedGymEnv also needs gym and the reference tree, which the GPU box does not hold: not run"
    except Exception as e:  # noqa: BLE001
        ref = f"unavailable ({type(e).__name__}: pybullet==3.2.5 of the reference's setup.py:10 is not installed on this box)"
    return dict(value=allt["rate"], unit="env-steps/s", cores=cores, kind="port", cpus_in_affinity_mask=visible, cgroup_cpu_quota=quota,
                single_thread=one["rate"], per_thread=allt["rate"] / cores,
                parallel_efficiency=allt["rate"] / (cores * one["rate"]),
                resets_in_sample=allt["resets"], settle_share_of_substeps=allt["settle_share"], single_thread_settle_share=one["settle_share"],
                reference_pybullet=ref,
                sample=(f"{allt['envs']} envs ({per_thread} per thread) x {allt['env_steps'] // allt['envs']} env-steps of the same workload on {cores} host "
                        f"threads, each stepping its own environments with no barrier between steps, auto-reset incl. its in-place 2500-substep settles "
                        f"({allt['resets']} resets = {100 * allt['settle_share']:.0f} % of the substeps; {allt['seconds']:.1f} s); single thread: "
                        f"{one['envs']} envs x {one['env_steps'] // one['envs']} env-steps, {one['resets']} resets ({one['seconds']:.1f} s)"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preroll", type=int, default=2048,
                    help="untimed steps before the warmup steps that bring episode ages, reset rate and the settle lanes "
                         "to their steady state (part of the preparation; 0 to skip)")
    ap.add_argument("--workload", default="jump_in_place_8192")
    ap.add_argument("--action-ring", type=int, default=64,
                    help="number of pre-generated U(-1,1) action batches resident in HBM; step i uses batch i mod ring.  (With 64 an environment whose "
                         "64-step pattern throws it down does so at the same ring phase episode after episode: the fall rate is periodic in the step "
                         "count, and a 20-step region sees a fixed stretch of that period, DESIGN.md 6)")
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--total-envs", type=int, default=0, help="strong scaling: this many environments split over the ranks (SURVEY 8e: 65536)")
    ap.add_argument("--reset-lookahead", type=int, default=16,
                    help="K: reset states kept ready per environment (its own next K episodes, settled ahead of time by extra workgroups of the step "
                         "kernel; results are bitwise those of 0 = every reset settles inside the step).  The benchmark's U(-1,1) actions come from a "
                         "ring of 64 batches, so some environments fall every ~65 steps, episode after episode: five settles of such an environment are "
                         "in flight at any time, and K must cover them (K = 16: 151 MB at N = 8192)")
    ap.add_argument("--no-settle-lanes", action="store_true",
                    help="experiments: leave the look-ahead states un-replenished (resets settle in place once an environment has used its K states)")
    ap.add_argument("--no-info-line", action="store_true", help="skip the second timed loop with info_fields=True (value_info_fields_true)")
    ap.add_argument("--no-body-contacts-line", action="store_true",
                    help="skip the timed loop of the OTHER setting of body_contacts (value_body_contacts_true / value_body_contacts_auto: the "
                         "reference's all-links contact response on / left to the task, DESIGN.md 4a)")
    ap.add_argument("--allow-short-preroll", action="store_true",
                    help="experiments: accept a --preroll below two episode lengths for the headline workload (the reset rate is then not the steady state's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of the cpu_baseline leg
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the --gpus N children exactly as a real run does, but each only prints its rank environment (one JSON line) and exits; touches no GPU")
    ap.add_argument("--friction-model", default="cone", choices=["pyramid", "cone"],
                    help="PyBullet's implicit cone (its default, and this build's) or the friction pyramid with Bullet's skip rule")
    ap.add_argument("--solver-residual-threshold", type=float, default=1e-7,
                    help="PyBullet solverResidualThreshold (its default 1e-7 is this build's default); 0 = always int(300/action_repeat) sweeps")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the benchmark.  gloo is the TEST MODE of this file's multi-rank code (tests/test_bench_launch.py): CPU "
                         "ranks run the same launcher, barrier, MAX / MIN reduction and JSON path around a stand-in environment named by "
                         "--standin; what it prints is not a measurement")
    ap.add_argument("--standin", default="", help=argparse.SUPPRESS)   # module:factory of the gloo test mode's environment (lives under tests/)
    ap.add_argument("--spread-steps", type=int, default=1000, help=argparse.SUPPRESS)   # steps of the episode-phase spreading in the preparation
    ap.add_argument("--env-kw", nargs="*", action="extend", default=[], metavar="KEY=VALUE",
                    help="extra QuadrupedVecEnv keywords for experiments (python literals, anything else is taken as a string), "
                         "e.g. self_collision=False payload=soft env_randomizer_mode=MASS_RANDOMIZER")
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime in the launcher (its children must be the first to do so): the
    KFD topology lists every node, CPUs with simd_count 0; HIP_/ROCR_/CUDA_VISIBLE_DEVICES restrict the list.  Falls back to
    torch.cuda.device_count() in a short-lived child process."""
    n = None
    try:
        root = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        n = None
    if not n:
        import subprocess
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
        try:
            n = int(out.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh children, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set), BEFORE anything in this process touches a GPU; rank 0's child prints the JSON line.  Returns the exit code."""
    import subprocess
    port = free_port()
    envs = []
    for r in range(args.gpus):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between the ranks
        envs.append(e)
    visible = args.gpus if args.backend == "gloo" else visible_gpus()      # (the gloo test mode runs its ranks on the CPU)
    if visible < args.gpus and not args.dry_launch:
        print(f"bench.py: --gpus {args.gpus} but {visible} GPU(s) visible on this box; refusing to report a {args.gpus}-GPU line from fewer devices",
              file=sys.stderr)
        return 2
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e) for e in envs]
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):      # one rank failed: stop exactly the children started here
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.wait()
            break
        time.sleep(0.2)
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return 1
    return 0


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("qs_build", os.path.join(REPO, "quadruped-springs_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def running_fingerprint():
    """(source fingerprint, reason it cannot be trusted or None): the tree this process's libqs_hip.so was built from.  Another
    library through QS_LIB_PATH, or a library older than its sources, belongs to no fingerprint."""
    b = _build_module()
    try:      # since round 6 the library says itself which sources it was compiled from (qs_version(), build.py's -DQS_SOURCE_SHA)
        from qs_amd import lib as _l
        sha = _l.source_sha()
        if sha:
            return sha, None
    except Exception:  # noqa: BLE001  (no device library in this process: the stand-in test mode)
        pass
    if os.environ.get("QS_LIB_PATH"):
        return None, "QS_LIB_PATH names another library than the tree's"
    if b.needs_build():
        return None, "libqs_hip.so is older than its sources"
    return b.source_fingerprint(), None


def pmc_for_run(profiles_dir, key, fingerprint, why_no_fingerprint=None):
    """The committed counter passes that belong to THIS run: the newest profiles/r*_pmc.json whose workload keys equal `key` AND whose
    `source_sha256` equals the fingerprint of the tree the running library was built from.  Returns (pmc dict or None, file or None,
    reason or None).  Counters of another binary are not reported (round 4 copied them unchecked)."""
    import glob
    stale = None
    for f in sorted(glob.glob(os.path.join(profiles_dir, "r*_pmc.json")), reverse=True):
        try:
            p = json.load(open(f))
            k = (p["workload"], p["envs_per_gpu"], p.get("reset_lookahead"), p.get("friction_model", "pyramid"), float(p.get("solver_residual_threshold", 0.0)),
                 p.get("body_contacts", "auto"))
        except (OSError, KeyError, ValueError):
            continue
        if k != key:
            continue
        if fingerprint is not None and p.get("source_sha256") == fingerprint:
            return p, f, None
        if stale is None:
            stale = (f, "carries no source fingerprint" if "source_sha256" not in p else "was taken on another source tree")
    if stale:
        return None, None, f"{os.path.basename(stale[0])} {stale[1]}" + (f" ({why_no_fingerprint})" if fingerprint is None and why_no_fingerprint else "") + ": counters not reported"
    return None, None, "no committed counter pass of this configuration"


def default_body_contacts(gloo=False):
    """What QuadrupedVecEnv does when the caller says nothing (qs_amd.config.build_config's default)."""
    if gloo:
        return "auto"
    import inspect
    from qs_amd.config import build_config
    return inspect.signature(build_config).parameters["body_contacts"].default


EPISODE_STEPS = 1001   # gym_env.py:35, 245: an episode ends once sim_time exceeds 10 s = after 1001 env steps of 10 ms


def metric_name(workload_name, kw, n, world, total_envs):
    if workload_name == "jump_in_place_8192" and n == 8192 and not total_envs:
        return "env-steps/sec (whole node), Go1+PEA jump-in-place, N=8192 envs"     # BASELINE.json's metric, verbatim
    task = kw["task_env"].lower().replace("_", "-")
    return f"env-steps/sec (whole node), Go1+PEA {task} ({workload_name}), N={n} envs per GPU x {world} GPU(s)"


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.cpu_baseline_only:   # runs in a child process that never touches the GPU or torch: its OpenMP runtime starts with the settings below
        _, kw = workload(args.workload)
        kw["solver_residual_threshold"] = args.solver_residual_threshold
        kw["friction_model"] = args.friction_model
        print(json.dumps(cpu_baseline(kw)))
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    headline_cmd = args.workload == "jump_in_place_8192" and not args.env_kw and args.reset_lookahead == 16 and not args.no_settle_lanes
    if headline_cmd and args.preroll < 2 * EPISODE_STEPS and not args.allow_short_preroll and not args.standin:
        raise SystemExit(f"bench.py: --preroll {args.preroll} is below two episode lengths ({2 * EPISODE_STEPS} steps): the headline is quoted at the steady-state "
                         "reset rate, which the pre-roll builds up (--allow-short-preroll for experiments)")
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if under_launcher and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; they must agree")
    if not under_launcher and (args.gpus > 1 or args.dry_launch):
        sys.exit(launch_ranks(args, argv))
    if args.dry_launch:   # a child of `--dry-launch`: report the rank environment it was started with and stop before any GPU call
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"pid": os.getpid(), "ppid": os.getppid()}), flush=True)
        return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus
    # Only rank 0 reports, and only its JSON line: whatever else any rank -- or the RCCL / gloo it loads, which print banners through C stdio --
    # writes to stdout goes to stderr.  Rank 0 keeps a duplicate of the real stdout for the line.
    sys.stdout.flush()
    real_stdout = os.dup(1) if rank == 0 else None
    os.dup2(2, 1)
    sharded = args.workload == "config4_sharded"
    gloo = args.backend == "gloo"
    if gloo and not args.standin:
        raise SystemExit("bench.py --backend gloo is the test mode of the multi-rank code and needs --standin module:factory (see tests/test_bench_launch.py); "
                         "the simulation step has no CPU path")
    if not gloo and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the simulation step has no CPU path")
    if world > 1 or sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
    if gloo:
        dev = torch.device("cpu")
        import importlib
        mod, _, fac = args.standin.partition(":")
        QuadrupedVecEnv = getattr(importlib.import_module(mod), fac)
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        from qs_amd.vec_env import QuadrupedVecEnv
    n_default, kw = workload(args.workload)
    kw["solver_residual_threshold"] = args.solver_residual_threshold
    kw["friction_model"] = args.friction_model
    import ast
    def literal(v):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    extra_kw = {k: literal(v) for k, v in (item.split("=", 1) for item in args.env_kw)}
    # a learner that consumes observations, rewards and done flags (SB3 PPO) never reads the records' info block (torques, foot forces,
    # the task's pose cache): the steps do not write it (info_fields = False; getters for it would fail loudly).  The rate of the default
    # handle (info_fields = True) is measured by a second timed loop and reported as value_info_fields_true.
    kw.setdefault("info_fields", False)
    kw.update(extra_kw)
    n = args.envs_per_gpu or n_default
    if args.total_envs:
        assert args.total_envs % (16 * world) == 0, "--total-envs must split into whole waves (16 environments) per rank"
        n = args.total_envs // world
    d_gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = max(1, args.action_ring)  # a ring of pre-generated U(-1,1) action batches, resident in HBM

    spin = os.environ.get("QS_BENCH_SPIN", "1") != "0"

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        if not gloo:
            if spin:   # poll an event behind the work first: the synchronize below then returns at once instead of waking up late
                       # (measured over four 20-step regions each way on one box: 104.1 against 102.9 M on average, inside the +-2.5 % of such regions)
                ev = torch.cuda.Event()
                ev.record()
                while not ev.query():
                    pass
            torch.cuda.synchronize()

    def run(env_kw, with_exchange):
        """Preparation + the timed region for one handle; returns the measurements of this rank."""
        env = QuadrupedVecEnv(num_envs=n, device=local_rank, auto_reset=True, reset_lookahead=args.reset_lookahead, env_id_offset=n * rank,
                              seed=1234, **env_kw)   # Philox streams keyed by the global environment id: one job of n x world environments
        if args.no_settle_lanes and args.reset_lookahead:
            env.settle_lanes(False)
        env.reset_tensor()
        d = env.action_dim
        acts = torch.rand((n_act, n, d), generator=d_gen, device=dev) * 2 - 1
        local_step = step_fn = env.step_tensor
        if with_exchange:
            # the centralised-learner exchange of SURVEY.md 8e on top of the same local step.  Rank 0 owns an SB3-PPO-shaped rollout
            # buffer (n_steps = 128): its action rows ARE the broadcast source and its result rows [N, o + 2] (observation | reward |
            # done + 2 truncated) ARE the all-gather destination, so a step adds no copy kernel: broadcast (world > 1 only), the step
            # kernel writing this rank's rows in place, one in-place all-gather (world > 1 only).
            from qs_amd.sharded import ShardedVecEnv
            shard = ShardedVecEnv(env, learner_rank=0)
            n_glob, o_dim = n * world, env.obs_dim
            n_roll = 128
            roll_act = torch.rand((n_roll, n_glob, d), generator=d_gen, device=dev) * 2 - 1   # stands in for the policy's outputs
            roll_res = torch.zeros((n_roll if rank == 0 else 1, n_glob, o_dim + 2), device=dev)
            state = dict(t=0)

            def sharded_step(_unused):
                k = state["t"] % n_roll
                shard.step(roll_act[k] if rank == 0 else None, out=roll_res[k if rank == 0 else 0], unpack=False)   # the rollout buffer keeps the fused rows
                state["t"] += 1

            step_fn = sharded_step
        # Auto-reset: every environment takes its own next reset state (randomizer draws of (seed, environment, episode), spawn, 2500
        # settle substeps -- computed ahead of time by extra workgroups of the step kernel, action_repeat substeps per launch), so the
        # settle work of the resets consumed in the timed region is done in the timed region, next to the stepping
        # (config.settle_work_ratio says how much of it this particular run did).
        # Untimed preparation: put the environments at evenly spread episode phases, as in the steady state of a training run
        # (they would otherwise all hit the 1000-step limit of gym_env.py:35 in the same step: one burst of N resets).
        groups = 125    # (16 groups left a burst of several hundred time-limit resets every 62 steps: a 20-step timed region either held one or did not)
        ids = torch.arange(n, device=dev)
        # (the robots stand still while their episode ages are spread: under the random actions some fall every ~20 steps from the very first
        # launch on and use up their 16 look-ahead states before the settle lanes -- whose first states arrive 300 launches after
        # qs_create -- deliver: ~155 in-step settles of 14.5 ms each in rounds 3-4's preparation, 2 s of the run and most of rocprofv3's
        # all-launches average.  The pre-roll below is where the random actions' steady state builds up.)
        still = torch.zeros((n, d), device=dev)
        for gidx in range(groups):
            env.reset_tensor((ids % groups == gidx).to(torch.uint8))
            for i in range(args.spread_steps // groups):
                local_step(still)
            if not gloo:
                torch.cuda.current_stream().synchronize()
        # ... and let the reset rate and with it the settle lanes reach their steady state (episodes of random actions last ~600 steps, a
        # settle takes 250 launches, a cohort of lanes starts every 50)
        for i in range(args.preroll):
            local_step(acts[i % n_act])
        for i in range(args.warmup):
            step_fn(acts[i % n_act])
        # the counters in front of the timed region: a stream-ordered snapshot into device memory behind the last warmup step, read after the
        # region (the synchronising reads that stood here left the device idle long enough for the region's first launch to start ~100 us
        # late, 5 us per step of a 20-step region)
        snap0 = env.counters_snapshot()
        barrier()
        # the step kernel's launches of the timed region between two HIP events on the kernel's own stream (qs_enable_timing: one event in
        # front of the first launch, one behind the last): per-launch duration for the roofline, measured over the timed region itself
        env.enable_timing(True)
        debug = bool(os.environ.get("QS_BENCH_DEBUG"))
        t0 = time.perf_counter()
        t_first = t0
        for i in range(args.steps):
            step_fn(acts[i % n_act])
            if debug and i == 0:
                t_first = time.perf_counter()
        t_loop = time.perf_counter()
        env.enable_timing(2)                        # the closing event, in stream order behind the last launch; waited for after the region
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = env.last_step_kernel_ms()
        if debug:
            print(f"[debug] first call {1e6 * (t_first - t0):.0f} us, launching {1e6 * (t_loop - t0):.0f} us, until the barrier returned {1e6 * elapsed:.0f} us, kernel_ms x K {1e3 * kernel_ms * args.steps:.0f} us", file=sys.stderr)
        env.enable_timing(False)
        c0 = {k: int(v) for k, v in zip(("settle_substeps", "resets", "lookahead_served", "lookahead_settled", "limit_path_substeps", "self_narrow_substeps",
                                         "reset_stalls"), snap0.cpu().tolist())}
        c1 = {k: env.counter(k) for k in c0}
        backlog = env.counter("lookahead_backlog") if args.reset_lookahead else 0
        local_elapsed = None
        if with_exchange:   # the same number of steps without the exchange, to price it (reported as config.exchange_us; not the headline)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                local_step(acts[i % n_act])
            barrier()
            local_elapsed = time.perf_counter() - t1
        res = dict(elapsed=elapsed, kernel_ms=kernel_ms, local_elapsed=local_elapsed, delta={k: c1[k] - c0[k] for k in c0}, backlog=backlog,
                   action_dim=d, obs_dim=env.obs_dim, settle_steps=env.cfg.settle_steps, lookahead=env.cfg.reset_lookahead)
        env.close()
        return res

    # body_contacts: the reference's contact response of EVERY link (quadruped.py:533-539) is `True`; "auto" leaves the non-foot links'
    # response off under a task that ends the episode on such a contact.  The line always says which one `value` was taken
    # with (config.body_contacts) and carries the other one next to it, timed the same way by the same process.
    main_bc = kw.get("body_contacts", default_body_contacts(gloo))
    kw["body_contacts"] = main_bc
    other_bc = "auto" if main_bc is True else True
    m = run(kw, sharded)
    m_info = None
    if not kw["info_fields"] and not args.no_info_line and not sharded:
        m_info = run(dict(kw, info_fields=True), False)
    m_bc = None
    if not args.no_body_contacts_line and not sharded and main_bc in (True, "auto"):
        m_bc = run(dict(kw, body_contacts=other_bc), False)
    elapsed_local = m["elapsed"]
    t = torch.tensor([m["elapsed"], m["local_elapsed"] or 0.0, m_info["elapsed"] if m_info else 0.0, m_bc["elapsed"] if m_bc else 0.0], dtype=torch.float64, device=dev)
    tmin = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
    rccl_ranks = 1
    if world > 1:
        rccl_ranks = torch.distributed.get_world_size()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tmin, op=torch.distributed.ReduceOp.MIN)
    elif sharded:
        rccl_ranks = torch.distributed.get_world_size()
    elapsed, local_elapsed, info_elapsed, bc_elapsed = float(t[0].item()), (float(t[1].item()) if sharded else None), float(t[2].item()), float(t[3].item())
    total_steps = n * world * args.steps
    if rank == 0:
        d = m["action_dim"]
        kavg = m["kernel_ms"] * 1e-3
        algo_bytes = 736 + 44 * d + 4 * m["obs_dim"] + (64 if kw["action_space_mode"] == "CPG" else 0)   # B(d, o); 1112 for d = 6, o = 28
        achieved = n * algo_bytes / kavg / 1e9
        # HBM bytes per launch: rocprofv3 cannot run inside this process, so the figure is the PMC byte count of the committed
        # passes of this very configuration (roofline.traffic_source names the file); null when the run differs from every profiled one
        traffic = valu = pmc = pmc_file = pmc_why = None
        try:
            key = (args.workload, n, args.reset_lookahead, args.friction_model, float(args.solver_residual_threshold), "true" if main_bc is True else str(main_bc).lower())
            if extra_kw and set(extra_kw) - {"body_contacts"}:   # (a run with other extra keywords is another configuration)
                pmc_why = "extra keywords: not a profiled configuration"
            elif gloo:
                pmc_why = "gloo test mode"
            else:
                fp, fp_why = running_fingerprint()
                pmc, pmc_file, pmc_why = pmc_for_run(os.path.join(REPO, "profiles"), key, fp, fp_why)
                pmc_file = os.path.relpath(pmc_file, REPO) if pmc_file else None
            if pmc is not None:
                traffic = (pmc["fetch_correction"] * pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024 / kavg / 1e9
                if pmc.get("sq_insts_valu"):
                    # the roof that does bound this kernel: one wave64 fp32 VALU instruction per SIMD every 4 cycles (one wave's issue rate)
                    prop = torch.cuda.get_device_properties(dev)
                    peak = prop.multi_processor_count * 4 * getattr(prop, "clock_rate", 2.4e6) * 1e3 / 4 / 1e9
                    valu = {"achieved": pmc["sq_insts_valu"] / kavg / 1e9, "peak": peak, "unit": "G wave-instructions/s",
                            "frac": pmc["sq_insts_valu"] / kavg / 1e9 / peak,
                            "note": "SQ_INSTS_VALU per launch (committed PMC passes, steady-state launches only) / live step-kernel duration, against SIMDs x clock / 4; " +
                                    (f"at N = {n} only {(n // 16) / (prop.multi_processor_count * 4):.0%} of the SIMDs hold a stepping wave"
                                     if n // 16 < prop.multi_processor_count * 4 else f"at N = {n} every SIMD holds {(n // 16) / (prop.multi_processor_count * 4):.0f} stepping waves")}
        except (OSError, KeyError, ValueError) as e:
            pmc, pmc_why = None, f"{type(e).__name__}: {e}"
        dl = m["delta"]
        resets, settle_sub = int(dl["resets"]), int(dl["settle_substeps"])
        K = m["lookahead"]
        out = {
            "metric": metric_name(args.workload, kw, n, world, args.total_envs),
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
            "config": {"backend": args.backend, "workload": args.workload, "envs_per_gpu": n, "substeps_per_env_step": kw["action_repeat"],
                       "solver_sweeps": int(300 / kw["action_repeat"]), "solver_residual_threshold": args.solver_residual_threshold, "friction_model": args.friction_model, "info_fields": bool(kw["info_fields"]), "extra_keywords": extra_kw, "dt": kw["time_step"], "actions": "U(-1,1), resident in HBM", "action_ring": n_act,
                       "auto_reset": True, "preroll_steps": args.preroll,
                       "body_contacts": main_bc,
                       # how the timed region is bracketed and how the preparation spreads the episode ages (both changed in round 4: lines of
                       # earlier rounds polled nothing and spread the ages under random actions)
                       "spin_barrier": bool(spin and not gloo), "spread_actions": "zero (robots stand still while episode ages are spread)", "spread_steps": args.spread_steps,
                       "reset_lookahead": K,
                       "reset": ((f"exact: every environment takes its own next reset state (randomizer draws of (seed, env, episode), spawn, {m['settle_steps']} settle "
                                  f"substeps), settled up to {K} episodes ahead by extra workgroups of the step kernel (settle lanes); bitwise the in-step settle")
                                 if K else f"{m['settle_steps']}-substep settle inside the step"),
                       "resets_in_timed_region": resets,
                       # resets whose state was not ready (K consecutive episodes shorter than one settle): they settle inside the step, 2500 substeps
                       # for their whole wave -- ~170 steps' time each
                       "stalls": int(dl["reset_stalls"]),
                       "lookahead_states_settled_in_timed_region": int(dl["lookahead_settled"]),
                       "lookahead_backlog_at_end": int(m["backlog"]),
                       "settle_substeps_in_timed_region": settle_sub,
                       # settle work executed inside the timed region / the settle work its resets are worth (1.0 = every reset paid for
                       # inside the region; a short region catches the bursts of the reset rate)
                       "settle_work_ratio": (settle_sub / (resets * m["settle_steps"])) if resets else None,
                       "joint_limit_path_wave_substeps": int(dl["limit_path_substeps"]),
                       "self_collision_narrow_phase_wave_substeps": int(dl["self_narrow_substeps"]),
                       "rccl_ranks": rccl_ranks,
                       "rank_ms_per_step_min_max": [1e3 * float(tmin[0].item()) / args.steps, 1e3 * elapsed / args.steps],
                       "parallelism": (f"env-sharded x{world}, actions broadcast + one all-gather of [n, o+2] per step (both skipped on one rank), results land in rank 0's rollout buffer" if sharded
                                       else f"env-sharded x{world}, no data-path collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": pmc_file, "traffic_note": pmc_why, "valu_issue": valu, "kernel": (pmc or {}).get("kernel", "k_step"), "kernel_ms": kavg * 1e3,
                         "algorithmic_bytes_per_env_step": algo_bytes,
                         "note": "achieved = N x algorithmic bytes per env-step (SURVEY 8d: 736 + 44 d + 4 o) / average k_step launch duration over the timed region (two HIP events on the kernel's stream around all its launches: includes the ~1 us between back-to-back launches, so it lies between rocprofv3's kernel average and ms_per_step); traffic = PMC bytes per launch of the committed profile named in traffic_source (2 x FETCH_SIZE + WRITE_SIZE) / the same duration, GB/s; the step is ~30 k dependent fp32 VALU instructions per wave per env-step: issue-bound, not HBM-bound -- 40 % of HBM peak would need 2.9 G env-steps/s; with body_contacts=True (the default: every link pushes back, as in PyBullet) a launch waits for the wave whose robot is going down -- kernel_ms_body_contacts_auto is the common path alone"},
        }
        ratio = out["config"]["settle_work_ratio"]
        if ratio is not None and ratio < 0.9:
            out["config"]["settle_work_note"] = (f"this timed region executed {ratio:.0%} of the settle work its {resets} resets are worth: a region of {args.steps} steps "
                                                 "catches the settle lanes' cohorts at a phase, not on average (a 1000-step region reads 1.00); the lanes run on SIMDs "
                                                 "the stepping waves leave idle, so the rate moves by less than the ratio suggests (DESIGN.md 5)")
        if m_info is not None:
            out["value_info_fields_true"] = total_steps / info_elapsed     # the default handle of QuadrupedVecEnv: every step also stores torques, foot forces, pose cache
            out["config"]["stalls_info_fields_true"] = int(m_info["delta"]["reset_stalls"])
        bc_key = lambda v: "value_body_contacts_" + ("true" if v is True else str(v).lower())
        out[bc_key(main_bc)] = out["value"]
        out["roofline"]["kernel_ms_body_contacts_" + ("true" if main_bc is True else str(main_bc).lower())] = kavg * 1e3
        if m_bc is not None:
            out[bc_key(other_bc)] = total_steps / bc_elapsed
            out["roofline"]["kernel_ms_body_contacts_" + ("true" if other_bc is True else str(other_bc).lower())] = m_bc["kernel_ms"]
            out["config"]["stalls_body_contacts_other"] = int(m_bc["delta"]["reset_stalls"])
            out["config"]["many_rows_wave_substeps_body_contacts_other"] = int(m_bc["delta"]["limit_path_substeps"])
        if sharded:
            out["config"]["local_ms_per_step"] = 1e3 * local_elapsed / args.steps
            out["config"]["exchange_us"] = 1e6 * (elapsed - local_elapsed) / args.steps
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is a property of the box, reported with the single-GPU line only
            import subprocess
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
                                    "--solver-residual-threshold", str(args.solver_residual_threshold), "--friction-model", args.friction_model], capture_output=True, text=True, timeout=400)
            if child.returncode != 0:
                raise SystemExit("cpu_baseline child failed:\n" + child.stderr[-2000:])
            out["cpu_baseline"] = json.loads(child.stdout.strip().splitlines()[-1])
    if world > 1 or sharded:
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line is the ONLY thing on the job's stdout (C stdio buffers of RCCL's banner are flushed towards stderr first)
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)


if __name__ == "__main__":
    main()
