"""bench.py --gpus N must start N ranks itself: checked here without a GPU through --dry-launch, which goes
through the same Popen path as a real run but whose children only report the rank environment they were given."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=300)


def test_dry_launch_starts_one_child_per_gpu():
    r = run(["--gpus", "4", "--dry-launch", "--steps", "7"])
    assert r.returncode == 0, r.stderr
    rows = [json.loads(x) for x in r.stdout.strip().splitlines()]
    assert sorted(int(x["RANK"]) for x in rows) == [0, 1, 2, 3]
    assert all(x["RANK"] == x["LOCAL_RANK"] and x["WORLD_SIZE"] == "4" and x["MASTER_ADDR"] == "127.0.0.1" for x in rows)
    assert len({x["MASTER_PORT"] for x in rows}) == 1 and len({x["pid"] for x in rows}) == 4 and len({x["ppid"] for x in rows}) == 1


def test_more_gpus_than_visible_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""    # no JSON line that could be read as a 2-GPU result


def test_launcher_world_size_must_match_gpus():
    r = run(["--gpus", "2"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "4"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_metric_name_follows_the_workload():
    sys.path.insert(0, REPO)
    import bench
    _, kw = bench.workload("jump_in_place_8192")
    assert bench.metric_name("jump_in_place_8192", kw, 8192, 1, 0) == json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"]
    _, kw3 = bench.workload("config3_8192")
    m = bench.metric_name("config3_8192", kw3, 8192, 2, 0)
    assert "continuous-jumping-forward" in m and "jump-in-place" not in m and "x 2" in m
    assert "N=4096" in bench.metric_name("jump_in_place_8192", kw, 4096, 1, 0)


def test_visible_gpus_counts_without_a_hip_call(tmp_path, monkeypatch):
    """The launcher counts devices from the KFD topology (CPUs are nodes with simd_count 0) and honours *_VISIBLE_DEVICES."""
    sys.path.insert(0, REPO)
    import bench
    for i, simd in enumerate((0, 1024, 1024, 0, 1024)):
        d = tmp_path / str(i); d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    real_listdir, real_open = os.listdir, open
    root = "/sys/class/kfd/kfd/topology/nodes"
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(tmp_path)) if p == root else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(root, str(tmp_path)), *a, **k))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2


def test_scale_table_reads_what_scale_sh_writes(tmp_path):
    """tools/scale.sh appends {"mode", "n_gpus", "line"} rows; tools/scale_table.py computes the efficiencies from the per-N values."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import scale_table
    def line(n, value, envs, scaling="weak"):
        return {"metric": "m", "value": value, "unit": "env-steps/s", "n_gpus": n, "steps": 1000, "warmup": 50, "ms_per_step": 8192 * n / value * 1e3,
                "higher_is_better": True, "scaling": scaling, "config": {"envs_per_gpu": envs, "rccl_ranks": n, "rank_ms_per_step_min_max": [0.07, 0.08], "stalls": 0}}
    rows = [dict(mode="weak", n_gpus=1, line=line(1, 100e6, 8192)), dict(mode="weak", n_gpus=8, line=line(8, 760e6, 8192)),
            dict(mode="strong", n_gpus=1, line=line(1, 300e6, 65536, "strong")), dict(mode="strong", n_gpus=8, line=line(8, 800e6, 8192, "strong")),
            dict(mode="sharded", n_gpus=2, line=None, error="see sharded_2.err")]
    p = tmp_path / "scale.jsonl"
    p.write_text("\n".join(json.dumps(r) for r in rows) + "\n")
    t = scale_table.table(scale_table.parse(str(p)))
    eff = {(r["mode"], r["n_gpus"]): r.get("efficiency") for r in t}
    assert abs(eff[("weak", 8)] - 0.95) < 1e-9 and abs(eff[("strong", 8)] - 800 / 2400) < 1e-9 and eff[("weak", 1)] == 1.0
    assert t[-1]["value"] is None and "sharded_2.err" in t[-1]["error"]
    # and the script itself hands bench.py the flags the table relies on
    sh = open(os.path.join(REPO, "tools", "scale.sh")).read()
    assert "--total-envs 65536" in sh and "--workload config4_sharded" in sh and "scale_table.py" in sh


def test_usable_cores_follow_the_cgroup_quota(monkeypatch):
    """The CPU baseline uses as many threads as the box grants CPU time for: the affinity mask cut to the cgroup quota (the GPU box of
    round 3: 256 CPUs in the mask, cpu.max = "1600000 100000" = 16)."""
    sys.path.insert(0, REPO)
    import bench
    import builtins, io
    real_open = builtins.open
    def fake(path, *a, **k):
        if str(path) == "/sys/fs/cgroup/cpu.max":
            return io.StringIO(fake.content)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    fake.content = "1600000 100000\n"
    assert bench.usable_cores() == (16, 256, 16.0)
    fake.content = "max 100000\n"
    assert bench.usable_cores()[0] == 256
    fake.content = "50000 100000\n"           # half a CPU: still one thread
    assert bench.usable_cores()[0] == 1


def test_bench_multi_rank_code_on_two_gloo_ranks():
    """bench.py's own multi-rank path -- the launcher's children, the process group, the barrier around the timed region, the MAX and MIN
    of the elapsed time over the ranks, rank 0's JSON line as the only thing on stdout -- executed with two CPU ranks over gloo around
    a stand-in environment (tests/bench_standin.py): `--backend gloo` is that test mode.  The numbers mean nothing; the path is the
    one `--gpus N` takes with RCCL."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "quadruped-springs_amd")]))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--standin", "bench_standin:StandInVecEnv",
                          "--envs-per-gpu", "16", "--steps", "3", "--warmup", "1", "--preroll", "2", "--spread-steps", "125", "--no-info-line"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must hold the JSON line only: {out.stdout[:500]!r}"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["backend"] == "gloo" and cfg["rccl_ranks"] == 2 and cfg["envs_per_gpu"] == 16
    lo, hi = cfg["rank_ms_per_step_min_max"]
    assert 0 < lo <= hi and abs(hi - line["ms_per_step"]) < 1e-9          # the line's time is the MAX over the ranks
    assert abs(line["value"] - 2 * 16 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]   # whole-job aggregate over both ranks
    assert "cpu_baseline" not in line                                     # (reported with the single-GPU line only)
    # strong scaling (--total-envs: SURVEY 8e's 65536 split over the ranks): the same path with the environments divided, "scaling": "strong"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--standin", "bench_standin:StandInVecEnv",
                          "--total-envs", "32", "--steps", "2", "--warmup", "1", "--preroll", "1", "--spread-steps", "125", "--no-info-line", "--no-body-contacts-line"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["envs_per_gpu"] == 16
    assert abs(line["value"] - 32 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]          # the job's 32 environments, not 32 per rank
    assert line["config"]["body_contacts"] == "auto" and line["value_body_contacts_auto"] == line["value"] and "value_body_contacts_true" not in line   # (the stand-in's default; the other loop was switched off)
    # without a stand-in the mode refuses: the step has no CPU path
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--backend", "gloo", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "no CPU path" in out.stderr


def test_counters_of_another_binary_are_not_reported(tmp_path):
    """bench.py copies roofline.traffic / valu_issue from a committed counter pass (rocprofv3 cannot run inside the benchmark).  It may do
    so only when the pass was taken on the source tree the running library was built from: a mismatching fingerprint, a profile without
    one, another configuration or an overridden library yield traffic = null and the reason."""
    sys.path.insert(0, REPO)
    import bench
    key = ("jump_in_place_8192", 8192, 16, "cone", 1e-7, "auto")
    base = {"kernel": "k_step<true, false>", "workload": key[0], "envs_per_gpu": key[1], "reset_lookahead": key[2], "friction_model": key[3],
            "solver_residual_threshold": key[4], "fetch_size_kb": 4000.0, "write_size_kb": 7000.0, "fetch_correction": 2.0, "sq_insts_valu": 2e7}
    (tmp_path / "r04_f_pmc.json").write_text(json.dumps(base))                                         # round 4's files: no fingerprint
    p, f, why = bench.pmc_for_run(str(tmp_path), key, "a" * 64)
    assert p is None and f is None and "no source fingerprint" in why
    (tmp_path / "r05_a_pmc.json").write_text(json.dumps(dict(base, source_sha256="b" * 64, body_contacts="auto")))
    p, f, why = bench.pmc_for_run(str(tmp_path), key, "a" * 64)
    assert p is None and "another source tree" in why
    p, f, why = bench.pmc_for_run(str(tmp_path), key, "b" * 64)
    assert p is not None and f.endswith("r05_a_pmc.json") and why is None
    p, f, why = bench.pmc_for_run(str(tmp_path), key, None, "QS_LIB_PATH names another library than the tree's")
    assert p is None and "QS_LIB_PATH" in why
    p, f, why = bench.pmc_for_run(str(tmp_path), key[:5] + ("true",), "b" * 64)                       # the other contact setting: another configuration
    assert p is None and "no committed counter pass" in why
    # the fingerprint itself: stable, and sensitive to the build's environment switches
    b = bench._build_module()
    fp = b.source_fingerprint()
    assert fp == b.source_fingerprint() and len(fp) == 64
    old = os.environ.get("QS_HIPCC_EXTRA")
    os.environ["QS_HIPCC_EXTRA"] = "-DQS_PROBE_SWEEPS"
    try:
        assert b.source_fingerprint() != fp
    finally:
        if old is None:
            del os.environ["QS_HIPCC_EXTRA"]
        else:
            os.environ["QS_HIPCC_EXTRA"] = old
    # bench.py's own verdict on this checkout: either a fingerprint or the reason there is none
    fp2, why2 = bench.running_fingerprint()
    assert (fp2 is None) != (why2 is None)


def test_headline_command_refuses_a_short_preroll():
    r = run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--preroll", "100"])
    assert r.returncode != 0 and "two episode lengths" in r.stderr
