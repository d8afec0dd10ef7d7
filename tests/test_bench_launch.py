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
