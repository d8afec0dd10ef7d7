"""Parity of the HIP path (through the C ABI of include/qs_amd.h) against the CPU oracle, on a real MI355X.

Bar (restated): float32 kernel vs float64 oracle, one env.step (10 substeps x 30 PGS sweeps) from an
identical state: |dq| <= 2e-5 rad, |dqd| <= 5e-3 rad/s, base velocity 5e-4, pose 5e-6, contact force 2 % (+0.5 N),
identical contact flags / done / truncated.  The oracle's own float32 build shows the same spread against float64
(tests/test_oracle_physics.py::test_f32_build_tracks_f64), i.e. these are rounding, not algorithm, differences."""
import ast
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_Q, TOL_QD, TOL_BASE_V, TOL_POS = 2e-5, 5e-3, 5e-4, 5e-6
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def make_pair(n, torch, oracle=True, **kw):
    from oracle.qso import Oracle
    from qs_amd.vec_env import QuadrupedVecEnv
    kw.setdefault("task_env", "JUMPING_IN_PLACE")
    kw.setdefault("observation_space_mode", "PPO_BASIC")
    kw.setdefault("enable_springs", True)
    kw.setdefault("enable_action_filter", True)
    kw.setdefault("env_randomizer_mode", "NONE")
    kw.setdefault("noise", False)
    auto_reset = kw.pop("auto_reset", False)
    keep = kw.pop("keep_params", False)
    from qs_amd.config import build_config
    cfg, meta = build_config(n_envs=n, auto_reset=auto_reset, **kw)     # built by hand so that cfg can be edited before qs_create
    if keep:
        cfg.randomizer_flags |= 8
    v = QuadrupedVecEnv.from_config(cfg, meta, load_demo=False)
    return (Oracle(cfg) if oracle else None), v, cfg


def test_native_library_is_the_path():
    from qs_amd import lib
    assert os.path.exists(lib.LIB_PATH)
    assert b"gfx950" in lib.load().qs_version()
    # the binary under test is the one this tree's sources compile to: build.py passes the tree's fingerprint into the library, and the
    # parity gate (tools/gate.sh, profiles/validated_libraries.jsonl) records which fingerprint it validated
    if not os.environ.get("QS_LIB_PATH"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("qs_build", os.path.join(REPO, "quadruped-springs_amd", "build.py"))
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        assert lib.source_sha() == b.source_fingerprint(), "libqs_hip.so was not built from this source tree: python quadruped-springs_amd/build.py --force"


def test_reset_settle_and_static_stance(torch_cuda):
    o, v, cfg = make_pair(8, torch_cuda)
    oo, vo = o.reset(), v.reset()
    np.testing.assert_allclose(vo, oo, atol=5e-4)
    f = v.get_info("foot_force").cpu().numpy()
    np.testing.assert_allclose(f, o.get_info(0), rtol=2e-3)
    np.testing.assert_allclose(f.sum(axis=1), 12.01301 * 9.8, rtol=5e-3)   # K6: the feet carry the robot's weight
    assert np.all(v.get_info("foot_contact").cpu().numpy() == 1)


CASES = [
    dict(),
    dict(enable_springs=False, enable_action_filter=False, observation_space_mode="ARS_BASIC"),
    dict(action_space_mode="DEFAULT", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC_CONTACT"),
    dict(motor_control_mode="CARTESIAN_PD", observation_space_mode="CARTESIAN_NO_IMU"),
    dict(task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD", action_space_mode="SYMMETRIC_NO_HIP"),
    dict(task_env="JUMPING_FORWARD_PPO", observation_space_mode="LANDING_SENSOR"),
    dict(task_env="JUMPING_IN_PLACE_PPO", observation_space_mode="PPO_BASIC_X"),
    dict(task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP"),
    dict(task_env="BACKFLIP_PPO", observation_space_mode="PPO_BACKFLIP"),
    dict(task_env="CONTINUOUS_JUMPING_FORWARD3", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD"),
    dict(task_env="CONTINUOUS_JUMPING_FORWARD_PPO", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD"),
    dict(solver_residual_threshold=1e-7),     # PyBullet's default solverResidualThreshold (per-environment early exit)
    dict(time_step=0.002, action_repeat=5),   # BASELINE.json config 2: dt = 1/500 s, 60 solver sweeps
    dict(task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD", env_randomizer_mode="SPRING_RANDOMIZER", seed=9),  # config 3
    dict(task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", action_space_mode="CPG", env_randomizer_mode="TEST_RANDOMIZER", seed=4),  # config 5
    dict(friction_model="cone"),                                              # PyBullet's implicit cone friction (enableConeFriction)
    dict(friction_model="cone", task_env="JUMPING_FORWARD", action_space_mode="DEFAULT", env_randomizer_mode="GROUND_RANDOMIZER", seed=2),
    dict(friction_model="pyramid", solver_residual_threshold=0.0, contact_erp=0.2, contact_slop=0.0),          # round 1's solver settings
    dict(env_randomizer_mode="MASS_RANDOMIZER", mass_inertia_rule="scale", seed=6),                           # the other mass-to-inertia rule
    dict(env_randomizer_mode="MASS_RANDOMIZER", seed=6),                                                      # Bullet's collision-shape rule
    dict(body_contacts=True, self_collision=False),               # a task that ends on the contact, with the links' response forced on
    dict(info_fields=False),
    # the hand-over to the full build inside an env step (round 4) under the layers whose prologue it has to carry across: the Hopf CPG
    # (oscillator state, per-substep commands), a landing wrapper (phase, timers, scripted gains), a 12-value action space
    dict(task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", action_space_mode="CPG", env_randomizer_mode="GROUND_RANDOMIZER", seed=4,
         body_contacts=True, self_collision=False),
    dict(wrapper="LANDING", body_contacts=True),
    dict(action_space_mode="DEFAULT", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC_CONTACT", body_contacts=True, friction_model="pyramid"),
]


def record_jsonl(name, rec):
    """one line per test into gpurun_out/<name>.jsonl (copied to profiles/ per round by tools/collect_profiles.py)"""
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", name + ".jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("kw", CASES)
def test_env_step_parity_resynced(torch_cuda, kw):
    """One env.step after another from the oracle's state: 100 steps of scripted hops, then 60 in which half of the robots start the step
    thrown at the floor.  An env step in which a non-foot link touched the ground went through the many-rows solve (the links' contact
    response is on by default since round 5) -- an impact, a discontinuity of the step map.  Round 5 held those rows to fixed loose bounds
    (0.5 m/s, 2 rad/s: the size of the effect the default exists for, and of a miscompiled library's error) and left their torques, foot
    forces and end-of-episode rewards out.  Now (tests/yardstick.py): every output of such a row -- state, observation, reward, torque,
    foot forces, get_reward_end_episode -- may sit `strict tolerance + 5 x |oracle float32 - oracle float64|` from the float64 oracle, per
    group of like quantities, widened by the float64 oracle's own step from twelve (then 48 more) states 1e-6 away only where that does
    not cover the device; a row that is still outside is an outlier -- at most one per case, within 3 x its bound (kw19 step 59 is one:
    0.05 rad/s in the trunk's roll rate against a bound of 0.027, where round 5 found the oracle itself to jump by 0.6 rad/s from states
    1e-6 away); all other rows stay strict.  The distribution of |device - oracle64| over the impact rows is recorded NEXT TO the oracle's own
    float32 / float64 spread (gpurun_out/impact_parity.jsonl) and held to it: 90th and 99th percentile within 2 x (+ the strict tolerance)."""
    import yardstick as Y
    from oracle.qso import Oracle
    n = 16
    o, v, cfg = make_pair(n, torch_cuda, **kw)
    o32 = Oracle(cfg, "f32")
    o.reset(); o32.reset(); v.reset()
    rec = Y.resynced_parity(o, o32, Y.VecEnvDevice(v), cfg, v.meta["layout"], steps=100, thrown_steps=60)
    out = dict(case=CASES.index(kw), kw={k: str(x) for k, x in kw.items()}, **{k: x for k, x in rec.items() if not isinstance(x, dict)})
    assert len(rec["outliers"]) <= 1, rec["outliers"]      # (each within 3 x its bound: asserted where it was found)
    for name, _, _, tol in Y.STATE_GROUPS:
        dev, own = Y.percentiles(rec["impact_dev"][name]), Y.percentiles(rec["impact_own"][name])
        out[name] = dict(device_p50_p90_p99=dev, oracle32_p50_p90_p99=own)
        if len(rec["impact_dev"][name]) >= 30:
            assert dev[1] <= tol + 2 * own[1] and dev[2] <= tol + 2 * own[2], f"{name}: |device - oracle64| p50 / p90 / p99 {dev} against the oracle's own {own}"
    print("impact-step parity:", out)
    record_jsonl("impact_parity", out)
    assert rec["impact_env_steps"] >= 30, rec["impact_env_steps"]
    o.close(); o32.close(); v.close()


@pytest.mark.parametrize("name", ["jip_s1", "jip_s0", "jf_s1", "cjf_s1", "cjf2_s1", "jipppo_s1", "jfppo_s1", "bf_s1", "bfppo_s1", "cjf3_s1", "cjfppo_s1", "cart_s1", "interp_f1", "interp_f0", "raw_tau", "raw_tau_s0", "jipppohp_s1", "jfppohp_s0", "dt2_s1", "cfg0_s0"])
def test_reference_traces(torch_cuda, golden, name):
    """Traces recorded from the REFERENCE's QuadrupedGymEnv (tests/golden/traces.npz).  Trajectories are chaotic, so the
    device state is re-synchronised to the recorded state before every step; what is compared is one full env.step."""
    g = golden("traces.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    _, v, cfg = make_pair(1, torch_cuda, oracle=False, keep_params=True, **kw)
    acts, obs_ref, rew_ref = g[f"{name}_actions"], g[f"{name}_obs"], g[f"{name}_rew"]
    done_ref, trunc_ref, state_ref = g[f"{name}_done"], g[f"{name}_trunc"], g[f"{name}_state"]
    reset_obs, reset_at, mus = g[f"{name}_reset_obs"], list(g[f"{name}_reset_at"]), g[f"{name}_mu"]
    ep = 0
    v.set_params("mu", np.array([[mus[0]]], np.float32))
    np.testing.assert_allclose(v.reset()[0], reset_obs[0], atol=1e-3)
    for t in range(len(acts)):
        if t > 0 and t not in reset_at:
            v.set_state(state_ref[t - 1][None].astype(np.float32))
        ob, r, dn, infos = v.step(acts[t][None].astype(np.float32))
        assert bool(dn[0]) == bool(done_ref[t]), f"done mismatch at step {t}"
        assert bool(infos[0].get("TimeLimit.truncated", False)) == bool(trunc_ref[t])
        sv = v.get_state().cpu().numpy()[0]
        np.testing.assert_allclose(sv[13:25], state_ref[t][13:25], atol=1e-4, err_msg=f"q step {t}")
        np.testing.assert_allclose(sv[25:], state_ref[t][25:], atol=2e-2, err_msg=f"qd step {t}")
        # the reference's reward of the step (incl. the end-of-episode bonus); tolerance of the CPU twin (tests/test_oracle_traces.py)
        np.testing.assert_allclose(r[0], rew_ref[t], atol=5e-4, rtol=1e-3, err_msg=f"reward step {t}")
        if t == 0 or (t - 1) not in [x - 1 for x in reset_at[1:]]:
            np.testing.assert_allclose(ob[0], obs_ref[t], atol=2e-2, rtol=1e-3, err_msg=f"obs step {t}")
        if dn[0]:
            ep += 1
            v.set_params("mu", np.array([[mus[ep]]], np.float32))
            np.testing.assert_allclose(v.reset()[0], reset_obs[ep], atol=1e-3)


@pytest.mark.parametrize("name", ["land_s1", "land_s0", "rest_s1", "rest_s0", "land2_s1", "landbf_s1", "landbf2_s1", "landc_s1", "landc2_s1"])
def test_reference_wrapper_traces(torch_cuda, golden, name):
    """Inner env.step calls of the REFERENCE's LandingWrapper / GoToRestWrapper (tests/golden/wrappers.npz) vs the on-device
    phase machine: scripted actions, swapped gains (through the state), scripted flag, rewards, dones.  Re-synchronised to
    the recorded dynamic state before every step; the phase machine's own state runs free."""
    g = golden("wrappers.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    _, v, cfg = make_pair(1, torch_cuda, oracle=False, keep_params=True, **kw)
    acts, outer, state_ref = g[f"{name}_actions"], g[f"{name}_outer_of_inner"], g[f"{name}_state"]
    reset_at, mus, d = list(g[f"{name}_reset_at"]), g[f"{name}_mu"], cfg.action_dim
    ep = 0
    v.set_params("mu", np.array([[mus[0]]], np.float32))
    np.testing.assert_allclose(v.reset()[0], g[f"{name}_reset_obs"][0], atol=1e-3)
    phases = set()
    for i in range(len(outer)):
        if i > 0 and i not in reset_at:
            v.set_state(state_ref[i - 1][None].astype(np.float32))
        ob, r, dn, infos = v.step(acts[outer[i]][None].astype(np.float32))
        assert infos[0].get("scripted", False) == (i > 0 and outer[i] == outer[i - 1]), f"scripted flag at inner step {i}"
        phases.add(infos[0].get("phase", "policy"))
        assert bool(dn[0]) == bool(g[f"{name}_done"][i]), f"done mismatch at inner step {i}"
        assert bool(infos[0].get("TimeLimit.truncated", False)) == bool(g[f"{name}_trunc"][i])
        np.testing.assert_allclose(v.get_info("last_action").cpu().numpy()[0, :d], g[f"{name}_inner_action"][i], atol=1e-4, err_msg=f"action {i}")
        sv = v.get_state().cpu().numpy()[0]
        np.testing.assert_allclose(sv[13:25], state_ref[i][13:25], atol=1e-4, err_msg=f"q step {i}")
        np.testing.assert_allclose(sv[25:], state_ref[i][25:], atol=2e-2, err_msg=f"qd step {i}")
        np.testing.assert_allclose(r[0], g[f"{name}_rew"][i], atol=5e-4, rtol=1e-3, err_msg=f"reward {i}")
        if i not in reset_at:
            np.testing.assert_allclose(ob[0], g[f"{name}_obs"][i], atol=2e-2, rtol=1e-3, err_msg=f"obs step {i}")
        if dn[0]:
            ep += 1
            v.set_params("mu", np.array([[mus[ep]]], np.float32))
            np.testing.assert_allclose(v.reset()[0], g[f"{name}_reset_obs"][ep], atol=1e-3)
    assert ep == len(reset_at) - 1
    expect = dict(rest_s1={"policy", "rest"}, rest_s0={"policy", "rest"}, landbf_s1={"policy", "take_off"}, landc2_s1={"policy"})
    assert phases == expect.get(name, {"policy", "take_off", "landing"})


# share of the shadowed env-steps that test_full_size_oracle_sampled compares STRICTLY (no foot touching down or lifting off inside the step):
# the round-5 measurement minus 0.03 (profiles/r05_*_full_size_oracle_sampled.jsonl)
STRICT_SHARE_FLOOR = {"jump_in_place_8192": 0.88, "config2_4096": 0.88, "config3_8192": 0.88, "config4_8192": 0.87, "config5_8192": 0.65}   # measured 0.913, 0.910, 0.911, 0.907, 0.686

FULL_SIZE = {   # BASELINE.json configs[0..4] at their full sizes (configs[3] = 8 x 8192: its per-GPU share)
    "jump_in_place_8192": (8192, dict(env_randomizer_mode="GROUND_RANDOMIZER")),
    "config2_4096": (4096, dict(time_step=0.002, action_repeat=5, env_randomizer_mode="NONE")),
    "config3_8192": (8192, dict(task_env="CONTINUOUS_JUMPING_FORWARD", observation_space_mode="PPO_CONTINUOUS_JUMPING_FORWARD", env_randomizer_mode="SPRING_RANDOMIZER")),
    "config4_8192": (8192, dict(task_env="JUMPING_FORWARD", env_randomizer_mode="GROUND_RANDOMIZER")),
    "config5_8192": (8192, dict(task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", action_space_mode="CPG", env_randomizer_mode="TEST_RANDOMIZER")),
}


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_properties(torch_cuda, name):
    """BASELINE.json sizes: size-independent properties instead of the (slow) oracle."""
    torch = torch_cuda
    n, kw = FULL_SIZE[name]
    kw = dict(kw, seed=7, noise=True)
    _, v, cfg = make_pair(n, torch, oracle=False, **kw)
    _, w, _ = make_pair(64, torch, oracle=False, **kw)
    ov, ow = v.reset(), w.reset()
    assert np.array_equal(ov[:64], ow)                       # batch-size invariance (counter-based RNG, no cross-env state)
    f = v.get_info("foot_force").cpu().numpy()
    par = v.get_info("params").cpu().numpy()
    # also under the mass randomizer: it takes the payload and the leg-mass changes out of the trunk (env_randomizer.py:61-65)
    np.testing.assert_allclose(f.sum(axis=1), 12.01301 * 9.8, rtol=1e-2)
    if kw["env_randomizer_mode"] == "TEST_RANDOMIZER":
        assert 0.3 < par[:, 20].mean() < 0.7 and par[:, 20].max() <= 1.0          # payload U(0, 1) kg (:78-83)
    mu = par[:, 0]
    if kw["env_randomizer_mode"] != "NONE":
        assert mu.min() >= 0.5 and mu.max() <= 1.0 and 0.70 < mu.mean() < 0.80   # env_randomizer.py:287-289
    _, v2, _ = make_pair(n, torch, oracle=False, **kw)
    assert torch.equal(v2.reset_tensor(), v._obs)
    gg = torch.Generator(device="cpu").manual_seed(0)
    for i in range(20):
        a = (torch.rand((n, cfg.action_dim), generator=gg) * 2 - 1).to(v.device)
        o, r, d, t = v.step_tensor(a)
        o2, r2, d2, t2 = v2.step_tensor(a)
        # determinism: two handles, same inputs -> bitwise identical outputs (K11)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
        if i < 5:   # and the first 64 environments do not depend on how many others share the launch
            o64 = w.step_tensor(a[:64].contiguous())[0]
            assert torch.equal(o[:64], o64)
    s = v.get_state()
    assert torch.isfinite(s).all()
    np.testing.assert_allclose(torch.linalg.norm(s[:, 3:7], dim=1).cpu().numpy(), 1.0, atol=1e-5)
    assert float(s[:, 25:].abs().max()) <= cfg.vel_cap + 1e-4   # K10


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_oracle_sampled(torch_cuda, name):
    """BASELINE.json sizes under the bench's settings (auto-reset, 16 look-ahead reset states, the configuration's randomizer): 256 of
    the launch's environments -- four blocks of 64 at random places -- are shadowed by the oracle.  Before every step the oracle is
    re-seated in the DEVICE's state of those environments (rigid-body state and contact warm start; the device itself runs free), and
    its step is held against theirs with this file's tolerances: pose, velocities, joint state, observation, reward, done / truncation
    flags, and the reset observations of the environments that finish.

    25 600 env-steps per configuration meet what 1 600 do not: states AT a discontinuity of the step map.  Two kinds showed up when the
    test was written (tools/diag/r04_full_size_case.py, r04_full_size_trace.py): a foot touching down inside the env step whose distance
    sits within float32 rounding of the 0.727 mm contact range at some substep -- the kernel sees the contact one substep (2 ms under
    configs[1]) later than the oracle's two builds, 1.3e-3 rad in that leg's calf at the end of the step --, and a state where the
    oracle's OWN float32 build parts from its float64 build by 5e-3 in a base velocity and the kernel lands on the float32 side.  So:
    an environment may deviate by 5 x what the oracle's two precisions deviate from the same state (the fuzz's yardstick), and one
    whose set of touching feet changes inside the step is held to loose bounds only; the strict comparison covers all the others, and
    the test counts how few are not under it."""
    torch = torch_cuda
    from oracle.qso import Oracle
    from qs_amd.config import build_config
    from qs_amd.vec_env import QuadrupedVecEnv
    n, kw = FULL_SIZE[name]
    kw = dict(dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True), **kw, seed=7, noise=False)
    v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, **kw)
    rng = np.random.default_rng(sum(map(ord, name)))
    blocks = [int(b) * 64 for b in sorted(rng.choice(n // 64, size=4, replace=False))]
    oracles = [Oracle(build_config(n_envs=64, auto_reset=True, env_id_offset=b, **kw)[0]) for b in blocks]
    oracles32 = [Oracle(build_config(n_envs=64, auto_reset=True, env_id_offset=b, **kw)[0], "f32") for b in blocks]
    d, dt = v.action_dim, float(v.cfg.dt)
    ov = v.reset_tensor().cpu().numpy()
    for o, p, b in zip(oracles, oracles32, blocks):
        np.testing.assert_allclose(ov[b:b + 64], o.reset(), atol=5e-4, err_msg=f"reset observation, block {b}")
        p.reset()
    import yardstick as Y
    out = Y.oracle_sampled_parity(Y.FreeVecEnv(v), n, [(b, 64) for b in blocks], oracles, oracles32, v.meta["layout"], d, steps=100, rng=rng)
    finished, strict, switching = out["episodes_finished"], out["strict"], out["switching"]
    assert finished > 0, "no episode of the shadowed environments ended: the run did not cover a reset"
    # what share of the env-steps the strict comparison covered -- recorded (gpurun_out/full_size_oracle_sampled.jsonl, copied to profiles/ per
    # round) and held to the share measured in round 5 minus a margin, not to a round number ("a bar, not a measurement")
    share = strict / max(strict + switching, 1)
    rec = dict(config=name, n_envs=n, env_steps_compared=strict + switching, strict=strict, switching=switching, strict_share=round(share, 4),
               episodes_finished=finished, floor=STRICT_SHARE_FLOOR.get(name),
               # what puts an env step outside the strict comparison (the feet on the ground before / after the step), and how the yardstick held it
               **{k: out[k] for k in ("switching_by_cause", "env_steps_that_needed_the_second_yardstick", "switching_abs_dev")})
    print("full-size oracle-sampled parity:", rec)
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "full_size_oracle_sampled.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    assert share >= STRICT_SHARE_FLOOR.get(name, 0.6), f"only {strict} of {strict + switching} env-steps were compared strictly ({share:.3f})"
    assert v.counter("reset_stalls") == 0
    for o in oracles + oracles32:
        o.close()
    v.close()


def test_terminal_observations_of_fall_ended_episodes_at_the_headline_size(torch_cuda):
    """The output the body_contacts=True default exists for, at the size the headline is quoted on (missing #1): N = 8192,
    jump-in-place, ground randomizer, auto-reset with 16 look-ahead states, every link's contact response on.  The device runs free; four
    blocks of 256 environments at random places are shadowed by the float64 and the float32 oracle, re-seated before every step in the
    device's state.  For at least 2000 episodes that end by a FALL: infos[i]["terminal_observation"] (qs_get_info(QS_INFO_TERMINAL_OBS): the
    array step_wait hands out) and the terminal step's reward against the float64 oracle's, per sensor, the 50th / 90th / 99th
    percentile of |device - oracle64| NEXT TO those of |oracle32 - oracle64| over the same episodes -> gpurun_out/terminal_observation_parity.json
    (committed as profiles/r06_terminal_observation_parity.json).  Asserted: the device's 90th and 99th percentile within 2 x the float32
    oracle's own (+ the strict tolerance of the quantity); episodes that end on one side only under 1 % of the falls."""
    import yardstick as Y
    from oracle.qso import Oracle
    from qs_amd.config import build_config
    from qs_amd.vec_env import QuadrupedVecEnv
    torch = torch_cuda
    n = 8192
    kw = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
              env_randomizer_mode="GROUND_RANDOMIZER", seed=11, noise=False)
    v = QuadrupedVecEnv(num_envs=n, auto_reset=True, reset_lookahead=16, **kw)
    assert v.cfg.body_contacts == 1
    v.reset_tensor()
    rng = np.random.default_rng(6)
    blocks = [(int(b) * 256, 256) for b in sorted(rng.choice(n // 256, size=4, replace=False))]
    made = []

    def make_oracle(b, k, precision):
        made.append(Oracle(build_config(n_envs=k, auto_reset=True, env_id_offset=b, **kw)[0], precision))
        made[-1].set_threads(Y.usable_cores())     # (the two builds are two libraries on one OpenMP runtime)
        return made[-1]

    rec = Y.terminal_observation_parity(Y.FreeVecEnv(v), n, blocks, make_oracle, v.meta["layout"], v.action_dim, target=2000, max_steps=700, seed=3)
    rec.update(config="jump_in_place_8192", n_envs=n, body_contacts=True, blocks=blocks, oracle_threads=Y.usable_cores(), reset_stalls=v.counter("reset_stalls"))
    print("terminal observations of fall-ended episodes:", json.dumps(rec))
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "terminal_observation_parity.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass
    made[0].set_threads(1); made[1].set_threads(1)
    assert rec["fall_ended_episodes"] >= 2000, rec["fall_ended_episodes"]
    assert rec["ended_on_the_device_only"] + rec["ended_in_the_oracle_only"] <= 0.01 * rec["fall_ended_episodes"], rec
    Y.assert_inside_own_spread(rec)
    v.close()


def test_auto_reset_and_terminal_observation(torch_cuda):
    o, v, cfg = make_pair(32, torch_cuda, auto_reset=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=3)
    oo, vo = o.reset(), v.reset()
    np.testing.assert_allclose(vo, oo, atol=5e-4)
    rng = np.random.default_rng(2)
    seen = 0
    for i in range(160):
        a = rng.uniform(-1, 1, size=(32, 6)).astype(np.float32)
        if i % 40 > 25:   # explosive extension -> flight, bad landings, terminations
            a[:16] = [0.0, -1.0, 1.0, 0.0, -1.0, 1.0]
        s = o.get_state(); o.set_state(s); v.set_state(s.astype(np.float32))
        oo, ro, do, to = o.step(a)
        vo, rv, dv, infos = v.step(a)
        np.testing.assert_array_equal(dv, do)
        np.testing.assert_allclose(vo, oo, atol=TOL_QD)     # rows of finished envs already hold the post-reset observation
        for k in np.nonzero(dv)[0]:
            seen += 1
            np.testing.assert_allclose(infos[k]["terminal_observation"], o.get_info(9)[k], atol=TOL_QD)
            assert infos[k]["TimeLimit.truncated"] == bool(to[k])
    assert seen > 0
    assert v.stats()["resets"] == seen + 32


def test_lookahead_reset_states_are_settled(torch_cuda):
    _, v, cfg = make_pair(256, torch_cuda, oracle=False, auto_reset=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5)
    v.close()
    from qs_amd.vec_env import QuadrupedVecEnv
    v = QuadrupedVecEnv(num_envs=256, auto_reset=True, reset_lookahead=4, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                        enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5, noise=False)
    v.reset()
    rng = np.random.default_rng(3)
    n_done = 0
    for i in range(80):
        obs, rew, done, infos = v.step(rng.uniform(-1, 1, size=(256, 6)).astype(np.float32))
        if done.any():
            n_done += int(done.sum())
            st = v.get_state().cpu().numpy()[done]
            assert np.all(np.abs(st[:, 2] - 0.328) < 0.01)          # settled standing height
            assert np.abs(st[:, 7:13]).max() < 0.05                 # at rest
            assert np.all(obs[done][:, 27] == 0)
    assert n_done > 0


def test_lookahead_states_are_consumed_and_resettled(torch_cuda):
    """Every reset takes its environment's own next state and queues the settle of the one K episodes ahead; the settle lanes of k_step
    deliver them (fresh randomizer draws each), and with enough of them ahead nobody has to settle in place."""
    import time
    from qs_amd.vec_env import QuadrupedVecEnv
    v = QuadrupedVecEnv(num_envs=1024, auto_reset=True, reset_lookahead=8, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                        enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5, noise=False)
    v.reset()
    assert v.counter("lookahead_served") == 1024 and v.counter("lookahead_settled") == 0 and v.counter("lookahead_backlog") == 1024
    rng = np.random.default_rng(3)
    n_done, mus = 0, set()
    t0 = time.time()
    while time.time() - t0 < 30.0:
        a = rng.uniform(-1, 1, size=(1024, 6)).astype(np.float32)
        a[:, 1::3] = -1.0; a[:, 2::3] = 1.0 if (n_done // 50) % 2 else -0.5      # violent leg motions: frequent falls
        obs, rew, done, infos = v.step(a)
        if done.any():
            n_done += int(done.sum())
            st = v.get_state().cpu().numpy()[done]
            assert np.all(np.abs(st[:, 2] - 0.328) < 0.01) and np.abs(st[:, 7:13]).max() < 0.05   # settled, at rest
            mus.update(np.round(v.get_info("params").cpu().numpy()[done, 0], 6).tolist())
        if v.counter("lookahead_settled") >= 2048 and n_done > 1200:
            break
    resets = v.stats()["resets"]
    served, settled, stalls, backlog = (v.counter(k) for k in ("lookahead_served", "lookahead_settled", "reset_stalls", "lookahead_backlog"))
    assert n_done > 1200 and settled >= 2048
    assert served + stalls == resets == n_done + 1024
    assert settled + backlog <= resets       # one state wanted per reset; those in the lanes are neither delivered nor waiting
    assert stalls == 0
    assert len(mus) > 1000 and min(mus) >= 0.5 and max(mus) <= 1.0     # every reset its own friction draw
    v.settle_lanes(False)
    v.step(rng.uniform(-1, 1, size=(1024, 6)).astype(np.float32))   # lanes off: stepping stays valid
    v.close()


@pytest.mark.parametrize("n", [1, 15, 17, 100])
def test_ragged_batch_sizes_stay_inside_their_arrays(torch_cuda, n):
    """N not a multiple of the 16 environments of a wave: the tail quads of the last wave compute on a replayed record and must
    neither write past row N of any caller array nor disturb the others; results equal those of the first N environments of a
    larger batch.  Every output array is allocated with guard rows behind it."""
    import ctypes as C
    from qs_amd import lib as L
    from qs_amd.vec_env import QuadrupedVecEnv
    torch = torch_cuda
    kw = dict(auto_reset=True, reset_lookahead=3, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
              enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=9, noise=True, settle_steps=200)
    v, w = QuadrupedVecEnv(num_envs=n, **kw), QuadrupedVecEnv(num_envs=n + 23, **kw)
    o, G = v.obs_dim, 8
    guard = dict(obs=torch.full((n + G, o), 777.0, device=v.device), rew=torch.full((n + G,), 777.0, device=v.device),
                 done=torch.full((n + G,), 77, dtype=torch.uint8, device=v.device), trunc=torch.full((n + G,), 77, dtype=torch.uint8, device=v.device),
                 fused=torch.full((n + G, o + 2), 777.0, device=v.device), state=torch.full((n + G, 37), 777.0, device=v.device),
                 info=torch.full((n + G, 48), 777.0, device=v.device))
    p = lambda t: C.c_void_p(t.data_ptr())
    v.reset_tensor(); w.reset_tensor()
    rng = np.random.default_rng(n)
    for i in range(60):
        a = rng.uniform(-1, 1, size=(n + 23, 6)).astype(np.float32)
        a[:, 1::3] = -1.0; a[:, 2::3] = 1.0 if (i // 8) % 2 else -0.5       # falls: auto-resets happen
        av = torch.as_tensor(a[:n].copy(), device=v.device)
        v._stream()
        if i % 2:
            L.check(v.lib.qs_step(v.h, p(av), p(guard["obs"]), p(guard["rew"]), p(guard["done"]), p(guard["trunc"])))
            got = guard["obs"][:n]
        else:
            L.check(v.lib.qs_step_fused(v.h, p(av), p(guard["fused"])))
            got = guard["fused"][:n, :o]
        ref = w.step_tensor(torch.as_tensor(a, device=w.device))[0][:n]
        assert torch.equal(got, ref), f"step {i}"
        L.check(v.lib.qs_get_state(v.h, p(guard["state"])))
        L.check(v.lib.qs_get_info(v.h, 4, p(guard["info"])))
    for k, t in guard.items():
        tail = t[n:]
        assert bool((tail == (77 if t.dtype == torch.uint8 else 777.0)).all()), f"guard rows of {k} were written"
    assert v.stats()["resets"] > n        # more than the initial reset of every environment
    v.close(); w.close()


def test_trace_tap(torch_cuda):
    """qs_set_trace: one row per physics substep of the chosen environment (evaluation_wrapper.py / monitor_state.py taps)."""
    n = 48
    o, v, cfg = make_pair(n, torch_cuda)
    o.reset(); v.reset()
    to = o.set_trace(37)
    v.set_trace(37)
    rng = np.random.default_rng(2)
    for i in range(20):
        s = o.get_state()
        o.set_state(s); v.set_state(s.astype(np.float32))
        a = rng.uniform(-1, 1, size=(n, cfg.action_dim)).astype(np.float32)
        o.step(a); v.step(a)
        tv = v.get_trace(as_dict=False)
        np.testing.assert_allclose(tv[:, 0], to[:, 0], atol=1e-6)
        np.testing.assert_allclose(tv[-1, 1:38], v.get_state().cpu().numpy()[37], atol=0)   # last row = state after the step
        np.testing.assert_allclose(tv[:, 1:8], to[:, 1:8], atol=2e-5)
        np.testing.assert_allclose(tv[:, 14:26], to[:, 14:26], atol=5e-5)
        np.testing.assert_allclose(tv[:, 26:38], to[:, 26:38], atol=2e-2)
        np.testing.assert_allclose(tv[:, 38:62], to[:, 38:62], atol=5e-2)
        np.testing.assert_array_equal(tv[:, 66:70], to[:, 66:70])
    d = v.get_trace()
    assert d["joint_angles"].shape == (10, 12) and d["time"].shape == (10,)
    v.set_trace(None)
    v.step(a)


def test_trace_tap_across_the_hand_over(torch_cuda):
    """The traced environment's wave hands over to the full build in the middle of an env step (a neighbour comes down on a thigh in
    some substep, `body_contacts=True`): the rows of the substeps before it are the common-path build's, the rows after it the full
    build's -- one row per substep, each equal to the oracle's, the last one the state after the step."""
    n = 16
    o, v, cfg = make_pair(n, torch_cuda, task_env="NO_TASK", observation_space_mode="ENCODER", body_contacts=True, self_collision=False)
    o.reset(); v.reset()
    to = o.set_trace(5)
    v.set_trace(5)
    from scipy.spatial.transform import Rotation as Rot
    s = o.get_state()
    s[3, 2] = 0.17; s[3, 3:7] = Rot.from_euler("x", 1.45).as_quat(); s[3, 13:25] = np.tile([0.0, 1.2, -2.4], 4)    # environment 3: on its side, 5 cm above where it will lie
    o.set_state(s); v.set_state(s.astype(np.float32))
    rng = np.random.default_rng(4)
    c0, handed = v.counter("limit_path_substeps"), []
    for i in range(16):
        s = o.get_state()
        o.set_state(s); v.set_state(s.astype(np.float32))
        a = rng.uniform(-1, 1, size=(n, cfg.action_dim)).astype(np.float32)
        o.step(a); v.step(a)
        c1 = v.counter("limit_path_substeps"); handed.append(c1 - c0); c0 = c1
        tv = v.get_trace(as_dict=False)
        np.testing.assert_allclose(tv[:, 0], to[:, 0], atol=1e-6)
        np.testing.assert_allclose(tv[-1, 1:38], v.get_state().cpu().numpy()[5], atol=0)
        np.testing.assert_allclose(tv[:, 1:8], to[:, 1:8], atol=2e-5, err_msg=f"pose rows, step {i}")
        np.testing.assert_allclose(tv[:, 14:26], to[:, 14:26], atol=5e-5, err_msg=f"joint rows, step {i}")
        np.testing.assert_allclose(tv[:, 26:38], to[:, 26:38], atol=2e-2, err_msg=f"joint velocity rows, step {i}")
        np.testing.assert_array_equal(tv[:, 66:70], to[:, 66:70])
    assert any(0 < h < cfg.action_repeat for h in handed), f"no env step was handed over in its middle: many-rows substeps per step {handed}"


def test_lookahead_resets_are_bitwise_the_exact_resets(torch_cuda):
    """reset() = randomizers + spawn + 2500-substep settle (gym_env.py:278-297) depends on (seed, environment, episode) only.  A handle
    that computes those states K episodes ahead in the settle lanes (slices of action_repeat substeps through the step's own loop, other
    wave-mates, other kernels) must produce the bits of the handle that settles every reset in place: observations, rewards, done and
    truncation flags of every step, and the records' state and parameters at the end.  3100 steps: every environment goes through at
    least three resets (the 10-s limit, gym_env.py:245), most through many more."""
    from qs_amd.vec_env import QuadrupedVecEnv
    torch = torch_cuda
    n, steps = 512, 3100
    kw = dict(num_envs=n, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
              enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5, noise=True)
    a, b = QuadrupedVecEnv(reset_lookahead=8, **kw), QuadrupedVecEnv(reset_lookahead=0, **kw)
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    gen = torch.Generator(device=a.device).manual_seed(7)
    acts = torch.rand((64, n, a.action_dim), generator=gen, device=a.device) * 2 - 1
    resets = torch.zeros(n, dtype=torch.int64, device=a.device)
    for t in range(steps):
        oa, ra, da, ta = a.step_tensor(acts[t % 64])
        ob, rb, db, tb = b.step_tensor(acts[t % 64])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ta, tb), f"step {t}"
        resets += da.to(torch.int64)
        if t % 500 == 0:
            assert torch.equal(a.get_info("terminal_obs"), b.get_info("terminal_obs"))
    assert int(resets.min()) >= 3, int(resets.min())
    assert torch.equal(a.get_state(), b.get_state()) and torch.equal(a.get_info("params"), b.get_info("params"))
    assert torch.equal(a.get_info("task"), b.get_info("task")) and torch.equal(a.get_info("counters"), b.get_info("counters"))
    total = int(resets.sum())
    assert a.counter("lookahead_served") + a.counter("reset_stalls") == total + n and b.counter("reset_stalls") == 0 == b.counter("lookahead_served")
    assert a.counter("reset_stalls") == 0, a.counter("reset_stalls")
    # the exact handle settled every reset in place; the look-ahead handle did the same work in its lanes (plus the K states ahead of everyone)
    assert b.stats()["settle_substeps"] == (total + n) * 2500
    print(f"{total} resets, min per environment {int(resets.min())}; look-ahead: {a.counter('lookahead_settled')} states delivered by the lanes, "
          f"{a.stats()['settle_substeps']} settle substeps")
    a.close(); b.close()


def test_lookahead_falls_back_to_the_exact_reset_when_it_runs_dry(torch_cuda):
    """Lanes off: the K states that are ready are used up, later resets settle in place (counted as stalls) -- same bits either way, also
    after the lanes come back."""
    from qs_amd.vec_env import QuadrupedVecEnv
    torch = torch_cuda
    n = 80
    kw = dict(num_envs=n, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
              enable_action_filter=True, env_randomizer_mode="TEST_RANDOMIZER", seed=6, noise=True, settle_steps=300)
    a, b = QuadrupedVecEnv(reset_lookahead=2, **kw), QuadrupedVecEnv(reset_lookahead=0, **kw)
    a.settle_lanes(False)
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    rng = np.random.default_rng(3)
    n_done = 0
    for i in range(260):
        if i == 150:
            a.settle_lanes(True)
        act = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        act[:, 1::3] = -1.0; act[:, 2::3] = 1.0 if (i // 8) % 2 else -0.5
        act[::3] = 0.0                                                # a third of them stays up: mixed waves
        t = torch.as_tensor(act, device=a.device)
        oa, ra, da, ta = a.step_tensor(t)
        ob, rb, db, tb = b.step_tensor(t)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ta, tb), f"step {i}"
        n_done += int(da.sum())
    assert torch.equal(a.get_state(), b.get_state()) and torch.equal(a.get_info("params"), b.get_info("params"))
    assert a.counter("reset_stalls") > 20 and a.counter("lookahead_served") > n + 20 and a.counter("lookahead_settled") > 20
    a.close(); b.close()


@pytest.mark.parametrize("training,norm_reward", [(True, True), (False, False)])
def test_device_vec_normalize(torch_cuda, training, norm_reward):
    """DeviceVecNormalize vs the numpy restatement of SB3's VecNormalize (oracle/vecnorm.py) on the same raw step outputs."""
    from oracle.vecnorm import VecNormalizeRef
    from qs_amd.vec_env import QuadrupedVecEnv
    from qs_amd.vec_normalize import DeviceVecNormalize
    n = 1000   # not a multiple of the block size
    venv = QuadrupedVecEnv(num_envs=n, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                           enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5)
    env = DeviceVecNormalize(venv, training=training, norm_reward=norm_reward)
    ref = VecNormalizeRef(n, venv.obs_dim, training=training, norm_reward=norm_reward, moments_dtype=np.float64)
    sb3 = VecNormalizeRef(n, venv.obs_dim, training=training, norm_reward=norm_reward)   # batch moments in float32, as SB3 takes them
    if not training:   # evaluation as in load_model.py:114-116: statistics come from a file
        rng0 = np.random.default_rng(0)
        mean, var = rng0.normal(size=venv.obs_dim), rng0.uniform(0.5, 4.0, size=venv.obs_dim)
        env.set_stats(mean, var, 12345.0, 0.3, 2.0, 777.0)
        ref.obs_rms.mean, ref.obs_rms.var, ref.obs_rms.count = mean, var, 12345.0
        ref.ret_rms.mean, ref.ret_rms.var, ref.ret_rms.count = 0.3, 2.0, 777.0
        sb3.obs_rms.mean, sb3.obs_rms.var, sb3.obs_rms.count = mean, var, 12345.0
        sb3.ret_rms.mean, sb3.ret_rms.var, sb3.ret_rms.count = 0.3, 2.0, 777.0
    obs = env.reset()
    np.testing.assert_allclose(obs, ref.reset(env.get_original_obs()), atol=2e-5)
    sb3.reset(env.get_original_obs())
    rng = np.random.default_rng(1)
    n_done = 0
    for i in range(60):
        a = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        a[:, 1::3] = -1.0; a[:, 2::3] = 1.0 if (i // 6) % 2 else -0.5
        obs, rew, done, infos = env.step(a)
        raw_obs, raw_rew = env.get_original_obs(), env.get_original_reward()
        term_raw = venv.get_info("terminal_obs").cpu().numpy()
        o_ref, r_ref, t_ref = ref.step(raw_obs, raw_rew, done, term_raw)
        o_sb3, r_sb3, _ = sb3.step(raw_obs, raw_rew, done, term_raw)
        wide = sb3.obs_rms.var > 1e-3   # where var + epsilon does not amplify float32 summation noise of the batch mean
        np.testing.assert_allclose(obs[:, wide], o_sb3[:, wide], atol=2e-3, err_msg=f"obs vs float32-moment variant, step {i}")
        np.testing.assert_allclose(rew, r_sb3, atol=2e-3, rtol=1e-3)
        np.testing.assert_allclose(obs, o_ref, atol=2e-5, err_msg=f"obs step {i}")
        np.testing.assert_allclose(rew, r_ref, atol=2e-5, rtol=1e-5, err_msg=f"reward step {i}")
        for k in np.nonzero(done)[0]:
            np.testing.assert_allclose(infos[k]["terminal_observation"], t_ref[k], atol=2e-5)
        n_done += int(done.sum())
    s = env.get_stats()
    np.testing.assert_allclose(s["obs_mean"], ref.obs_rms.mean, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(s["obs_var"], ref.obs_rms.var, rtol=1e-5)
    np.testing.assert_allclose([s["obs_count"], s["ret_mean"], s["ret_var"], s["ret_count"]],
                               [ref.obs_rms.count, ref.ret_rms.mean, ref.ret_rms.var, ref.ret_rms.count], rtol=1e-5)
    assert n_done > 0
    if training:
        assert abs(s["obs_count"] - (1e-4 + 61 * n)) < 1e-6
    env.close()


@pytest.mark.parametrize("n", [8192 + 37, 65536, 50])
def test_vec_normalize_kernels_against_torch_float64(torch_cuda, n):
    """qs_norm_reset / qs_norm_step_io (k_norm_moments -> k_norm_finish) against SB3's formulas written out in plain torch float64
    (running_mean_std.py update_from_moments, vec_normalize.py step_wait) on the same random arrays: training and evaluation steps,
    terminal observations, the host path's compact list, the raw copies, returns reset at episode ends; sizes that end in a partial
    block, that need the most blocks, and that fit one."""
    import ctypes as C
    from qs_amd import lib as L
    torch = torch_cuda
    lib = L.load()
    o, cap, gamma, eps, clip = 28, 40, 0.99, 1e-8, 10.0
    h = C.c_void_p()
    L.check(lib.qs_norm_create(n, o, clip, clip, gamma, eps, 0, C.byref(h)))
    g = torch.Generator(device="cuda").manual_seed(3)
    p = lambda t: C.c_void_p(t.data_ptr())
    f64 = torch.float64
    scale = torch.linspace(0.01, 30.0, o, device="cuda"); shift = torch.linspace(-5.0, 5.0, o, device="cuda")

    class Rms:     # RunningMeanStd(epsilon=1e-4)
        def __init__(self, shape):
            self.mean, self.var, self.count = torch.zeros(shape, dtype=f64, device="cuda"), torch.ones(shape, dtype=f64, device="cuda"), 1e-4

        def update(self, x):
            x = x.to(f64)
            bm, bv, bc = x.mean(0), x.var(0, unbiased=False), x.shape[0]
            delta, tot = bm - self.mean, self.count + bc
            m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
            self.mean, self.var, self.count = self.mean + delta * bc / tot, m2 / tot, tot

    obs_rms, ret_rms, returns = Rms((o,)), Rms(()), torch.zeros(n, dtype=f64, device="cuda")
    norm = lambda x: torch.clamp((x.to(f64) - obs_rms.mean) / torch.sqrt(obs_rms.var + eps), -clip, clip).to(torch.float32)

    def stats():
        mean, var = np.zeros(o), np.zeros(o)
        c = [C.c_double() for _ in range(4)]
        L.check(lib.qs_norm_get_stats(h, mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), *[C.byref(x) for x in c]))
        return mean, var, np.array([x.value for x in c])

    obs = torch.randn((n, o), generator=g, device="cuda") * scale + shift
    dev = obs.clone()
    L.check(lib.qs_norm_reset(h, p(dev), 1, 1))
    obs_rms.update(obs)
    torch.testing.assert_close(dev, norm(obs), atol=2e-6, rtol=2e-6)
    for step in range(30):
        training = 0 if step >= 25 else 1
        obs = torch.randn((n, o), generator=g, device="cuda") * scale * (1 + 0.1 * step) + shift
        rew = torch.randn(n, generator=g, device="cuda") * 3 + 1
        done = (torch.rand(n, generator=g, device="cuda") < 0.02).to(torch.uint8)
        term = torch.randn((n, o), generator=g, device="cuda") * scale + shift
        tail = torch.randn((cap, o + 1), generator=g, device="cuda") * 5
        a = [obs.clone(), rew.clone(), term.clone(), tail.clone(), torch.zeros_like(obs), torch.zeros_like(rew)]
        v = lambda t: t.data_ptr()
        filled = cap                                   # rows of the list this "step" filled (qs_norm_io::tail_count; NULL = all)
        if step % 3 == 2:
            filled = int(torch.randint(0, cap + 5, (1,), generator=torch.Generator().manual_seed(step)).item())
            cnt_dev = torch.tensor([filled], dtype=torch.int64, device="cuda")
        tail_count = v(cnt_dev) if step % 3 == 2 else None
        filled = min(filled, cap)
        if step % 2 == 0:     # in place
            io = L.NormIO(obs=v(a[0]), rew=v(a[1]), done=v(done), term_obs=v(a[2]), tail_rows=v(a[3]), tail_cap=cap, raw_obs=v(a[4]), raw_rew=v(a[5]), tail_count=tail_count)
            L.check(lib.qs_norm_step_io(h, C.byref(io), training, 1, 1))
        else:                 # into other arrays (the host path's mapped block), the flags and the list copied along; the inputs stay raw
            trunc = (torch.rand(n, generator=g, device="cuda") < 0.5).to(torch.uint8)
            out = [torch.zeros_like(obs), torch.zeros_like(rew), torch.zeros_like(done), torch.zeros_like(trunc), torch.zeros_like(tail)]
            io = L.NormIO(obs=v(a[0]), rew=v(a[1]), done=v(done), trunc=v(trunc), term_obs=v(a[2]), tail_rows=v(a[3]), tail_cap=cap, raw_obs=v(a[4]), raw_rew=v(a[5]),
                          out_obs=v(out[0]), out_rew=v(out[1]), out_done=v(out[2]), out_trunc=v(out[3]), out_tail=v(out[4]), tail_count=tail_count)
            L.check(lib.qs_norm_step_io(h, C.byref(io), training, 1, 1))
            assert torch.equal(a[0], obs) and torch.equal(a[1], rew) and torch.equal(a[3], tail) and torch.equal(out[2], done) and torch.equal(out[3], trunc)
            assert torch.equal(out[4][filled:], torch.zeros_like(tail)[filled:])      # rows beyond the step's count are not written
            out[4][filled:] = tail[filled:]
            a[0], a[1], a[3] = out[0], out[1], out[4]
        if training:                                   # vec_normalize.py step_wait: the statistics and the returns move only in training
            obs_rms.update(obs)
            returns = returns * gamma + rew.to(f64)
            ret_rms.update(returns)
        r_ref = torch.clamp(rew.to(f64) / torch.sqrt(ret_rms.var + eps), -clip, clip).to(torch.float32)
        torch.testing.assert_close(a[0], norm(obs), atol=2e-6, rtol=2e-6)
        torch.testing.assert_close(a[2], norm(term), atol=2e-6, rtol=2e-6)
        torch.testing.assert_close(a[3][:filled, 1:], norm(tail[:filled, 1:]), atol=2e-6, rtol=2e-6)
        assert torch.equal(a[3][filled:], tail[filled:])                                 # stale rows of earlier steps are left alone
        torch.testing.assert_close(a[1], r_ref, atol=2e-6, rtol=2e-6)
        assert torch.equal(a[4], obs) and torch.equal(a[5], rew) and torch.equal(a[3][:, 0], tail[:, 0])   # raw copies; the list's index column
        returns = torch.where(done.bool(), torch.zeros_like(returns), returns)
        mean, var, cnt = stats()
        np.testing.assert_allclose(mean, obs_rms.mean.cpu().numpy(), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(var, obs_rms.var.cpu().numpy(), rtol=1e-9)
        np.testing.assert_allclose(cnt, [obs_rms.count, float(ret_rms.mean), float(ret_rms.var), ret_rms.count], rtol=1e-9, atol=1e-10)
    assert abs(cnt[0] - (1e-4 + 26 * n)) < 1e-6     # reset + 25 training steps; the five evaluation steps left the statistics alone
    lib.qs_norm_destroy(h)


def test_host_path_refuses_a_normalisation_handle_of_another_shape(torch_cuda):
    """qs_host_set_norm checks what can be checked BEFORE any step is launched (a failure behind the launch of a host-path step costs
    that step's results): a qs_norm created for another number of environments or another observation width is refused with the reason, the
    simulation handle is left as it was and steps on."""
    import ctypes as C
    from qs_amd import lib as L
    from qs_amd.vec_env import QuadrupedVecEnv
    lib = L.load()
    venv = QuadrupedVecEnv(num_envs=32, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, seed=2)
    venv.reset()
    for n_, o_ in ((16, venv.obs_dim), (32, venv.obs_dim + 1)):
        h = C.c_void_p()
        L.check(lib.qs_norm_create(n_, o_, 10.0, 10.0, 0.99, 1e-8, 0, C.byref(h)))
        n, o, d = C.c_int(), C.c_int(), C.c_int()
        L.check(lib.qs_norm_dims(h, C.byref(n), C.byref(o), C.byref(d)))
        assert (n.value, o.value, d.value) == (n_, o_, 0)
        assert lib.qs_host_set_norm(venv.h, h, 1, 1, 1, None, None) != 0
        assert b"normalisation handle is for" in lib.qs_last_error()
        lib.qs_norm_destroy(h)
    obs, rew, done, infos = venv.step(np.zeros((32, 6), np.float32))     # nothing was attached
    assert obs.shape == (32, venv.obs_dim) and np.isfinite(obs).all()
    venv.close()


@pytest.mark.parametrize("n,host_path", [(1, "zero"), (50, "zero"), (600, "zero"), (50, "copy"), (600, "copy")])
def test_device_vec_normalize_numpy_path_edges(torch_cuda, monkeypatch, n, host_path):
    """DeviceVecNormalize.step (the wrapped environment's host path with qs_host_set_norm) where the batch is one environment, ends in a
    partial wave, and where MORE episodes end in one step than the compact list of terminal observations holds (600 robots reach the time
    limit together: the per-environment array is fetched and normalised through the hook): every array against the numpy restatement of
    SB3's VecNormalize on the raw values, evaluation mode with loaded statistics as in load_model.py:109-137, then a training step."""
    from oracle.vecnorm import VecNormalizeRef
    from qs_amd.vec_env import QuadrupedVecEnv
    from qs_amd.vec_normalize import DeviceVecNormalize
    # QS_HOST_PATH=copy: the handle's other host path (H2D / D2H copies around the step instead of mapped host memory: the normalisation then
    # works in place on the device block and one copy brings it over); read when the handle's host path is first used
    monkeypatch.setenv("QS_HOST_PATH", "copy" if host_path == "copy" else "zero_copy")
    venv = QuadrupedVecEnv(num_envs=n, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC",
                           enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=11)
    env = DeviceVecNormalize(venv, training=False, norm_reward=False)
    ref = VecNormalizeRef(n, venv.obs_dim, training=False, norm_reward=False, moments_dtype=np.float64)
    rng0 = np.random.default_rng(0)
    mean, var = rng0.normal(size=venv.obs_dim), rng0.uniform(0.5, 4.0, size=venv.obs_dim)
    env.set_stats(mean, var, 12345.0, 0.3, 2.0, 777.0)
    ref.obs_rms.mean, ref.obs_rms.var, ref.obs_rms.count = mean, var, 12345.0
    ref.ret_rms.mean, ref.ret_rms.var, ref.ret_rms.count = 0.3, 2.0, 777.0
    obs = env.reset()
    np.testing.assert_allclose(obs, ref.reset(env.get_original_obs()), atol=2e-5)
    still = np.zeros((n, 6), np.float32)
    ends = 0
    for i in range(1003):      # standing robots run into the 1000-step limit of gym_env.py:35 in the same step
        if i == 1001:
            env.training = ref.training = True; env.norm_reward = ref.norm_reward = True
        obs, rew, done, infos = env.step(still)
        if i < 3 or i >= 995:
            raw_obs, raw_rew = env.get_original_obs(), env.get_original_reward()
            term_raw = venv.get_info("terminal_obs").cpu().numpy()
            o_ref, r_ref, t_ref = ref.step(raw_obs, raw_rew, done, term_raw)
            np.testing.assert_allclose(obs, o_ref, atol=2e-5, err_msg=f"obs step {i}")
            np.testing.assert_allclose(rew, r_ref, atol=2e-5, rtol=1e-5, err_msg=f"reward step {i}")
            for k in np.nonzero(done)[0]:
                assert infos[k]["TimeLimit.truncated"] is True
                np.testing.assert_allclose(infos[k]["terminal_observation"], t_ref[k], atol=2e-5)
            ends += int(done.sum())
        else:
            assert not done.any()
    assert ends == n, ends      # every robot's episode ended once, at the limit
    if n == 50 and host_path == "zero":     # the host-side helpers of VecNormalize's surface, on the statistics the device holds now
        st = env.get_stats()
        np.testing.assert_allclose(env.obs_rms.mean, st["obs_mean"]); assert env.ret_rms.count == st["ret_count"]
        x = rng0.normal(size=(7, venv.obs_dim))
        back = env.unnormalize_obs(env.normalize_obs(x).astype(np.float64))
        inside = np.abs((x - st["obs_mean"]) / np.sqrt(st["obs_var"] + 1e-8)) < 10        # (not clipped)
        np.testing.assert_allclose(back[inside], x[inside], rtol=2e-5, atol=2e-5)         # normalize_obs hands out float32
        r = rng0.normal(size=9)
        np.testing.assert_allclose(env.unnormalize_reward(env.normalize_reward(r)), np.clip(r, -10 * np.sqrt(st["ret_var"] + 1e-8), 10 * np.sqrt(st["ret_var"] + 1e-8)), rtol=1e-12)
    env.close()


def test_gym_env_view_task_and_robot_getters(torch_cuda):
    """QuadrupedGymEnv's task / robot getters that analysis code of the reference reads (evaluation_wrapper.py:38
    task.compute_jumping_distance; quadruped.py:141-175 orientation matrix and body rates): the jumping distance is the quantity whose
    running maximum the device keeps as _max_forward_distance (task_base.py:108-121), the matrix agrees with the device's roll / pitch /
    yaw, the body-frame rate with the PitchRate sensor's formula."""
    from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv
    env = QuadrupedGymEnv(task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC", action_space_mode="SYMMETRIC", motor_control_mode="PD",
                          enable_springs=True, enable_action_filter=True, seed=4, noise=False)
    env.reset()
    rng = np.random.default_rng(2)
    prev, rises = 0.0, 0
    for i in range(400):
        a = rng.uniform(-1, 1, 6)
        a[[1, 4]] = 1.0 if (i // 12) % 2 else -1.0        # pump the thighs: hops
        a[[2, 5]] = -a[[1, 4]]
        _, _, done, _ = env.step(a)
        m, d = env.task._max_forward_distance, env.task.compute_jumping_distance()
        assert d >= 0.0 and isinstance(env.task.is_jumping, bool)
        if done:
            env.reset(); prev = 0.0
            continue
        if m > prev + 1e-7:                               # the maximum moved in this step: by this step's distance
            assert abs(m - d) < 2e-6, (i, m, d)
            rises += 1
        prev = m
        R = env.robot.GetBaseOrientationMatrix()
        roll, pitch, yaw = env.robot.GetBaseOrientationRollPitchYaw()
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
        np.testing.assert_allclose([np.arctan2(R[2, 1], R[2, 2]), -np.arcsin(R[2, 0]), np.arctan2(R[1, 0], R[0, 0])], [roll, pitch, yaw], atol=2e-6)
        np.testing.assert_allclose(env.robot.GetTrueBaseRollPitchYawRate(), R.T @ env.robot.GetBaseAngularVelocity(), atol=1e-9)
        assert abs(env.robot.getHeight() - env.robot.GetBasePosition()[2]) == 0
    assert rises > 3, rises
    assert env._vec.cfg.body_contacts == 1      # the single-environment view: every link pushes back, as in the reference (a task notwithstanding)
    k, b, rest = env.robot.get_spring_real_stiffness_and_damping()
    assert k.shape == (12,) and set(np.unique(k)) <= {0.0, 20.0, 30.0} and abs(sum(env.robot.GetTotalMassFromURDF()) - 12.01301) < 1e-9
    env.close()
    # ... and stepping on after the fall that ended an episode leaves the robot lying ON the floor
    env = QuadrupedGymEnv(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, seed=4, noise=False)
    env.reset()
    st = env._vec.get_state().cpu().numpy()
    st[0, 2] = 0.25; st[0, 3:7] = [np.sin(0.7), 0.0, 0.0, np.cos(0.7)]      # rolled onto its side, dropped from 25 cm
    env._vec.set_state(st)
    done_seen = False
    for i in range(150):
        _, _, done, _ = env.step(np.zeros(6))
        done_seen = done_seen or done
    z = env.robot.getHeight()
    assert done_seen and 0.03 < z < 0.25 and np.abs(env.robot.GetBaseLinearVelocity()).max() < 0.5, z
    env.close()
    auto = QuadrupedGymEnv(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, body_contacts="auto")
    assert auto._vec.cfg.body_contacts == 0
    auto.close()


def test_gym_env_view_runs_a_host_side_landing_wrapper(torch_cuda, golden):
    """The N = 1 QuadrupedGymEnv view with the getters the reference's wrappers use (get_sim_time, get_landing_action,
    task.is_switched_controller, task.compute_time_for_peak_heihgt, robot._motor_model._kp/_kd).  A host-side loop with the control
    flow of landing_wrapper.py:40-69, written against those getters only, must issue the inner steps the reference's own
    LandingWrapper issued (tests/golden/wrappers.npz, land_s1): same scripted actions, same dones, same observations."""
    from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv
    g = golden("wrappers.npz")
    name = "land_s1"
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    kw.pop("wrapper")
    env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", seed=3, noise=False, **kw)
    assert env.action_dim == 6 and abs(env.env_time_step - 0.01) < 1e-12 and env.are_springs_enabled()
    state_ref, reset_at, mus = g[f"{name}_state"], list(g[f"{name}_reset_at"]), g[f"{name}_mu"]
    inner = []

    def inner_step(a):
        i = len(inner)
        if i > 0 and i not in reset_at:
            env._vec.set_state(state_ref[i - 1][None].astype(np.float32))
        out = env.step(a)
        inner.append((np.array(a, float), out))
        return out

    def wrapper_step(action):   # landing_wrapper.py:40-69 + utils/timer.py
        obs, r, done, info = inner_step(action)
        if env.task.is_switched_controller() and not done:
            timer = env.get_sim_time()
            end = timer + env.task.compute_time_for_peak_heihgt()
            while not (timer > end or done):
                timer += env.env_time_step
                _, r, done, info = inner_step(action)
            if not done:
                mm = env.robot._motor_model
                kp, kd = mm._kp, mm._kd
                mm._kp, mm._kd = 60.0, 1.5
                while not done:
                    _, r, done, info = inner_step(env.get_landing_action())
                mm._kp, mm._kd = kp, kd
        return obs, r, done, info

    ep = 0
    obs = env.reset()
    env._vec.set_params("mu", np.array([[mus[0]]], np.float32))
    assert list(obs.keys()) == [str(k) for k in golden("traces.npz")["jip_s1_keys"]]
    for t, a in enumerate(g[f"{name}_actions"]):
        _, _, done, info = wrapper_step(a)
        if done:
            assert "TimeLimit.truncated" in info
            ep += 1
            env.reset()
            env._vec.set_params("mu", np.array([[mus[ep]]], np.float32))
            np.testing.assert_allclose(env.robot._motor_model._kp, 75.0)   # the swapped gains were restored (and reset re-draws them anyway)
    assert len(inner) == len(g[f"{name}_inner_action"]) and ep == len(reset_at) - 1
    keys = list(obs.keys())
    for i, (a, (ob, r, dn, info)) in enumerate(inner):
        np.testing.assert_allclose(a, g[f"{name}_inner_action"][i], atol=1e-6, err_msg=f"inner action {i}")
        assert dn == bool(g[f"{name}_done"][i]), f"done {i}"
        if i not in reset_at:
            flat = np.concatenate([np.atleast_1d(ob[k]) for k in keys])
            np.testing.assert_allclose(flat, g[f"{name}_obs"][i], atol=2e-2, rtol=1e-3, err_msg=f"obs {i}")
            np.testing.assert_allclose(r, g[f"{name}_rew"][i], atol=5e-4, rtol=1e-3, err_msg=f"reward {i}")
    # sub-step callback (evaluation_wrapper.py:14,36-41): fired once per physics substep with that substep's time and state
    seen = []
    env.set_sub_step_callback(lambda: seen.append((env.get_sim_time(), env.robot.GetBasePosition()[2])))
    env.reset()
    env.step(np.zeros(6))
    env.step(np.zeros(6))
    assert len(seen) == 20
    np.testing.assert_allclose([s[0] for s in seen], (np.arange(20) + 1) * 1e-3, atol=1e-6)
    assert abs(seen[-1][1] - env.robot.GetBasePosition()[2]) < 1e-6 and len(set(round(s[1], 7) for s in seen)) > 3
    env.set_sub_step_callback(None)
    np.testing.assert_allclose(env.get_ac_interface().get_init_action(), env.get_settling_action())
    q = env.robot.GetMotorAngles()
    assert env.get_ac_interface()._transform_motor_command_to_action(q).shape == (6,)
    assert env.get_last_filtered_action().shape == (6,)
    env.close()


def test_fused_step_output(torch_cuda):
    """qs_step_fused writes [obs | reward | done + 2 truncated] rows: same numbers as the four arrays of qs_step."""
    from qs_amd.vec_env import QuadrupedVecEnv
    kw = dict(num_envs=300, auto_reset=True, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True,
              enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=5)
    a, b = QuadrupedVecEnv(**kw), QuadrupedVecEnv(**kw)
    a.reset(); b.reset()
    out = torch_cuda.zeros((300, a.obs_dim + 2), dtype=torch_cuda.float32, device="cuda")
    rng = np.random.default_rng(3)
    seen = set()
    for i in range(60):
        act = rng.uniform(-1, 1, size=(300, 6)).astype(np.float32)
        act[:, 1::3] = -1.0; act[:, 2::3] = 1.0 if (i // 6) % 2 else -0.5
        t = torch_cuda.from_numpy(act).cuda()
        obs, rew, done, trunc = a.step_tensor(t)
        b.step_fused(t, out)
        o = out.cpu().numpy()
        assert np.array_equal(o[:, :-2], obs.cpu().numpy()) and np.array_equal(o[:, -2], rew.cpu().numpy())
        assert np.array_equal(o[:, -1], done.cpu().numpy() + 2.0 * trunc.cpu().numpy())
        seen.update(o[:, -1].tolist())
    assert 1.0 in seen and 0.0 in seen
    a.close(); b.close()


def test_both_step_kernels_agree_bitwise(torch_cuda):
    """k_step (one wave per SIMD) and k_step_dense (two, with spills) are the same body: same results to the last bit; the
    automatic choice at the largest BASELINE.json launch size (65536 environments) is the dense one and stays physical."""
    from qs_amd.vec_env import QuadrupedVecEnv
    kw = dict(num_envs=2048, auto_reset=True, reset_lookahead=4, task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC", enable_springs=True,
              enable_action_filter=True, env_randomizer_mode="TEST_RANDOMIZER", seed=9)
    envs = []
    for variant in ("1", "2"):
        os.environ["QS_STEP_VARIANT"] = variant
        envs.append(QuadrupedVecEnv(**kw))
    os.environ.pop("QS_STEP_VARIANT")
    a, b = envs
    np.testing.assert_array_equal(a.reset(), b.reset())
    rng = np.random.default_rng(3)
    n_done = 0
    for i in range(80):
        act = rng.uniform(-1, 1, size=(2048, 6)).astype(np.float32)
        act[:, 1::3] = -1.0; act[:, 2::3] = 1.0 if (i // 8) % 2 else -0.5
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db), f"step {i}"
        n_done += int(da.sum())
    assert n_done > 50
    a.close(); b.close()
    big = QuadrupedVecEnv(num_envs=65536, auto_reset=True, reset_lookahead=2, task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC",
                          enable_springs=True, enable_action_filter=True, env_randomizer_mode="GROUND_RANDOMIZER", seed=2)
    big.reset_tensor()
    g = torch_cuda.Generator(device="cuda").manual_seed(0)
    for i in range(30):
        obs, rew, done, trunc = big.step_tensor(torch_cuda.rand((65536, 6), generator=g, device="cuda") * 2 - 1)
    st = big.get_state()
    assert torch_cuda.isfinite(st).all() and torch_cuda.isfinite(obs).all()
    assert ((st[:, 3:7].norm(dim=1) - 1).abs() < 1e-4).all() and st[:, 2].min() > 0.05 and st[:, 2].max() < 1.0
    big.close()


def test_demo_rows(torch_cuda):
    """demo_rows(): the layout GetDemonstrationWrapper._get_demo records (get_demonstration_wrapper.py:35-59)."""
    o, v, cfg = make_pair(8, torch_cuda, auto_reset=False)
    o.reset(); v.reset()
    latched = False
    for i in range(120):
        a = np.tile([0.0, 0.9, -0.9, 0.0, 0.9, -0.9] if i < 45 else [0.0, -0.8, 1.0, 0.0, -0.8, 1.0], (8, 1)).astype(np.float32)
        _, _, done, _ = v.step(a)
        if done[0] or (latched and i > 80):
            break
        rows = v.demo_rows(done).cpu().numpy()
        assert rows.shape == (8, 6 + 38)
        act, q, qd, pos, quat, lin, ang, landing = v.read_demo(rows[0])
        st = v.get_state().cpu().numpy()[0]
        np.testing.assert_array_equal(np.concatenate([q, qd]), st[13:37])
        np.testing.assert_array_equal(np.concatenate([pos, quat, lin, ang]), st[:13])
        np.testing.assert_allclose(act, v.get_info("filtered_action").cpu().numpy()[0, :6])
        sw = v.get_info("task").cpu().numpy()[0, 0] > 0.5
        latched = latched or (sw and lin[2] <= 0)
        assert bool(landing[0]) == latched
    assert latched


def test_raw_torque_interface_standing_controller(torch_cuda):
    """hopf_network.py:183-190 drives the environment with isRLGymInterface=False, motor_control_mode="TORQUE", action_repeat=1:
    raw joint torques computed on the host from the robot getters.  A joint-space PD written that way keeps the robot standing,
    and the HIP path matches the oracle step by step (both step the same torques)."""
    from oracle.qso import Oracle
    from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv
    env = QuadrupedGymEnv(isRLGymInterface=False, motor_control_mode="TORQUE", action_repeat=1, time_step=0.001, task_env="NO_TASK",
                          observation_space_mode="ENCODER", env_randomizer_mode="GROUND_RANDOMIZER", noise=False, seed=4)
    assert env.action_dim == 12
    obs = env.reset()
    o = Oracle(env._vec.cfg)
    o.reset()
    o.set_params(0, env._vec.get_info("params")[:, 0].cpu().numpy())
    q_des = np.array([0.0, 0.8, -1.6] * 4)
    for i in range(400):
        s = env._vec.get_state().cpu().numpy().astype(np.float64)
        o.set_state(s)
        q, dq = env.robot.GetMotorAngles(), env.robot.GetMotorVelocities()
        tau = np.clip(60.0 * (q_des - q) - 1.5 * dq, -23.7, 23.7)
        obs, r, done, info = env.step(tau)
        oo, _, do, _ = o.step(tau[None].astype(np.float32))
        assert not done and not do[0]
        np.testing.assert_allclose(env._vec.get_state().cpu().numpy()[0, 13:25], o.get_state()[0, 13:25], atol=2e-5)
        np.testing.assert_allclose(env._vec.get_state().cpu().numpy()[0, 25:], o.get_state()[0, 25:], atol=5e-3)
    z = env.robot.GetBasePosition()[2]
    assert 0.22 < z < 0.36 and abs(env.get_sim_time() - 0.4) < 1e-6
    assert np.abs(env.robot.GetMotorAngles() - q_des).max() < 0.1
    n_valid, n_invalid, forces, flags = env.robot.GetContactInfo()
    assert n_valid == 4 and n_invalid == 0 and abs(sum(forces) - 12.01301 * 9.8) < 6.0
    env.close()


@pytest.mark.parametrize("name", ["rsi_s1", "rsi_s0"])
def test_reference_state_initialisation(torch_cuda, golden, name):
    """qs_reset_to vs the REFERENCE's reset with set_robot_desired_state (tests/golden/rsi.npz): reset observation, then steps."""
    g = golden("rsi.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    _, v, cfg = make_pair(1, torch_cuda, oracle=False, keep_params=True, **kw)
    v.reset()
    v.set_params("mu", np.array([[float(g[f"{name}_mu"])]], np.float32))
    ob = v.reset_tensor(states=g[f"{name}_desired"][None].astype(np.float32)).cpu().numpy()
    np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"], atol=1e-4)
    np.testing.assert_allclose(v.get_state().cpu().numpy()[0], g[f"{name}_desired"], atol=1e-6)
    assert np.all(v.get_info("last_action").cpu().numpy() == 0) and np.all(v.get_info("filtered_action").cpu().numpy() == 0)
    for t, a in enumerate(g[f"{name}_actions"]):
        if t > 0:
            v.set_state(g[f"{name}_state"][t - 1][None].astype(np.float32))
        ob, r, dn, _ = v.step(a[None].astype(np.float32))
        sv = v.get_state().cpu().numpy()[0]
        np.testing.assert_allclose(sv[13:25], g[f"{name}_state"][t][13:25], atol=1e-4, err_msg=f"q step {t}")
        np.testing.assert_allclose(ob[0], g[f"{name}_obs"][t], atol=2e-2, rtol=1e-3, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r[0], g[f"{name}_rew"][t], atol=5e-4, rtol=1e-3)
        assert bool(dn[0]) == bool(g[f"{name}_done"][t])
    # the N = 1 view: set_robot_desired_state + reset
    if name == "rsi_s1":
        from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv
        env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", seed=1, noise=False, **kw)
        st = g[f"{name}_desired"]
        env.set_robot_desired_state((np.zeros(6), st[13:25], st[25:37], st[0:3], st[3:7], st[7:10], st[10:13], [0.0]))
        o = env.reset()
        np.testing.assert_allclose(env.robot.GetMotorAngles(), st[13:25], atol=1e-6)
        np.testing.assert_allclose(np.concatenate([np.atleast_1d(o[k]) for k in o]), g[f"{name}_reset_obs"], atol=1e-4)
        env.close()


@pytest.mark.parametrize("name", ["demo_jip", "demo_bf", "demo_jf12", "demo_cjf"])
def test_demo_tasks_and_rsi(torch_cuda, golden, name):
    """Imitation tasks + reference-state initialisation through the C ABI against the REFERENCE's run (tests/golden/demo.npz: its DEMO
    tasks on a demonstration its GetDemonstrationWrapper recorded, resets by its ReferenceStateInitializationWrapper)."""
    from test_oracle_traces import demo_state
    g = golden("demo.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    _, v, cfg = make_pair(1, torch_cuda, oracle=False, keep_params=True, demo=g[f"{name}_demo"], **kw)
    d, demo = cfg.action_dim, g[f"{name}_demo"]
    with pytest.raises(RuntimeError, match="qs_set_demo"):
        v.step(np.zeros((1, d), np.float32))
    v.set_demo(v.meta["demo"])
    starts = list(g[f"{name}_reset_at"]) + [len(g[f"{name}_actions"])]
    for ep, el in enumerate(g[f"{name}_reset_el"]):
        if ep == 0:
            v.reset()
        v.set_params("mu", np.array([[float(g[f"{name}_mu"][ep])]], np.float32))
        if el < 0:
            ob = v.reset()
        else:
            ob = v.reset_tensor(states=demo_state(demo[el], d)[None].astype(np.float32)).cpu().numpy()
            v.set_demo_counter(int(el))
        np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"][ep], atol=2e-3, rtol=1e-3, err_msg=f"reset obs episode {ep}")
        for t in range(starts[ep], starts[ep + 1]):
            ob, r, dn, infos = v.step(g[f"{name}_actions"][t][None].astype(np.float32))
            np.testing.assert_allclose(r[0], g[f"{name}_rew"][t], atol=1e-6, rtol=2e-4, err_msg=f"reward step {t}")
            assert bool(dn[0]) == bool(g[f"{name}_done"][t]), t
            assert bool(infos[0].get("TimeLimit.truncated", False)) == bool(g[f"{name}_trunc"][t])
            assert int(v.demo_counter()[0]) == int(g[f"{name}_counter"][t])
            np.testing.assert_allclose(v.get_state().cpu().numpy()[0][13:25], g[f"{name}_state"][t][13:25], atol=2e-2, err_msg=f"q step {t}")
            np.testing.assert_allclose(ob[0], g[f"{name}_obs"][t], atol=5e-2, rtol=1e-2, err_msg=f"obs step {t}")
            v.set_state(g[f"{name}_state"][t][None].astype(np.float32))   # float32 vs the float64 run: re-seat the rigid-body state
        assert dn[0]


def test_rsi_vec_env_and_gym_view(torch_cuda, golden):
    """ReferenceStateInitVecEnv (N environments re-seated in random rows of the demonstration by masked qs_reset_to +
    qs_set_demo_counter) and the reference's wrapper flow on the N = 1 view (task.demo_list / set_demo_counter + set_robot_desired_state)."""
    from qs_amd import QuadrupedVecEnv, ReferenceStateInitVecEnv
    from qs_amd.env.quadruped_gym_env import QuadrupedGymEnv
    torch = torch_cuda
    g = golden("demo.npz")
    kw = ast.literal_eval(str(g["demo_jip_kwargs"]))
    demo = g["demo_jip_demo"]
    L, n, d = len(demo), 64, 6
    with pytest.raises(ValueError, match="demo="):
        QuadrupedVecEnv(num_envs=4, **kw)
    venv = ReferenceStateInitVecEnv(QuadrupedVecEnv(num_envs=n, auto_reset=True, demo=demo, noise=False, seed=3, **kw), seed=11)
    obs = venv.reset_tensor()
    st = venv.get_state().cpu().numpy()
    np.testing.assert_allclose(st, venv.demo_states(demo[venv.random_el], d), atol=1e-6)
    assert np.array_equal(venv.demo_counter().cpu().numpy(), venv.random_el) and venv.random_el.max() < L - 5
    resets, firsts = np.ones(n, int), []
    for i in range(260):
        c = venv.demo_counter().cpu().numpy()
        a = torch.as_tensor(demo[np.minimum(c, L - 1), :d], device=venv.device)
        el0 = venv.random_el.copy()
        obs, rew, done, trunc = venv.step_tensor(a)
        dn = done.bool().cpu().numpy()
        r = rew.cpu().numpy()
        # the demonstration's own actions are its filtered ones, the comparison is with the unfiltered input: distance 0 here
        np.testing.assert_allclose(r, 1.0 / (L - el0), rtol=1e-5)
        assert np.array_equal(dn, (c + 1 == L) | dn) and not trunc.any()
        if dn.any():
            idx = np.nonzero(dn)[0]
            np.testing.assert_allclose(venv.get_state().cpu().numpy()[idx], venv.demo_states(demo[venv.random_el[idx]], d), atol=1e-6)
            assert np.array_equal(venv.demo_counter().cpu().numpy()[idx], venv.random_el[idx])
            firsts += [(resets[j] % 6 == 5, venv.random_el[j]) for j in idx]   # every sixth reset draws from the first fifth
            resets[idx] += 1
    assert len(firsts) > 100 and all(el < L // 5 for short, el in firsts if short) and any(el >= L // 5 for short, el in firsts if not short)
    venv.close()
    # N = 1 view driven the way reference_state_initialization_wrapper.py:25-33 drives the reference's environment
    env = QuadrupedGymEnv(env_randomizer_mode="GROUND_RANDOMIZER", seed=1, noise=False, demo=demo, **kw)
    assert env.task.demo_length == L and env.task.demo_list.shape == demo.shape
    ep = 1
    el = int(g["demo_jip_reset_el"][ep])
    env.set_robot_desired_state(QuadrupedVecEnv.read_demo(env.task.demo_list[el]))
    env.task.set_demo_counter(value=el)
    env.reset()
    assert env.task.demo_counter == el
    t0 = int(g["demo_jip_reset_at"][ep])
    _, r, dn, _ = env.step(g["demo_jip_actions"][t0])
    np.testing.assert_allclose(r, g["demo_jip_rew"][t0], rtol=2e-4)
    assert env.task.demo_counter == el + 1
    env.reset()                                   # the desired state stays set: the counter survives (task_base.py:177-179)
    assert env.task.demo_counter == el + 1
    env.close()


def test_joint_limit_solver_path(torch_cuda):
    """Raw torques drive joints into their stops (calves to the lower stop, then hips outwards and thighs back): the 6-rows-per-leg
    solver path against the FLOAT32 build of the oracle on the same float32 states (see tests/test_emu_vs_oracle.py::
    test_joint_limit_rows_all_joints for why float32); the telemetry counter shows the path was taken."""
    from oracle.qso import Oracle
    n = 32
    _, v, cfg = make_pair(n, torch_cuda, oracle=False, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK",
                          observation_space_mode="ENCODER", enable_action_filter=False, enable_springs=False, solver_residual_threshold=0.0)
    o = Oracle(cfg, "f32")
    o.reset(); v.reset()
    c0 = v.counter("limit_path_substeps")
    rng = np.random.default_rng(4)
    hit = np.zeros(3, bool)
    flips = 0
    for i in range(90):
        tau = 2.0 * rng.normal(size=(n, 12)).astype(np.float32)
        if i < 45:
            tau[:, 2::3] = -30.0                     # calves against the lower stop (-2.72 rad)
        else:
            tau[:, 0::3] = np.array([-20.0, 20.0, -20.0, 20.0], np.float32)   # hips outwards
            tau[:, 1::3] = -20.0                     # thighs against their lower stop
        s = o.get_state()
        s[:, 2] = np.maximum(s[:, 2], 0.6)           # keep the robots in the air: only the joint stops act
        s[:, 7:13] = 0
        o.set_state(s); v.set_state(s)
        o.step(tau); v.step(tau)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        # a joint within one float32 rounding of its stop may get its row on one side only: allowed for a handful of values
        dq, dqd = np.abs(sv[:, 13:25] - so[:, 13:25]), np.abs(sv[:, 25:] - so[:, 25:])
        assert dq.max() < 5e-3 and dqd.max() < 5.0, f"step {i}: {dq.max()} {dqd.max()}"    # a flipped row changes that joint's velocity by up to the bounce it stops
        flips += int((dq > 2e-5).sum()) + int((dqd > 1e-2).sum())
        q = so[:, 13:25]
        hit |= np.array([(q[:, 2::3] < -2.70).any(), (np.abs(q[:, 0::3]) > 1.03).any(), (q[:, 1::3] < -0.65).any()])
    assert hit.all(), hit
    assert flips <= 24, flips                        # of 90 x 32 x 24 values
    assert v.counter("limit_path_substeps") - c0 > 200


def test_free_running_statistics(torch_cuda):
    """No resynchronisation: 512 environments run 150 env.steps (1500 substeps, several take-offs and landings each) on the kernel
    and on the float64 oracle from the same settled states with the same actions.  Individual trajectories separate at contact-mode
    changes (rounding decides which substep a foot lands in), so the comparison is (i) the part of the batch that has not separated
    stays within loose trajectory bounds and is the large majority, (ii) the batch statistics -- height, pitch, return, episode ends --
    agree: a systematic bias in the float32 path would show here and not in the resynchronised tests."""
    from scipy.stats import ks_2samp
    n, T = 512, 150
    o, v, cfg = make_pair(n, torch_cuda, env_randomizer_mode="GROUND_RANDOMIZER", seed=21)
    o.set_threads(min(32, os.cpu_count() or 1))
    o.reset(); v.reset()
    rng = np.random.default_rng(8)
    period = rng.integers(25, 60, size=n)
    depth = rng.uniform(0.3, 1.0, size=(n, 1))
    ret_o, ret_v = np.zeros(n), np.zeros(n)
    end_o, end_v = np.full(n, T), np.full(n, T)
    zmax_o, zmax_v = np.zeros(n), np.zeros(n)
    for i in range(T):
        a = 0.3 * rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        crouch = ((i % period) < period // 2)[:, None]
        a += np.where(crouch, depth * np.tile([0.0, 1.0, -1.0], 2), depth * np.tile([0.0, -1.0, 1.0], 2)).astype(np.float32)
        _, ro, do, _ = o.step(a)
        _, rv, dv, _ = v.step(a)
        live_o, live_v = end_o == T, end_v == T
        ret_o += np.where(live_o, ro, 0); ret_v += np.where(live_v, rv, 0)
        end_o = np.where(live_o & do, i, end_o); end_v = np.where(live_v & dv, i, end_v)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        zmax_o = np.maximum(zmax_o, np.where(live_o, so[:, 2], 0)); zmax_v = np.maximum(zmax_v, np.where(live_v, sv[:, 2], 0))
    assert np.isfinite(sv).all()
    assert zmax_o.max() > 0.45 and (end_o < T).mean() > 0.05     # the script makes robots jump, and some episodes end
    same = (end_o == end_v) & (np.abs(zmax_o - zmax_v) < 5e-3)
    print("same outcome:", same.mean(), "ended:", (end_o < T).mean(), (end_v < T).mean(), "zmax:", zmax_o.mean(), zmax_v.mean(),
          "return:", ret_o.mean(), ret_v.mean())
    assert same.mean() > 0.95      # measured on MI355X: 1.0 (no environment separated within 150 steps)
    assert abs((end_o < T).mean() - (end_v < T).mean()) < 0.05
    for name, x, y in (("max height", zmax_o, zmax_v), ("return", ret_o, ret_v), ("episode end", end_o, end_v),
                       ("final pitch rate", so[:, 11], sv[:, 11])):
        assert ks_2samp(x, y).pvalue > 0.01, name
    assert abs(zmax_o.mean() - zmax_v.mean()) < 5e-3


@pytest.mark.parametrize("model", ["pyramid", "cone"])
def test_joint_limits_together_with_sliding_contacts(torch_cuda, model):
    """The 6-rows-per-leg solver path with ground contact (calves folded to their stops while the robots stand and are pushed sideways),
    both friction models, against the float32 oracle on the same float32 state; see tests/test_emu_vs_oracle.py for the CPU twin."""
    from oracle.qso import Oracle
    n = 32
    _, v, cfg = make_pair(n, torch_cuda, oracle=False, isRLGymInterface=False, motor_control_mode="TORQUE", task_env="NO_TASK",
                          observation_space_mode="ENCODER", enable_action_filter=False, enable_springs=False, friction_model=model,
                          solver_residual_threshold=0.0)     # all sweeps: the two solvers converge to the same impulses
    o = Oracle(cfg, "f32")
    o.reset(); v.reset()
    mu = np.full((n, 1), 0.5, np.float32)
    o.set_params(0, mu); v.set_params("mu", mu)
    c0 = v.counter("limit_path_substeps")
    rng = np.random.default_rng(6)
    at_stop = sliding = False
    flips = 0
    for i in range(120):
        tau = 2.0 * rng.normal(size=(n, 12)).astype(np.float32)
        tau[:, 2::3] -= 12.0
        tau[:, 0::3] += 10.0 * np.sign(np.sin(0.2 * i))
        s = o.get_state()
        o.set_state(s); v.set_state(s)
        o.step(tau); v.step(tau)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        # (a joint within one float32 rounding of its stop may get its row on one side only: a handful of values may differ by ~1e-4)
        dq, dqd = np.abs(sv[:, 13:25] - so[:, 13:25]), np.abs(sv[:, 25:] - so[:, 25:])
        dv = np.abs(sv[:, 7:13] - so[:, 7:13])
        assert dq.max() < 5e-3 and dqd.max() < 5.0 and dv.max() < 0.2, f"step {i}: {dq.max()} {dqd.max()} {dv.max()}"
        flips += int((dq > 5e-5).sum()) + int((dqd > 2e-2).sum()) + int((dv > 4e-3).sum())
        df = np.abs(v.get_info("foot_force").cpu().numpy() - o.get_info(0)) > 3e-2 * np.abs(o.get_info(0)) + 1.0
        flips += int(df.sum())
        at_stop |= bool(((so[:, 15:25:3] < -2.715) & (o.get_info(1) > 0)).any())
        sliding |= bool((np.abs(so[:, 8]) > 0.05).any())
    assert at_stop and sliding and v.counter("limit_path_substeps") - c0 > 100
    assert flips <= 40, flips                        # of 120 x 32 x 30 values


def test_create_rejects_bad_config(torch_cuda):
    import ctypes as C
    from qs_amd import lib as L
    from qs_amd.config import build_config
    cfg, _ = build_config(n_envs=4, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC")
    cfg.obs_dim = 27
    h = C.c_void_p()
    assert L.load().qs_create(C.byref(cfg), 0, C.byref(h)) != 0
    assert b"obs_dim" in L.load().qs_last_error()
    with pytest.raises(ValueError):   # gym_env.py:167-168
        build_config(motor_control_mode="TORQUE", isRLGymInterface=True)
    # use after close fails with the library's error text instead of touching freed memory
    from qs_amd.vec_env import QuadrupedVecEnv
    v = QuadrupedVecEnv(num_envs=4, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC")
    v.reset(); v.close(); v.close()
    with pytest.raises(RuntimeError, match="null"):
        v.reset()
