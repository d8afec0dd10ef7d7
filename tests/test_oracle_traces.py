"""Oracle full env step vs traces recorded from the REFERENCE's own QuadrupedGymEnv (tests/golden/gen_golden.py, g15).

The reference env ran on a fake BulletClient whose rigid-body step is oracle/qso_phys.c, so every difference found
here is a difference in the restated caller semantics: action filter, action->command map, PD + PEA torques,
counters, task state machine, rewards, termination/truncation, sensor layout, reset/settle."""
import ast

import numpy as np
import pytest

from oracle.qso import Oracle
from qs_amd.config import build_config

CASES = ["jip_s1", "jip_s0", "jf_s1", "cjf_s1", "cjf2_s1", "jipppo_s1", "jfppo_s1", "bf_s1", "bfppo_s1", "cjf3_s1", "cjfppo_s1", "cart_s1", "interp_f1", "interp_f0", "raw_tau", "raw_tau_s0", "jipppohp_s1", "jfppohp_s0", "dt2_s1", "cfg0_s0"]


@pytest.mark.parametrize("name", CASES)
def test_trace(golden, name):
    g = golden("traces.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    cfg, meta = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", **kw)
    cfg.randomizer_flags = 8  # keep parameters across reset; mu of each episode comes from the recorded trace
    o = Oracle(cfg)
    assert list(g[f"{name}_keys"]) == meta["layout"]["keys"]
    acts, obs_ref, rew_ref = g[f"{name}_actions"], g[f"{name}_obs"], g[f"{name}_rew"]
    done_ref, trunc_ref = g[f"{name}_done"], g[f"{name}_trunc"]
    reset_obs, reset_at, mus = g[f"{name}_reset_obs"], list(g[f"{name}_reset_at"]), g[f"{name}_mu"]
    if kw.get("isRLGymInterface", True):   # the raw interface has no action scaling (the reference returns nan here)
        np.testing.assert_allclose(o.command_to_action(meta["landing_pose"]), g[f"{name}_landing_action"], atol=1e-6)
    ep = 0
    o.set_params(0, np.array([mus[0]]))
    ob = o.reset()
    np.testing.assert_allclose(ob[0], reset_obs[0], atol=2e-5, rtol=1e-5)
    state_ref = g[f"{name}_state"]
    # the trajectories run free (no re-synchronisation): the float32 rounding of the config limits is amplified by the
    # contact dynamics; the Cartesian mode adds the IK's square roots on top
    # (with PyBullet's solverResidualThreshold the number of sweeps of a substep can differ by one between the two replays)
    tol = 5e-3 if kw["motor_control_mode"] == "CARTESIAN_PD" else 1e-3
    resync = kw["motor_control_mode"] == "TORQUE"   # an open-loop torque script has no feedback to hold a free-running replay together
    for t in range(len(acts)):
        if resync and t > 0 and t not in reset_at:
            st = o.get_state(); st[0] = state_ref[t - 1]; o.set_state(st)
        ob, r, dn, tr = o.step(acts[t][None])
        np.testing.assert_allclose(o.get_state()[0], state_ref[t], atol=tol, rtol=1e-4, err_msg=f"state step {t}")
        assert bool(dn[0]) == bool(done_ref[t]), f"done mismatch at step {t}"
        assert bool(tr[0]) == bool(trunc_ref[t]), f"trunc mismatch at step {t}"
        np.testing.assert_allclose(ob[0], obs_ref[t], atol=tol, rtol=1e-4, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r[0], rew_ref[t], atol=2e-4, rtol=1e-4, err_msg=f"reward step {t}")
        if dn[0]:
            ep += 1
            assert reset_at[ep] == t + 1
            o.set_params(0, np.array([mus[ep]]))
            ob = o.reset()
            np.testing.assert_allclose(ob[0], reset_obs[ep], atol=2e-5, rtol=1e-5)
    assert ep == len(reset_at) - 1


@pytest.mark.parametrize("name", ["rsi_s1", "rsi_s0"])
def test_reference_state_initialisation(golden, name):
    """reset with a desired robot state (ReferenceStateInitializationWrapper -> set_robot_desired_state): no settle, zero action history."""
    g = golden("rsi.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    cfg, meta = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", **kw)
    cfg.randomizer_flags = 8
    o = Oracle(cfg)
    o.reset()
    o.set_params(0, np.array([float(g[f"{name}_mu"])]))
    ob = o.reset_to(g[f"{name}_desired"][None])
    np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(o.get_state()[0], g[f"{name}_desired"], atol=1e-12)
    for t, a in enumerate(g[f"{name}_actions"]):
        ob, r, dn, tr = o.step(a[None])
        np.testing.assert_allclose(o.get_state()[0], g[f"{name}_state"][t], atol=5e-4, rtol=1e-4, err_msg=f"state step {t}")
        np.testing.assert_allclose(ob[0], g[f"{name}_obs"][t], atol=5e-4, rtol=1e-4, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r[0], g[f"{name}_rew"][t], atol=2e-4, rtol=1e-4, err_msg=f"reward step {t}")
        assert bool(dn[0]) == bool(g[f"{name}_done"][t])


def demo_state(row, d):
    """get_demonstration_wrapper.py:61-70 -> the 37-float state row of get_state (pos3 quat4 vlin3 vang3 q12 qd12)."""
    q, qd, pos, quat, vlin, vang = row[d:d + 12], row[d + 12:d + 24], row[d + 24:d + 27], row[d + 27:d + 31], row[d + 31:d + 34], row[d + 34:d + 37]
    return np.concatenate([pos, quat, vlin, vang, q, qd])


@pytest.mark.parametrize("name", ["demo_jip", "demo_bf", "demo_jf12", "demo_cjf"])
def test_demo_tasks_and_rsi(golden, name):
    """The imitation tasks on a demonstration the reference's GetDemonstrationWrapper recorded, plain resets and resets by its
    ReferenceStateInitializationWrapper (a random row of the demonstration becomes the initial state and the demo counter)."""
    g = golden("demo.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    cfg, meta = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", demo=g[f"{name}_demo"], **kw)
    cfg.randomizer_flags = 8
    o = Oracle(cfg)
    d, demo, L = cfg.action_dim, g[f"{name}_demo"], len(g[f"{name}_demo"])
    with pytest.raises(RuntimeError):     # no demonstration yet
        o.step(np.zeros((1, d), np.float32))
    o.set_demo(meta["demo"])
    starts = list(g[f"{name}_reset_at"]) + [len(g[f"{name}_actions"])]
    for ep, el in enumerate(g[f"{name}_reset_el"]):
        o.set_params(0, np.array([float(g[f"{name}_mu"][ep])]))
        if el < 0:
            ob = o.reset()
        else:
            ob = o.reset_to(demo_state(demo[el], d)[None])
            o.set_demo_counter(int(el))
        np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"][ep], atol=5e-4, rtol=1e-4, err_msg=f"reset obs episode {ep}")
        for t in range(starts[ep], starts[ep + 1]):
            ob, r, dn, tr = o.step(g[f"{name}_actions"][t][None])
            np.testing.assert_allclose(o.get_state()[0], g[f"{name}_state"][t], atol=5e-4, rtol=1e-4, err_msg=f"state step {t}")
            np.testing.assert_allclose(ob[0], g[f"{name}_obs"][t], atol=5e-4, rtol=1e-4, err_msg=f"obs step {t}")
            np.testing.assert_allclose(r[0], g[f"{name}_rew"][t], atol=1e-6, rtol=1e-4, err_msg=f"reward step {t}")
            assert bool(dn[0]) == bool(g[f"{name}_done"][t]) and bool(tr[0]) == bool(g[f"{name}_trunc"][t]), t
            assert int(o.get_info(4)[0, 44]) == int(g[f"{name}_counter"][t])
        assert dn[0]
