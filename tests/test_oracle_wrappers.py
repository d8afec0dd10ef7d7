"""Landing / go-to-rest phase machine vs the REFERENCE's LandingWrapper and GoToRestWrapper.

tests/golden/gen_golden.py wraps the reference's own QuadrupedGymEnv (fake Bullet on oracle physics) in the reference's wrappers
and logs every inner env.step they issue; here each of those inner steps is one oracle step in wrapper mode, which has to
reproduce the scripted action, the swapped motor gains (through the state), the phase flag, rewards and dones."""
import ast

import numpy as np
import pytest

from oracle.qso import Oracle

INFO_LAST_ACTION, INFO_WRAPPER = 8, 10
from qs_amd.config import build_config

CASES = ["land_s1", "land_s0", "rest_s1", "rest_s0", "land2_s1", "landbf_s1", "landbf2_s1", "landc_s1", "landc2_s1"]


def replay(golden, name, make, check_every=1):
    g = golden("wrappers.npz")
    kw = ast.literal_eval(str(g[f"{name}_kwargs"]))
    o, d = make(kw)
    acts, outer = g[f"{name}_actions"], g[f"{name}_outer_of_inner"]
    reset_at, mus = list(g[f"{name}_reset_at"]), g[f"{name}_mu"]
    ep = 0
    o.set_params(0, np.array([mus[0]], np.float32))
    ob = o.reset()
    np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"][0], atol=2e-5, rtol=1e-5)
    n_scripted, phases = 0, set()
    for i in range(len(outer)):
        ob, r, dn, tr = o.step(acts[outer[i]][None])
        scripted = i > 0 and outer[i] == outer[i - 1]
        n_scripted += scripted
        info = o.get_info(INFO_WRAPPER)[0]
        assert bool(info[1]) == scripted, f"scripted flag at inner step {i}"
        phases.add(int(info[0]))
        assert bool(dn[0]) == bool(g[f"{name}_done"][i]), f"done at inner step {i}"
        assert bool(tr[0]) == bool(g[f"{name}_trunc"][i]), f"trunc at inner step {i}"
        if i % check_every == 0 or dn[0]:
            np.testing.assert_allclose(o.get_info(INFO_LAST_ACTION)[0][:d], g[f"{name}_inner_action"][i], atol=2e-5, err_msg=f"action {i}")
            np.testing.assert_allclose(o.get_state()[0], g[f"{name}_state"][i], atol=1e-3, rtol=1e-4, err_msg=f"state {i}")
            np.testing.assert_allclose(ob[0], g[f"{name}_obs"][i], atol=1e-3, rtol=1e-4, err_msg=f"obs {i}")
            np.testing.assert_allclose(r[0], g[f"{name}_rew"][i], atol=2e-4, rtol=1e-4, err_msg=f"reward {i}")
        if dn[0]:
            ep += 1
            assert reset_at[ep] == i + 1
            o.set_params(0, np.array([mus[ep]], np.float32))
            ob = o.reset()
            np.testing.assert_allclose(ob[0], g[f"{name}_reset_obs"][ep], atol=2e-5, rtol=1e-5)
    assert ep == len(reset_at) - 1
    assert n_scripted == 0 if name == "landc2_s1" else n_scripted > 15   # LandingWrapperContinuous2 never triggers (see config.py)
    return phases


@pytest.mark.parametrize("name", CASES)
def test_wrapper_trace(golden, name):
    def make(kw):
        cfg, meta = build_config(n_envs=1, noise=False, env_randomizer_mode="NONE", **kw)
        cfg.randomizer_flags = 8
        return Oracle(cfg), cfg.action_dim

    phases = replay(golden, name, make)
    print(name, "phases", phases)
