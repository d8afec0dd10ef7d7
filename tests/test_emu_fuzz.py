"""Random configurations of the kernel arithmetic (host lane emulation) against the float32 oracle: the CPU twin of
tools/fuzz_parity.py / tests/test_gpu_round2.py::test_random_configurations_against_the_oracle."""
import numpy as np
import pytest

from qs_amd import config as C
from qs_amd.config import build_config
from oracle.qso import Oracle
from emu.emu import Emu

WRAPPERS = [None, None, "LANDING", "GO_TO_REST", "LANDING2", "LANDING_BACKFLIP", "LANDING_BACKFLIP2", "LANDING_CONTINUOUS"]


def draw(rng):
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    kw = dict(task_env=pick([t for t in C.TASKS if not t.endswith("_DEMO")]), observation_space_mode=pick(list(C.SENSOR_BUNDLES)),
              action_space_mode=pick(list(C.ACTION_SPACE_MODES)), motor_control_mode=pick(["PD", "PD", "CARTESIAN_PD", "TORQUE"]),
              env_randomizer_mode=pick(list(C.RANDOMIZERS)), wrapper=pick(WRAPPERS), friction_model=pick(["cone", "pyramid"]),
              solver_residual_threshold=pick([0.0, 1e-7]), enable_springs=bool(rng.integers(2)), enable_action_filter=bool(rng.integers(2)),
              payload=pick(["weld", "weld", "soft"]), mass_inertia_rule=pick(["collision_shape", "scale"]), seed=int(rng.integers(1000)),
              settle_steps=200, noise=False)
    if kw["motor_control_mode"] == "TORQUE":
        kw.update(isRLGymInterface=False, action_space_mode="DEFAULT")
    if rng.integers(4) == 0:
        kw.update(time_step=0.002, action_repeat=5)
    return kw


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_configurations(seed):
    rng = np.random.default_rng(seed)
    ran = 0
    while ran < 6:
        kw = draw(rng)
        try:
            cfg, _ = build_config(n_envs=3, auto_reset=False, **kw)
        except (ValueError, KeyError):          # combinations the reference refuses
            continue
        ran += 1
        o, e = Oracle(cfg, "f32"), Emu(cfg)
        np.testing.assert_allclose(e.reset(), o.reset(), atol=2e-3, err_msg=f"reset observation {kw}")
        for i in range(4):
            a = rng.uniform(-1, 1, size=(3, cfg.action_dim)).astype(np.float32)
            s = o.get_state()
            o.set_state(s); e.set_state(s.astype(np.float32))
            oo, ro, do, _ = o.step(a)
            eo, re_, de, _ = e.step(a)
            so, se = o.get_state(), e.get_state()
            np.testing.assert_allclose(se[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i} {kw}")
            np.testing.assert_allclose(se[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i} {kw}")
            np.testing.assert_array_equal(de, do, err_msg=f"done step {i} {kw}")
            np.testing.assert_allclose(re_, ro, atol=1e-3, rtol=5e-3, err_msg=f"reward step {i} {kw}")
            np.testing.assert_allclose(eo, oo, atol=1e-1, err_msg=f"obs step {i} {kw}")
            if do.any():
                m = do.astype(np.uint8)
                o.reset(m); e.reset(m)
