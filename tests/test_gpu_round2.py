"""Round-2 GPU tests (through the C ABI): contact response of the non-foot links, the calf self-collision rule, the device noise
stream, the sharded exchange on RCCL, exact-mode auto-reset in partial waves, the parameter draws of look-ahead resets."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def vec_env(n, **kw):
    from qs_amd.vec_env import QuadrupedVecEnv
    kw.setdefault("task_env", "JUMPING_IN_PLACE")
    kw.setdefault("observation_space_mode", "PPO_BASIC")
    kw.setdefault("enable_springs", True)
    kw.setdefault("enable_action_filter", True)
    kw.setdefault("env_randomizer_mode", "NONE")
    kw.setdefault("noise", False)
    kw.setdefault("auto_reset", False)
    return QuadrupedVecEnv(num_envs=n, **kw)


RAW = dict(task_env="NO_TASK", observation_space_mode="ENCODER", enable_action_filter=False, isRLGymInterface=False, motor_control_mode="TORQUE")


def fallen_states(s, rng):
    from scipy.spatial.transform import Rotation as Rot
    n = len(s)
    s = s.copy()
    s[:, :3] = [0, 0, 0.2]
    eul = np.stack([rng.choice([0.0, 1.45, -1.45, 3.0, 0.7], n), rng.uniform(-0.5, 0.5, n), np.zeros(n)], 1)
    s[:, 3:7] = Rot.from_euler("xyz", eul).as_quat()
    s[:, 7:] = 0
    s[:, 13:25] = np.tile([0.0, 1.2, -2.4], 4)
    return s


@pytest.mark.parametrize("resid", [0.0, 1e-7])
@pytest.mark.parametrize("model", ["cone", "pyramid"])
def test_fallen_robots_parity(torch_cuda, model, resid):
    """Waves that mix standing robots with robots lying on trunk, hips, thighs and calves: the 12-rows-per-leg rare path against the
    float32 oracle from identical float32 states, re-seated every step; nobody sinks into the floor."""
    from oracle.qso import Oracle
    n = 40
    v = vec_env(n, friction_model=model, solver_residual_threshold=resid, **RAW)
    o = Oracle(v.cfg, "f32")
    o.reset(); v.reset()
    rng = np.random.default_rng(5)
    s = o.get_state()
    lying = np.arange(n) % 3 != 0
    s[lying] = fallen_states(s[lying], rng)
    o.set_state(s); v.set_state(s.astype(np.float32))
    loaded = 0
    for i in range(60):
        tau = (4.0 * rng.normal(size=(n, 12))).astype(np.float32) if i > 15 else np.zeros((n, 12), np.float32)
        st = o.get_state()
        o.set_state(st); v.set_state(st.astype(np.float32))
        o.step(tau); v.step(tau)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        np.testing.assert_allclose(sv[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i}")
        np.testing.assert_allclose(sv[:, 7:13], so[:, 7:13], atol=2e-2, err_msg=f"base velocity step {i}")
        np.testing.assert_allclose(sv[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i}")
        np.testing.assert_allclose(sv[:, 25:], so[:, 25:], atol=1e-1, err_msg=f"qd step {i}")
        np.testing.assert_array_equal(v.get_info("n_invalid").cpu().numpy()[:, 0] > 0, o.get_info(5)[:, 0] > 0)
        loaded += sum(1 for k in range(n) for c in o.contacts(k) if c[1] == 0 and c[2] not in (5, 9, 13, 17) and c[5] > 1.0)
    assert loaded > 300, loaded
    assert sv[:, 2].min() > 0.03
    assert v.counter("limit_path_substeps") > 0
    v.close()


@pytest.mark.parametrize("resid", [0.0, 1e-7])
def test_fallen_robots_with_the_soft_payload(torch_cuda, resid):
    """Every kind of row at once: robots lying on trunk / hips / thighs / calves (12 rows per leg), joints at their stops, and the payload
    block on its six-row fixed constraint (payload="soft", mass randomizer) -- against the float32 oracle, re-seated every step."""
    from oracle.qso import Oracle
    n = 24
    v = vec_env(n, payload="soft", env_randomizer_mode="MASS_RANDOMIZER", solver_residual_threshold=resid, seed=3, settle_steps=300, **RAW)
    o = Oracle(v.cfg, "f32")
    o.reset(); v.reset()
    assert 0 < o.get_info(6)[:, 20].min() < 0.05          # the draws include a block of a few grams: the stiffest case for float32
    rng = np.random.default_rng(6)
    s = o.get_state()
    lying = np.arange(n) % 3 != 0
    s[lying] = fallen_states(s[lying], rng)
    o.set_state(s); v.set_state(s.astype(np.float32))
    lam = 0.0
    for i in range(40):
        tau = (4.0 * rng.normal(size=(n, 12))).astype(np.float32) if i > 10 else np.zeros((n, 12), np.float32)
        st = o.get_state()
        o.set_state(st); v.set_state(st.astype(np.float32))
        o.step(tau); v.step(tau)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        np.testing.assert_allclose(sv[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i}")
        np.testing.assert_allclose(sv[:, 7:13], so[:, 7:13], atol=2e-2, err_msg=f"base velocity step {i}")
        np.testing.assert_allclose(sv[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i}")
        b0, b1 = o.block(), v.get_info("payload_block").cpu().numpy()
        np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=5e-5, err_msg=f"block position step {i}")
        np.testing.assert_allclose(b1[:, 13:19], b0["lam"], atol=1e-3, err_msg=f"constraint impulses step {i}")
        lam = max(lam, np.abs(b0["lam"]).max())
    assert sv[:, 2].min() > 0.03 and 1e-3 < lam <= 0.5 + 1e-6, lam
    v.close()


def test_fallen_robot_comes_to_rest_on_the_floor(torch_cuda):
    """The known answer on the device: dropped on its side / belly / back without torques, the robot ends up at rest on trunk, hip and
    leg links (NO_TASK never terminates) -- within 1e-4 m of where the float64 oracle with FOUR contact points per collision primitive
    (what a btPersistentManifold can hold; qso_phys_set_manifold(1)) comes to rest, in all three attitudes (round 6: on its back the
    two-support-points cap of rounds 2-5 left the robot 1-2 mm off and creeping; a leg whose foot is in the air now has three,
    DESIGN.md 7); with body_contacts=False (round 1's behaviour) it keeps falling through the floor."""
    from oracle.qso import Oracle
    from scipy.spatial.transform import Rotation as Rot
    poses = [(1.45, 0.0), (0.0, 0.0), (3.0, 0.0)]          # side, belly, back
    out = {}
    for bc in (True, False):
        # (springs off, no torques: the folded legs stay folded and the robot really comes to rest -- with the springs unfolding the legs it
        # rocks on feet and thighs for seconds, and where it is after 3 s is float32 / float64 contact-switching history, not the cap)
        v = vec_env(16, body_contacts=bc, solver_residual_threshold=0.0, enable_springs=False, **RAW)
        v.reset()
        s = v.get_state().cpu().numpy()
        for i in range(16):
            r, p = poses[i % 3]
            s[i, :3] = [0, 0, 0.16]; s[i, 3:7] = Rot.from_euler("xyz", [r, p, 0]).as_quat()
        s[:, 7:] = 0; s[:, 13:25] = np.tile([0.0, 1.2, -2.4], 4)
        v.set_state(s)
        for _ in range(300):
            v.step(np.zeros((16, 12), np.float32))
        out[bc] = v.get_state().cpu().numpy()
        if bc:
            cfg = v.cfg
        v.close()
    assert np.abs(out[True][:, 7:13]).max() < 2e-2 and np.abs(out[True][:, 25:]).max() < 1.0      # the body rests; a free leg may still swing a little
    assert 0.03 < out[True][:, 2].min() and out[True][:, 2].max() < 0.2
    assert out[False][:, 2].max() < -0.05
    o = Oracle(cfg)                      # float64, four points per primitive, the same drop
    o.reset(); o.set_manifold(1)
    o.set_state(s.astype(np.float64))
    for _ in range(300):
        o.step(np.zeros((16, 12), np.float32))
    ref = o.get_state()
    # height and attitude (the plane's normal in trunk coordinates) are what the support points decide; where the robot has slid to on the
    # floor while it fell is the friction's stick / slip history (float32 against float64: a millimetre), held loosely
    up = lambda st: Rot.from_quat(st[:, 3:7]).as_matrix()[:, 2, :]
    for i, name in enumerate(("side", "belly", "back")):
        dz, du = np.abs(out[True][i::3, 2] - ref[i::3, 2]).max(), np.abs(up(out[True][i::3]) - up(ref[i::3])).max()
        assert dz < 1e-4 and du < 1e-3, f"on its {name}: resting height {dz:.2e} m, attitude {du:.2e} from the four-points-per-primitive oracle's"
        assert np.abs(out[True][i::3, :2] - ref[i::3, :2]).max() < 5e-3


def test_support_points_get_their_rows_once_they_can_act(torch_cuda):
    """The rule of DESIGN.md 4a on a known situation: a robot lying on its side under PD (no joint near a stop) is lifted 3 mm off its resting
    place and released.  For the first env step its trunk, hip and leg links are inside their 4-mm contact range (invalid contacts are
    reported) but 2.5 mm of gap cannot close at 0.1 m/s: no support point gets rows, the many-rows solve does not run, and the state is
    the float32 oracle's all the same -- there the rows exist and end every sweep at zero impulse.  Two env steps later the links touch
    down: the solve runs, the robot rests where it rested before."""
    from oracle.qso import Oracle
    from scipy.spatial.transform import Rotation as Rot
    n = 16
    PD = dict(task_env="NO_TASK", observation_space_mode="ENCODER", enable_action_filter=False, isRLGymInterface=False, motor_control_mode="PD")
    v = vec_env(n, **PD)
    o = Oracle(v.cfg, "f32")
    v.reset(); o.reset()
    hold = np.tile([0.0, 1.2, -2.4], (n, 4)).astype(np.float32)
    s = v.get_state().cpu().numpy()
    s[:, :3] = [0, 0, 0.16]; s[:, 3:7] = Rot.from_euler("x", 1.45).as_quat(); s[:, 7:] = 0; s[:, 13:25] = hold
    v.set_state(s)
    for _ in range(300):
        v.step(hold)
    rest = v.get_state().cpu().numpy()
    assert np.abs(rest[:, 7:13]).max() < 2e-2 and np.abs(rest[:, 13:25] - hold).max() < 0.5      # at rest, the joints held well inside their ranges (the springs pull the calves 0.3 rad off their target)
    lifted = rest.copy()
    lifted[:, 2] += 0.003; lifted[:, 7:13] = 0; lifted[:, 25:] = 0
    v.set_state(lifted); o.set_state(lifted.astype(np.float64))
    c0 = v.counter("limit_path_substeps")
    solves = []
    for i in range(6):
        st = o.get_state()
        o.set_state(st); v.set_state(st.astype(np.float32))
        o.step(hold); v.step(hold)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        np.testing.assert_allclose(sv[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {i}")
        np.testing.assert_allclose(sv[:, 7:13], so[:, 7:13], atol=2e-2, err_msg=f"base velocity step {i}")
        np.testing.assert_allclose(sv[:, 13:25], so[:, 13:25], atol=2e-4, err_msg=f"q step {i}")
        c1 = v.counter("limit_path_substeps")
        solves.append(c1 - c0); c0 = c1
        if i == 0:
            assert (v.get_info("n_invalid").cpu().numpy()[:, 0] > 0).all() and (o.get_info(5)[:, 0] > 0).all()      # in range: reported as contacts ...
    assert solves[0] == 0, solves              # ... whose rows cannot act yet: no many-rows solve in the first env step
    assert sum(solves[2:]) > 0, solves         # touch-down: the solve runs
    for _ in range(100):
        v.step(hold)
    end = v.get_state().cpu().numpy()
    assert np.abs(end[:, 2] - rest[:, 2]).max() < 2e-3 and np.abs(end[:, 7:13]).max() < 5e-2
    v.close()


def crossed(s):
    s = s.copy()
    s[:, :3] = [0, 0, 0.6]; s[:, 3:7] = [0, 0, 0, 1]; s[:, 7:] = 0
    q = np.tile([0.0, 0.8, -1.6], 4).astype(np.float32)
    q[0], q[3] = 0.55, -0.55
    s[:, 13:25] = q
    return s


def test_crossed_calves_terminate_the_episode(torch_cuda):
    """quadruped.py:237-241 -> task_base.py:137-147: a self-contact that involves a calf ends the episode (terminated, not truncated)."""
    v = vec_env(32)
    v.reset()
    s = v.get_state().cpu().numpy()
    bad = np.arange(32) % 4 == 1
    s[bad] = crossed(s[bad])
    v.set_state(s)
    _, _, done, infos = v.step(np.zeros((32, 6), np.float32))
    assert np.array_equal(done, bad), done
    assert all(not infos[i]["TimeLimit.truncated"] for i in np.nonzero(bad)[0])
    assert np.array_equal(v.get_info("n_invalid").cpu().numpy()[:, 0] > 0, bad)
    v.close()
    w = vec_env(32, self_collision=False)
    w.reset()
    w.set_state(s)
    _, _, done, _ = w.step(np.zeros((32, 6), np.float32))
    assert not done.any()
    w.close()


def test_self_contact_counts_match_the_oracle(torch_cuda):
    from oracle.qso import Oracle
    from scipy.spatial.transform import Rotation as Rot
    n = 64
    v = vec_env(n, action_repeat=1, **RAW)             # one substep per step: the count is that of the given configuration
    o = Oracle(v.cfg)
    o.reset(); v.reset()
    rng = np.random.default_rng(2)
    hits = 0
    for i in range(40):
        s = o.get_state()
        s[:, :3] = [0, 0, 1.0]; s[:, 3:7] = Rot.random(n, random_state=i).as_quat(); s[:, 7:] = 0
        s[:, 13:25] = rng.uniform(np.tile([-1.04, -0.66, -2.72], 4), np.tile([1.04, 2.96, -0.84], 4), size=(n, 12))
        half = np.arange(n) % 2 == 0
        s[half, 13:25] = crossed(s[half])[:, 13:25] + 0.15 * rng.normal(size=(int(half.sum()), 12))
        v.set_state(s.astype(np.float32))
        o.set_state(v.get_state().cpu().numpy().astype(np.float64))       # the same float32 numbers on both sides
        o.step(np.zeros((n, 12), np.float32))
        v.step(np.zeros((n, 12), np.float32))
        no, nv = o.get_info(5)[:, 0], v.get_info("n_invalid").cpu().numpy()[:, 0]
        hits += int((no > 0).sum())
        assert ((nv > 0) != (no > 0)).sum() <= 1, f"config {i}: {no} {nv}"     # float32 / float64 may differ on a grazing contact
    assert hits > 300
    v.close()


def test_sensor_noise_statistics(torch_cuda):
    """The device noise path (Philox + Box-Muller on v_log / v_sin / v_cos; sensor.py:25-32, 46-52): two handles that differ only in
    `noise` run the same actions, so their observations differ by the noise alone.  Per element: mean 0, sigma of go1_config within
    3 %, flags and sigma-0 sensors noise-free, normal distribution, no correlation between elements or steps."""
    from scipy import stats
    n, steps = 8192, 200
    kw = dict(observation_space_mode="PPO_BASIC_CONTACT", env_randomizer_mode="GROUND_RANDOMIZER", seed=7)
    a, b = vec_env(n, noise=True, **kw), vec_env(n, noise=False, **kw)
    oa, ob = a.reset_tensor().clone(), b.reset_tensor().clone()
    std = np.array(a.meta["layout"]["std"])
    assert (std > 0).sum() >= 27 and (std == 0).sum() >= 5
    torch = torch_cuda
    gen = torch.Generator(device=a.device).manual_seed(0)
    diffs = [(oa - ob).cpu().numpy()]
    for t in range(steps):
        act = torch.rand((n, a.action_dim), generator=gen, device=a.device) * 2 - 1
        xa = a.step_tensor(act)[0].clone()
        xb = b.step_tensor(act)[0].clone()
        diffs.append((xa - xb).cpu().numpy())
    assert torch.equal(a.get_state(), b.get_state())          # the noise does not touch the simulation
    d = np.stack(diffs)                                       # [steps + 1, n, obs]
    for k in range(d.shape[2]):
        x = d[:, :, k].ravel()
        if std[k] == 0:
            assert np.all(x == 0), f"element {k} must be noise-free"
            continue
        assert abs(x.std() / std[k] - 1) < 0.03, (k, x.std(), std[k])
        assert abs(x.mean()) < 5 * std[k] / np.sqrt(x.size), (k, x.mean())
        sub = x[:: max(1, x.size // 200000)] / std[k]
        # the observation is rounded to float32 after the noise was added: for |obs| ~ 1 and sigma ~ 2e-4 the quantum is 3e-4 sigma
        assert stats.kstest(sub, "norm").pvalue > 1e-4, (k, stats.kstest(sub, "norm"))
    nz = np.nonzero(std > 0)[0]
    z = d[:, :, nz] / std[nz]
    c_el = np.corrcoef(z.reshape(-1, len(nz)).T)
    assert np.abs(c_el - np.eye(len(nz))).max() < 0.01                                   # between elements
    assert abs(np.mean(z[1:] * z[:-1])) < 0.005                                          # between consecutive steps
    assert abs(np.mean(z[:, 1:] * z[:, :-1])) < 0.005                                    # between neighbouring environments
    a.close(); b.close()


def test_sharded_step_on_one_nccl_rank(torch_cuda):
    """The production branch of ShardedVecEnv.step (qs_step_fused writing into the gathered buffer, in place; RCCL as the backend)
    equals step_tensor bit for bit, into the internal buffer and into a caller's rollout row."""
    import torch.distributed as dist
    from qs_amd.sharded import ShardedVecEnv
    torch = torch_cuda
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 512
        kw = dict(env_randomizer_mode="GROUND_RANDOMIZER", auto_reset=True, seed=3, noise=True)
        a, b = vec_env(n, **kw), vec_env(n, **kw)
        sh = ShardedVecEnv(a, learner_rank=0)
        o1 = sh.reset().clone()
        o2 = b.reset_tensor().clone()
        assert torch.equal(o1, o2)
        roll = torch.zeros((4, n, a.obs_dim + 2), device=a.device)
        gen = torch.Generator(device=a.device).manual_seed(1)
        saw_done = False
        for t in range(150):
            act = torch.rand((n, a.action_dim), generator=gen, device=a.device) * 2 - 1
            if t % 50 > 35:
                act[:] = torch.tensor([0.0, -1.0, 1.0, 0.0, -1.0, 1.0], device=a.device)
            out = roll[t % 4] if t % 2 else None
            obs, rew, done, trunc = sh.step(act, out=out)
            eo, er, ed, et = b.step_tensor(act)
            assert torch.equal(obs, eo) and torch.equal(rew, er) and torch.equal(done, ed.bool()) and torch.equal(trunc, et.bool()), t
            if out is not None:
                assert obs.data_ptr() == out.data_ptr()          # a view of the caller's row: nothing was copied
            saw_done |= bool(done.any())
        assert saw_done
        a.close(); b.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [48, 23])
def test_exact_auto_reset_in_a_partial_wave(torch_cuda, n):
    """reset_lookahead = 0: a finished environment settles inside the step while its wave's other environments are done stepping.  Those
    must come out exactly as in a run without any reset, and the reset ones as a reset of their own."""
    kw = dict(env_randomizer_mode="GROUND_RANDOMIZER", seed=5, settle_steps=400)
    a, b = vec_env(n, auto_reset=True, reset_lookahead=0, **kw), vec_env(n, auto_reset=False, **kw)
    a.reset(); b.reset()
    s = b.get_state().cpu().numpy()
    fall = np.zeros(n, bool); fall[[1, 17, 18, n - 1]] = True
    s[fall, 2] = 0.05; s[fall, 3:7] = [0.7071, 0, 0, 0.7071]            # on their side, below the fallen height: terminate at once
    a.set_state(s); b.set_state(s)
    act = np.random.default_rng(0).uniform(-1, 1, size=(n, 6)).astype(np.float32)
    oa, ra, da, ia = a.step(act)
    ob, rb, db, _ = b.step(act)
    assert np.array_equal(da, db) and np.array_equal(da, fall)
    assert np.array_equal(ra, rb)
    keep = ~fall
    assert np.array_equal(oa[keep], ob[keep])
    assert np.array_equal(a.get_state().cpu().numpy()[keep], b.get_state().cpu().numpy()[keep])
    for i in np.nonzero(fall)[0]:
        assert np.array_equal(ia[i]["terminal_observation"], ob[i])
    # the reset ones: standing again, and a second run of the same thing gives the same bits
    sa = a.get_state().cpu().numpy()
    assert np.all(sa[fall, 2] > 0.2) and np.all(a.get_info("foot_contact").cpu().numpy()[fall] == 1)
    c = vec_env(n, auto_reset=True, reset_lookahead=0, **kw)
    c.reset(); c.set_state(s)
    oc = c.step(act)[0]
    assert np.array_equal(oc, oa) and np.array_equal(c.get_state().cpu().numpy(), sa)
    # and the steps after it agree as well (the environments that were not reset carry nothing over from the settle)
    act2 = np.random.default_rng(1).uniform(-1, 1, size=(n, 6)).astype(np.float32)
    o2a = a.step(act2)[0]; o2b = b.step(act2)[0]
    assert np.array_equal(o2a[keep], o2b[keep])
    a.close(); b.close(); c.close()


def test_lookahead_resets_draw_their_own_parameters(torch_cuda):
    """Every reset is the reference's own: its parameters are the draws of (seed, environment, episode) -- no two resets of a 2000-step run
    at N = 8192 share a draw --, distributed as the reference's randomizers prescribe (env_randomizer.py:67-83, 110-117, 287-289); the
    settle work of the run's resets is done inside the run, and nobody waits for it."""
    from scipy import stats
    torch = torch_cuda
    n, steps = 8192, 2000
    v = vec_env(n, env_randomizer_mode="TEST_RANDOMIZER", auto_reset=True, reset_lookahead=16, seed=11)
    v.reset_tensor()
    gen = torch.Generator(device=v.device).manual_seed(2)
    draws = []
    st0 = v.stats()
    for t in range(steps):
        act = torch.rand((n, v.action_dim), generator=gen, device=v.device) * 2 - 1
        _, _, done, _ = v.step_tensor(act)
        if t % 4 == 0:                                    # sample every fourth step (reading params costs a launch)
            idx = torch.nonzero(done).flatten()
            if len(idx):
                draws.append(v.get_info("params")[idx].cpu().numpy())
    st = v.stats()
    d = np.concatenate(draws)
    assert len(d) > 3000, len(d)
    mu, k, b, m_leg, m_pay = d[:, 0], d[:, 1:4], d[:, 4:7], d[:, 17:20], d[:, 20]
    _, counts = np.unique(d.view(np.dtype((np.void, d.dtype.itemsize * d.shape[1]))), return_counts=True)
    assert int((counts - 1).sum()) == 0                   # every reset its own draw
    # (the first reset of all n environments queued n settles too: they are worked off inside the run)
    ratio = (st["settle_substeps"] - st0["settle_substeps"]) / max(1, (st["resets"] - st0["resets"] + n) * 2500)
    print(f"resets sampled: {len(d)}; settle work done in the run / settle work its resets are worth: {ratio:.2f}; stalls {v.counter('reset_stalls')}")
    assert 0.7 < ratio < 1.3 and v.counter("reset_stalls") == 0
    for name, x, lo, hi in [("mu", mu, 0.5, 1.0), ("k_hip", k[:, 0], 18.0, 22.0), ("k_calf", k[:, 2], 27.0, 33.0), ("b", b[:, 1], 0.27, 0.33),
                            ("m_thigh", m_leg[:, 1], 0.828, 1.012), ("m_payload", m_pay, 0.0, 1.0)]:
        assert x.min() >= lo - 1e-6 and x.max() <= hi + 1e-6, name
        p = stats.kstest((x - lo) / (hi - lo), "uniform").pvalue
        assert p > 1e-3, (name, p)
    v.close()


LAYERS = {
    "plain": dict(),
    "landing_wrapper": dict(wrapper="LANDING"),
    "go_to_rest_wrapper": dict(wrapper="GO_TO_REST", task_env="JUMPING_FORWARD"),
    "cpg": dict(task_env="BACKFLIP", observation_space_mode="PPO_BACKFLIP", action_space_mode="CPG", env_randomizer_mode="TEST_RANDOMIZER"),
    "default_space": dict(action_space_mode="DEFAULT", task_env="JUMPING_FORWARD", observation_space_mode="PPO_BASIC_CONTACT"),
}


@pytest.mark.parametrize("layer", sorted(LAYERS))
def test_info_block_is_optional_and_changes_nothing_else(torch_cuda, layer):
    """info_fields=False: the steps skip the stores of the records' info block (torques, foot forces and flags, pose cache) and move the
    shortest range of the record that the handle's layers need (qs_layout.h: wrapper machine, CPG state only where used).  Observations,
    rewards, done flags and the state must be bit for bit those of the default handle, through falls and look-ahead auto-resets with the
    settle lanes running; the getters of the info block fail loudly, the others keep working."""
    torch = torch_cuda
    n = 200
    kw = dict(env_randomizer_mode="GROUND_RANDOMIZER", auto_reset=True, reset_lookahead=4, seed=4, noise=True)
    kw.update(LAYERS[layer])
    a, b = vec_env(n, info_fields=True, **kw), vec_env(n, info_fields=False, **kw)
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    gen = torch.Generator(device=a.device).manual_seed(3)
    resets = 0
    for t in range(200):
        act = torch.rand((n, a.action_dim), generator=gen, device=a.device) * 2 - 1
        if t % 50 > 35 and a.action_dim == 6:
            act[:] = torch.tensor([0.0, -1.0, 1.0, 0.0, -1.0, 1.0], device=a.device)
        ra, rb = a.step_tensor(act), b.step_tensor(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
        resets += int(ra[2].sum())
    assert resets > (50 if layer == "plain" else 5), resets
    assert torch.equal(a.get_state(), b.get_state())
    ta, tb = a.get_info("task"), b.get_info("task")
    assert torch.equal(ta[:, :32], tb[:, :32]) and torch.equal(ta[:, 41], tb[:, 41]) and torch.equal(ta[:, 43:], tb[:, 43:])
    assert torch.equal(a.get_info("reward_end"), b.get_info("reward_end"))
    assert float(a.get_info("foot_force").sum()) > 0
    for which in ("foot_force", "foot_contact", "torque", "spring_torque"):
        with pytest.raises(RuntimeError, match="info block"):
            b.get_info(which)
    a.close(); b.close()


@pytest.mark.parametrize("model", ["cone", "pyramid"])
def test_soft_payload_parity(torch_cuda, model):
    """payload="soft": the mass randomizer's block as a body of its own on a six-row fixed constraint (quadruped.py:796-819), solved with the
    contacts.  HIP against the float32 oracle, free-running between occasional re-seats, jump episodes with random actions: robot state,
    the block's state, the constraint impulses, the pivot gap, done flags; 20 environments = a full and a partial wave."""
    from oracle.qso import Oracle
    n = 20
    v = vec_env(n, payload="soft", env_randomizer_mode="TEST_RANDOMIZER", friction_model=model, solver_residual_threshold=0.0, seed=5, settle_steps=400)
    o = Oracle(v.cfg, "f32")
    oo, ov = o.reset(), v.reset()
    np.testing.assert_allclose(ov, oo, atol=1e-3)
    b1, b0 = v.get_info("payload_block").cpu().numpy(), o.block()
    np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=1e-5)
    assert b1[:, 19].max() < 1e-4
    rng = np.random.default_rng(0)
    lam = 0.0
    for t in range(60):
        a = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        if t % 30 > 20:
            a[:] = [0, -1, 1, 0, -1, 1]
        if t % 10 == 9:
            st = o.get_state()
            o.set_state(st); v.set_state(st)
        ro, rv = o.step(a), v.step(a)
        so, sv = o.get_state(), v.get_state().cpu().numpy()
        np.testing.assert_allclose(sv[:, :7], so[:, :7], atol=5e-5, err_msg=f"pose step {t}")
        np.testing.assert_allclose(sv[:, 7:13], so[:, 7:13], atol=5e-3, err_msg=f"base velocity step {t}")
        np.testing.assert_allclose(rv[0], ro[0], atol=5e-3, err_msg=f"obs step {t}")
        b0, b1 = o.block(), v.get_info("payload_block").cpu().numpy()
        np.testing.assert_allclose(b1[:, :3], b0["pos"], atol=5e-5, err_msg=f"block position step {t}")
        np.testing.assert_allclose(b1[:, 3:7], b0["quat"], atol=5e-5)
        np.testing.assert_allclose(b1[:, 7:10], b0["v"], atol=5e-3)
        np.testing.assert_allclose(b1[:, 13:19], b0["lam"], atol=2e-4, err_msg=f"constraint impulses step {t}")
        np.testing.assert_allclose(b1[:, 19], b0["gap"], atol=2e-5)
        lam = max(lam, np.abs(b0["lam"]).max())
        np.testing.assert_array_equal(rv[2], ro[2])
        done = ro[2]
        if done.any():
            o.reset(done.astype(np.uint8)); v.reset_tensor(mask=done.astype(np.uint8))
            st = o.get_state()
            o.set_state(st); v.set_state(st)
    assert 1e-3 < lam < 0.1, lam


def test_soft_payload_through_lookahead_auto_resets(torch_cuda):
    """The look-ahead reset states and the settle lanes carry the block with the robot: after hundreds of auto-resets every block
    still sits on its pivot, and the motion stays what the welded model gives (same seed, same actions) to a fraction of a millimetre
    over the first steps."""
    import torch
    n = 256
    kw = dict(env_randomizer_mode="TEST_RANDOMIZER", seed=9, auto_reset=True, reset_lookahead=4, settle_steps=500)
    vs, vw = vec_env(n, payload="soft", **kw), vec_env(n, payload="weld", **kw)
    os_, ow = vs.reset_tensor(), vw.reset_tensor()
    assert (os_ - ow).abs().max() < 2e-3
    g = torch.Generator(device="cuda").manual_seed(3)
    resets = 0
    for t in range(300):
        a = torch.rand((n, 6), generator=g, device="cuda") * 2 - 1
        o1, r1, d1, _ = vs.step_tensor(a)
        if t < 5:
            o2, r2, d2, _ = vw.step_tensor(a)
            assert (o1 - o2).abs().max() < 2e-2, t
        resets += int(d1.sum())
        if t % 50 == 49:
            b = vs.get_info("payload_block")
            assert float(b[:, 19].max()) < 1e-3 and float(b[:, 13:19].abs().max()) < 0.2
            assert bool(torch.isfinite(b).all())
    assert resets > 30, resets
    with pytest.raises(RuntimeError, match="payload_soft"):
        vw.get_info("payload_block")


def test_random_configurations_against_the_oracle(torch_cuda):
    """tools/fuzz_parity.py: random combinations of task, sensor bundle, action space, motor mode, randomizer, wrapper, friction model,
    early exit, springs, filter, time step, info block, payload model and mass rule -- reset and six re-seated steps each against the
    float32 oracle (1014 configurations ran clean when the round closed; here 60 draws, about 40 of them valid)."""
    ran, bad = _fuzz().run(60, 7, verbose=False)
    assert ran >= 25 and not bad, bad[:2]


def _fuzz():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    return fuzz


@pytest.mark.parametrize("step_kernel", ["1", "2"], ids=["k_step", "k_step_dense"])
def test_fuzz_fallen_robots_against_the_oracle(torch_cuda, monkeypatch, step_kernel):
    """tools/fuzz_parity.py's `fallen` mode inside the driver-run gate (it is the mode that caught round 5's miscompiled
    library -- 55 of 500 configurations off -- and it ran from a builder's script only): NO_TASK with the links' contact response on, two
    thirds of the robots thrown onto trunk / hips / thighs / calves in random attitudes with random joint angles under raw torques x 4:
    every kind of row of the many-rows solve (support points, joint stops, the soft payload's six), both friction models, both residual
    thresholds -- reset and six re-seated steps each against the float32 oracle, tolerance + 5 x the oracle's own float64 / float32 spread.
    160 draws with each step kernel forced (QS_STEP_VARIANT is read by qs_create)."""
    monkeypatch.setenv("QS_STEP_VARIANT", step_kernel)
    ran, bad = _fuzz().run(160, 600 + int(step_kernel), verbose=False, fallen=True)
    assert ran >= 150 and not bad, (ran, bad[:2])


@pytest.mark.parametrize("step_kernel", ["1", "2"], ids=["k_step", "k_step_dense"])
def test_fuzz_lookahead_resets_bitwise(torch_cuda, monkeypatch, step_kernel):
    """tools/fuzz_parity.py's `lookahead` mode inside the driver-run gate: random configurations with auto-reset, one handle with K look-ahead
    reset states per environment against one that settles every reset in place, rough actions, every output of every step BITWISE (two
    handles of the SAME library: a kernel that depends on its wave-mates -- what a miscompiled spill does -- shows here; 30 of 232 did in
    round 5).  190 draws (about 150 valid) with each step kernel forced."""
    monkeypatch.setenv("QS_STEP_VARIANT", step_kernel)
    ran, bad = _fuzz().run_lookahead(190, 700 + int(step_kernel), verbose=False)
    assert ran >= 120 and not bad, (ran, bad[:2])


@pytest.mark.parametrize("variant", [dict(), dict(friction_model="pyramid"), dict(payload="soft", env_randomizer_mode="MASS_RANDOMIZER", seed=3),
                                     dict(friction_model="pyramid", solver_residual_threshold=0.0)], ids=["cone", "pyramid", "soft", "pyramid_resid0"])
@pytest.mark.parametrize("neighbour", ["fallen", "joint_limit"])
def test_results_do_not_depend_on_wave_mates(torch_cuda, neighbour, variant):
    """A wave in which some environment needs a rare path (a link on the floor, a joint at its stop) gives up its common-path attempt and
    repeats the env step of all its 16 environments with the full build of the step.  The other 15 must come out of that
    with the bits the common-path build gives them in a wave without such a neighbour: environment k of wave 0 (with the neighbour) against
    its twin k + 16 of wave 1 (same state, same actions, no neighbour), step after step."""
    torch = torch_cuda
    n = 32
    v = vec_env(n, **dict(RAW, **variant))    # NO_TASK: body_contacts "auto" is on, nothing ends the episode
    v.reset()
    if "payload" in variant:                  # twins carry the same payload
        par = v.get_info("params").cpu().numpy(); par[16:] = par[:16]; v.set_params("all", par)
    rng = np.random.default_rng(11)
    s = v.get_state().cpu().numpy()
    s[16:] = s[:16]
    s[:, 13:25] += np.tile(rng.uniform(-0.1, 0.1, size=(16, 12)), (2, 1)).astype(np.float32)
    s[16:] = s[:16]
    odd = 5                                    # the neighbour, wave 0 only
    if neighbour == "fallen":
        s[odd] = fallen_states(s[odd:odd + 1], rng)[0]
        s[odd, 2] = 0.16                      # on its side, just above the floor: it lands within the first steps
    else:
        s[odd, 13 + 2] = -2.76                # FR calf beyond its lower stop (-2.7227): the limit row is there from the first substep
    v.set_state(s)
    rare0 = v.counter("limit_path_substeps")
    twins = np.array([k for k in range(16) if k != odd])
    for t in range(25):
        tau = np.tile(rng.uniform(-4, 4, size=(16, 12)), (2, 1)).astype(np.float32)
        if neighbour == "joint_limit" and t % 5 == 0:      # ... and again every few steps
            s2 = v.get_state().cpu().numpy(); s2[odd, 13 + 2] = -2.76; s2[odd, 25 + 2] = 0.0
            v.set_state(s2)
        obs = v.step(tau)[0]
        st = v.get_state().cpu().numpy()
        assert np.array_equal(st[twins], st[twins + 16]), f"step {t}"
        assert np.array_equal(obs[twins], obs[twins + 16]), f"step {t}"
        assert np.array_equal(v.get_info("foot_force").cpu().numpy()[twins], v.get_info("foot_force").cpu().numpy()[twins + 16])
    assert v.counter("limit_path_substeps") > rare0 + 20       # wave 0 did take the rare path
    v.close()


def test_a_fallen_robot_does_not_depend_on_its_wave_mates_either(torch_cuda):
    """The other direction: the SAME fallen robot in two waves whose other 15 environments stand, fly or lie differently --
    which rows of the many-rows solver are skipped as empty in the whole wave and when the wave leaves the sweeps differ, its bits must not."""
    n = 32
    v = vec_env(n, **RAW)
    v.reset()
    rng = np.random.default_rng(21)
    s = v.get_state().cpu().numpy()
    k0, k1 = 5, 16 + 9                                                       # the same robot, lane position differs too
    lying = fallen_states(s[:1], rng)[0]
    lying[2] = 0.16                                                          # just above the floor: it lands within the first steps
    mates = fallen_states(s[16:], np.random.default_rng(22))                # wave 1: everybody else lies about in other attitudes ...
    s[16:] = mates
    s[20:24, 2] = 0.6                                                        # ... or is in the air
    s[k0] = lying; s[k1] = lying                                             # wave 0: 15 standing robots around it
    v.set_state(s)
    rare0 = v.counter("limit_path_substeps")
    for t in range(40):
        tau = rng.uniform(-4, 4, size=(n, 12)).astype(np.float32)
        tau[k1] = tau[k0]
        obs = v.step(tau)[0]
        st = v.get_state().cpu().numpy()
        assert np.array_equal(st[k0], st[k1]), f"step {t}"
        assert np.array_equal(obs[k0], obs[k1]), f"step {t}"
    ff = v.get_info("foot_force").cpu().numpy()
    assert np.array_equal(ff[k0], ff[k1])
    assert st[k0, 2] < 0.25 and v.counter("limit_path_substeps") > rare0 + 100
    v.close()


def test_lookahead_keeps_up_with_robots_that_fall_all_the_time(torch_cuda):
    """The start of a training run: every robot is thrown down within a few dozen steps, so the resets' settle work is a multiple of the
    stepping work.  The settle lanes size themselves on the device (every cohort takes what the environments' windows lack), so with
    K = 16 states per environment nobody has to settle inside a step -- round 3's first version (lanes trimmed to the idle SIMDs by the
    host) fell to 0.6 M env-steps/s here with hundreds of stalls per step."""
    torch = torch_cuda
    n, steps = 4096, 1200
    v = vec_env(n, env_randomizer_mode="GROUND_RANDOMIZER", auto_reset=True, reset_lookahead=16, seed=3, info_fields=False)
    v.reset_tensor()
    gen = torch.Generator(device=v.device).manual_seed(1)
    acts = torch.rand((64, n, v.action_dim), generator=gen, device=v.device) * 2 - 1
    acts[:, :, 1::3] = -1.0
    acts[0::2, :, 2::3] = 1.0; acts[1::2, :, 2::3] = -0.5
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        v.step_tensor(acts[i % 64])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = v.counter("resets") - n
    print(f"{resets / steps:.1f} resets per step, {v.counter('reset_stalls')} stalls, {n * steps / dt / 1e6:.1f} M env-steps/s, "
          f"settle substeps per env substep {v.counter('settle_substeps') / (n * steps * 10):.2f}")
    assert resets / steps > 60                      # every environment every ~40 steps
    assert v.counter("reset_stalls") == 0
    assert n * steps / dt > 5e6
    v.close()


@pytest.mark.parametrize("mode", ["zero_copy", "copy"])
def test_host_path_terminal_observations_and_views(torch_cuda, mode, monkeypatch):
    """The numpy path (qs_host_step_begin / _end): results arrive in page-locked host memory (written by the kernel itself, or by one D2H
    copy under QS_HOST_PATH=copy); the terminal observations come as a compact list of at most 256 rows with the per-environment array as
    the fallback when more episodes end in one step; copy_outputs=False hands out views of two alternating blocks."""
    torch = torch_cuda
    if mode == "copy":
        monkeypatch.setenv("QS_HOST_PATH", "copy")
    n = 1024
    kw = dict(env_randomizer_mode="GROUND_RANDOMIZER", auto_reset=True, seed=2, noise=True)
    a, b = vec_env(n, copy_outputs=False, **kw), vec_env(n, **kw)      # a: numpy path with views, b: device path
    oa, ob = a.reset(), b.reset_tensor().cpu().numpy()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(4)
    prev = None
    for t in range(40):
        act = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
        if t in (10, 25):                     # everybody falls at once: 1024 episode ends in one step, four times the compact list
            s = b.get_state().cpu().numpy(); s[:, 2] = 0.05; s[:, 3:7] = [0.7071, 0, 0, 0.7071]
            a.set_state(s); b.set_state(s)
        if t == 30:                           # ... and a few: the compact list alone
            s = b.get_state().cpu().numpy(); s[::97, 2] = 0.05; s[::97, 3:7] = [0.7071, 0, 0, 0.7071]
            a.set_state(s); b.set_state(s)
        obs, rew, done, infos = a.step(act)
        ob, rb, db, tb = (x.cpu().numpy() for x in b.step_tensor(torch.as_tensor(act, device=b.device)))
        assert np.array_equal(obs, ob) and np.array_equal(rew, rb) and np.array_equal(done, db.astype(bool))
        term = b.get_info("terminal_obs").cpu().numpy()
        for i in np.flatnonzero(done):
            assert np.array_equal(infos[i]["terminal_observation"], term[i]) and infos[i]["TimeLimit.truncated"] == bool(tb[i])
        assert all(not infos[i] for i in np.flatnonzero(~done))
        if t in (10, 25):
            assert done.all()
        if prev is not None:                  # the views of the step before are still that step's results
            assert np.array_equal(prev[0], prev[1])
        prev = (obs, obs.copy())
    a.close(); b.close()


@pytest.mark.parametrize("cmd", [["examples/rollout.py", "--envs", "256", "--steps", "60"],
                                 ["examples/rollout.py", "--envs", "256", "--steps", "60", "--device-policy", "--wrapper", "LANDING"],
                                 ["examples/imitation.py", "--envs", "128", "--steps", "40"],
                                 ["examples/cpg_gait.py", "--envs", "64", "--seconds", "0.5"]])
def test_examples_run(torch_cuda, cmd):
    """The example programs (the reference's load_model.py loop, the imitation smoke test, the CPG driver) run to the end on the GPU."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, cmd[0])] + cmd[1:], cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_body_contacts_auto_moves_no_done_step_and_no_reward():
    """What INTEGRATION.md / DESIGN.md 4a say body_contacts="auto" (the non-foot links' contact response left off under a task) changes against
    the default (on), in small: two handles, same seed, the e-th episode of an environment from the same reset state under the same actions
    (a function of environment and step of the episode): every episode ends in the same step, the terminal rewards agree, the terminal
    observations of some fall-ended episodes do not (tools/body_contacts_delta.py is the 10^5-episode version of this)."""
    import torch
    from qs_amd.vec_env import QuadrupedVecEnv
    n, ring, steps, emax = 2048, 64, 1300, 48
    kw = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", env_randomizer_mode="GROUND_RANDOMIZER", enable_springs=True,
              enable_action_filter=True, noise=False, seed=1234, auto_reset=True, reset_lookahead=16)
    dev = torch.device("cuda", 0)
    acts = torch.rand((ring, n, 6), generator=torch.Generator(device=dev).manual_seed(5), device=dev) * 2 - 1
    ids = torch.arange(n, device=dev)
    out = []
    for bc in (True, "auto"):
        env = QuadrupedVecEnv(num_envs=n, body_contacts=bc, **kw)
        env.reset_tensor()
        ep_step, ep_idx = torch.zeros(n, dtype=torch.long, device=dev), torch.zeros(n, dtype=torch.long, device=dev)
        length = torch.zeros((n, emax), dtype=torch.int32, device=dev)
        rew_t = torch.zeros((n, emax), device=dev)
        term = torch.zeros((n, emax, env.obs_dim), device=dev)
        trunc_t = torch.zeros((n, emax), dtype=torch.uint8, device=dev)
        for _ in range(steps):
            obs, rew, done, trunc = env.step_tensor(acts[ep_step % ring, ids])
            d = done.bool()
            if bool(d.any()):
                i = d.nonzero().squeeze(1)
                e = ep_idx[i]
                ok = e < emax
                i, e = i[ok], e[ok]
                length[i, e] = (ep_step[i] + 1).to(torch.int32); rew_t[i, e] = rew[i]; trunc_t[i, e] = trunc[i]
                term[i, e] = env.get_info("terminal_obs")[i]
            ep_step = torch.where(d, torch.zeros_like(ep_step), ep_step + 1)
            ep_idx = ep_idx + d.long()
        out.append((ep_idx.cpu().numpy(), length.cpu().numpy(), rew_t.cpu().numpy(), term.cpu().numpy(), trunc_t.cpu().numpy().astype(bool)))
        env.close()
    (na, la, ra, ta, ua), (nb, lb, rb, tb, ub) = out
    both = np.arange(emax)[None, :] < np.minimum(np.minimum(na, nb), emax)[:, None]
    assert both.sum() > 1500 and (~ua[both]).sum() > 500                      # thousands of episodes, hundreds of them ended by a fall
    np.testing.assert_array_equal(la[both], lb[both])                          # the step every episode ends in
    np.testing.assert_array_equal(ua[both], ub[both])
    np.testing.assert_allclose(ra[both], rb[both], atol=1e-3)                  # the terminal step's reward (measured at scale: 5e-5)
    fell = both & ~ua
    delta = np.abs(ta[fell] - tb[fell]).max(axis=1)
    assert (delta > 5e-3).mean() > 0.05                                        # ... and the switch is not a no-op: terminal observations do move
    tl = both & ua
    assert np.array_equal(ta[tl], tb[tl])                                      # episodes that ran to the time limit: bitwise the same
