"""Physics known-answer tests for the oracle's rigid-body step (oracle/qso_phys.c).

PyBullet is absent (parity with the engine is unpinned, see oracle/qso.h); these analytic checks pin the restated
step instead: K1 free fall, K2 momentum conservation, K3 ABA == CRBA + RNEA, K5 energy drift, K6 static stance,
K7 Coulomb cone, K8 joint-limit stop, K9 spring gating continuity, K10 velocity cap, K11 determinism."""
import numpy as np
import pytest

from oracle.qso import Oracle
from qs_amd.config import build_config

TOTAL_MASS = 12.01301


def make(n=1, springs=True, dt=0.001, **kw):
    kw.setdefault("task_env", "JUMPING_IN_PLACE")
    kw.setdefault("observation_space_mode", "PPO_BASIC")
    kw.setdefault("env_randomizer_mode", "NONE")
    # the known answers are those of the fully converged solve: all `solver_iters` sweeps, no early exit (PyBullet's default
    # solverResidualThreshold = 1e-7, this build's default too, stops a few 1e-4 m/s short of it)
    kw.setdefault("solver_residual_threshold", 0.0)
    cfg, meta = build_config(n_envs=n, enable_springs=springs, noise=False, time_step=dt, **kw)
    return Oracle(cfg), cfg


def random_state(o, rng, height=1.0):
    s = o.get_state()
    s[0, 3:7] = rng.normal(size=4)
    s[0, 3:7] /= np.linalg.norm(s[0, 3:7])
    s[0, 7:13] = rng.normal(size=6)
    s[0, 13:25] += 0.3 * rng.normal(size=12)
    s[0, 25:37] = 2 * rng.normal(size=12)
    s[0, 2] = height
    return s


def test_k3_aba_equals_crba_rnea():
    o, _ = make()
    rng = np.random.default_rng(0)
    for _ in range(20):
        o.set_state(random_state(o, rng))
        tau = 5 * rng.normal(size=12)
        acc = o.aba(0, tau)
        H, C = o.crba_rnea(0)
        assert np.abs(H - H.T).max() < 1e-12
        assert np.all(np.linalg.eigvalsh(H) > 0)
        np.testing.assert_allclose(H[3, 3], TOTAL_MASS, rtol=1e-12)
        res = H @ acc + C - np.concatenate([np.zeros(6), tau])
        assert np.abs(res).max() < 1e-10


def test_k1_free_fall():
    o, cfg = make()
    rng = np.random.default_rng(1)
    o.set_state(random_state(o, rng, height=5.0))
    e0 = o.energy(0)
    n = 200
    for _ in range(n):
        o.phys_step(0, np.zeros(12))
    e1 = o.energy(0)
    dpdt = (e1["p"] - e0["p"]) / (n * cfg.dt)
    np.testing.assert_allclose(dpdt, [0, 0, -9.8 * TOTAL_MASS], atol=2e-2)  # O(dt) integrator error
    # com follows the parabola (semi-implicit Euler: z_n = z_0 + v n dt - g dt^2 n(n+1)/2)
    v0 = e0["p"] / TOTAL_MASS
    t = n * cfg.dt
    z_expected = e0["com"][2] + v0[2] * t - 0.5 * 9.8 * cfg.dt ** 2 * n * (n + 1)
    assert abs(e1["com"][2] - z_expected) < 2e-4


def test_k2_zero_gravity_momentum_first_order():
    """With internal torques only, linear and angular momentum are conserved up to the integrator's O(dt) error."""
    drift = []
    for dt, rep in ((1e-3, 1), (1e-4, 10)):
        o, cfg = make(dt=dt)
        rng = np.random.default_rng(2)
        o.set_gravity(0.0)
        o.set_state(random_state(o, rng))
        e0 = o.energy(0)
        for _ in range(100):
            tau = 3 * rng.normal(size=12)
            for _ in range(rep):
                o.phys_step(0, tau)
        e1 = o.energy(0)
        drift.append(max(np.abs(e1["p"] - e0["p"]).max(), np.abs(e1["L"] - e0["L"]).max()))
    assert drift[0] < 1e-2 and drift[1] < drift[0] / 5


def test_k5_energy_drift_passive():
    o, cfg = make(dt=1e-4)
    rng = np.random.default_rng(3)
    o.set_gravity(0.0)
    s = random_state(o, rng)
    s[0, 25:37] *= 0.5
    o.set_state(s)
    e0 = o.energy(0)["KE"]
    for _ in range(2000):
        o.phys_step(0, np.zeros(12))
    e1 = o.energy(0)["KE"]
    assert abs(e1 - e0) / e0 < 2e-2


@pytest.mark.parametrize("springs", [True, False])
def test_k6_static_stance(springs):
    o, cfg = make(springs=springs, enable_action_filter=True)
    o.reset()
    for _ in range(50):  # a little more settling under the same command
        o.step(np.array(cfg.settle_action)[None, :6])
    f = o.get_info(0)[0]
    assert np.all(o.get_info(1)[0] == 1)
    np.testing.assert_allclose(f.sum(), TOTAL_MASS * 9.8, rtol=5e-3)
    np.testing.assert_allclose(f[0], f[1], rtol=2e-2)  # left/right symmetry
    np.testing.assert_allclose(f[2], f[3], rtol=2e-2)
    s = o.get_state()[0]
    assert np.abs(s[7:13]).max() < 2e-2 and np.abs(s[25:]).max() < 0.1
    assert 0.25 < s[2] < 0.34


@pytest.mark.parametrize("model", ["pyramid", "cone"])
def test_k7_coulomb_cone(model):
    """Sliding feet: the horizontal impulse of every substep equals mu * (sum of normal impulses) exactly (with the implicit cone it
    is the NORM of the impulse that does: its direction follows the contact inertia, see k7b)."""
    mu = 0.5
    o, cfg = make(friction_model=model)
    cfg.randomizer_flags = 8
    o = Oracle(cfg)
    o.set_params(0, np.array([mu]))
    o.reset()
    s = o.get_state()
    s[0, 7] = 3.0  # whole robot translating along +x at 3 m/s: all four feet slide
    o.set_state(s)
    tau_hold = o.get_info(2)[0] + o.get_info(3)[0]
    for _ in range(5):
        p0 = o.energy(0)["p"][:2].copy()
        o.phys_step(0, tau_hold)
        dpv = o.energy(0)["p"][:2] - p0
        dp = dpv[0] if model == "pyramid" else -np.linalg.norm(dpv)
        fn = o.get_info(0)[0].sum()
        assert fn > 50 and dpv[0] < 0
        # cone: each foot's impulse has norm mu x its normal impulse, but the four directions differ by a few degrees (they follow
        # each contact's inertia), so the norm of their sum falls short of the sum of their norms by ~0.2 %
        assert dp == pytest.approx(-mu * fn * cfg.dt, rel=2e-3 if model == "pyramid" else 4e-3)
    # and a foot at rest on the ground is not dragged: |f_t| stays inside the cone (robot keeps standing still)
    o.reset()
    for _ in range(100):
        o.phys_step(0, tau_hold)
    assert abs(o.get_state()[0, 7]) < 5e-2


@pytest.mark.parametrize("model,expect", [("pyramid", np.sqrt(2.0)), ("cone", 1.0)])
def test_k7b_friction_model_on_a_diagonal_slide(model, expect):
    """All four feet sliding along the diagonal x = y: the pyramid clamps each tangent direction at mu N on its own, so the friction
    force is sqrt(2) mu N; the implicit cone (PyBullet's enableConeFriction) projects the pair onto the disc: mu N, opposite to
    the sliding direction."""
    mu = 0.5
    cfg, _ = build_config(n_envs=1, task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", enable_springs=True, enable_action_filter=True,
                          env_randomizer_mode="NONE", friction_model=model)
    cfg.randomizer_flags = 8
    o = Oracle(cfg)
    o.set_params(0, np.array([mu]))
    o.reset()
    s = o.get_state()
    s[0, 7], s[0, 8] = 2.0, 2.0
    o.set_state(s)
    tau_hold = o.get_info(2)[0] + o.get_info(3)[0]
    for _ in range(5):
        p0 = o.energy(0)["p"][:2].copy()
        o.phys_step(0, tau_hold)
        dp = o.energy(0)["p"][:2] - p0
        fn = o.get_info(0)[0].sum()
        assert fn > 50
        assert np.linalg.norm(dp) == pytest.approx(expect * mu * fn * cfg.dt, rel=3e-3)
        d = dp / np.linalg.norm(dp)
        # against the motion; the cone projects the IMPULSE that would stop the feet, whose direction follows the (anisotropic) contact
        # inertia, so it is only roughly opposite to the sliding velocity
        assert d[0] < -0.5 and d[1] < -0.5, d


def test_k8_joint_limit_stop():
    o, cfg = make()
    o.set_gravity(0.0)
    s = o.get_state()
    s[0, 2] = 2.0
    o.set_state(s)
    tau = np.zeros(12)
    tau[2] = -33.55  # drive the FR calf into its lower limit
    for _ in range(400):
        o.phys_step(0, tau)
    q = o.get_state()[0, 15]
    assert q > -2.72271363311 - 0.02 and q < -2.6


def test_k9_spring_gating_continuity():
    o, cfg = make()
    k, b, rest = np.array(cfg.spring_k, float), np.array(cfg.spring_b, float), np.array(cfg.spring_rest, float)
    q0 = np.tile(rest, 4)
    for j in range(12):
        for eps in (-1e-9, 1e-9):
            q = q0.copy()
            q[j] += eps
            t = o.spring_torque(k, b, rest, q, np.zeros(12))
            assert abs(t[j]) < 1e-7  # torque is continuous through the rest angle
    # static deflection under a known load: k * dq = tau
    q = q0.copy()
    q[1] += 0.1   # thigh above rest -> spring engaged
    q[2] -= 0.2   # calf below rest -> engaged
    t = o.spring_torque(k, b, rest, q, np.zeros(12))
    np.testing.assert_allclose([t[1], t[2]], [-k[1] * 0.1, k[2] * 0.2], rtol=1e-6)


def test_k10_velocity_cap():
    o, cfg = make()
    o.set_gravity(0.0)
    s = o.get_state()
    s[0, 2] = 2.0
    s[0, 25:] = 100.0
    o.set_state(s)
    o.phys_step(0, np.zeros(12))
    assert np.abs(o.get_state()[0, 25:]).max() <= cfg.vel_cap + 1e-6


def test_k11_determinism_across_batch_size():
    rng = np.random.default_rng(7)
    acts = rng.uniform(-1, 1, size=(30, 4, 6)).astype(np.float32)
    o4, _ = make(n=4, env_randomizer_mode="GROUND_RANDOMIZER", seed=3)
    o4.reset()
    out4 = [o4.step(a)[0].copy() for a in acts]
    o2, _ = make(n=2, env_randomizer_mode="GROUND_RANDOMIZER", seed=3)
    o2.reset()
    out2 = [o2.step(a[:2])[0].copy() for a in acts]
    for a, b in zip(out4, out2):
        assert np.array_equal(a[:2], b)


def test_f32_build_tracks_f64():
    rng = np.random.default_rng(11)
    cfg, _ = build_config(n_envs=2, enable_springs=True, noise=False, task_env="JUMPING_IN_PLACE",
                          observation_space_mode="PPO_BASIC", enable_action_filter=True, env_randomizer_mode="NONE")
    a, b = Oracle(cfg, "f64"), Oracle(cfg, "f32")
    a.reset(); b.reset()
    np.testing.assert_allclose(a.get_state(), b.get_state(), atol=2e-3)
    for _ in range(5):
        act = rng.uniform(-1, 1, size=(2, 6))
        b.set_state(a.get_state())
        oa, ob = a.step(act)[0], b.step(act)[0]
        np.testing.assert_allclose(oa, ob, atol=5e-3, rtol=1e-3)


def _normal_mode_period(step, get_state, set_state, H, s0, k, mode_index, dt, n_periods=6, eps=2e-3):
    """Excite ONE small-oscillation mode of the robot whose joints are held by torsional springs of stiffness k about the pose s0 (zero
    gravity, floating base) and measure its period from the zero crossings of the mode's largest joint."""
    import scipy.linalg
    K = np.diag([0.0] * 6 + [k] * 12)
    w2, V = scipy.linalg.eigh(K, H)                 # K v = w^2 H v ; six rigid-body modes at w = 0, twelve elastic ones
    assert np.abs(w2[:6]).max() < 1e-6 * w2[6] and w2[6] > 0
    w, v = np.sqrt(w2[mode_index]), V[:, mode_index]
    v = v / np.abs(v[6:]).max()
    s = s0.copy()
    q0 = s0[0, 13:25].copy()
    s[0, 10:13] = eps * w * v[0:3]                  # H's coordinates: base angular 3, base linear 3 (H[3,3] = total mass), joints 12;
    s[0, 7:10] = eps * w * v[3:6]                   # identity orientation: base frame = world frame
    s[0, 25:37] = eps * w * v[6:]
    set_state(s)
    j = int(np.abs(v[6:]).argmax())
    n = int(np.ceil(n_periods * 2 * np.pi / w / dt)) + 5
    x = np.zeros(n + 1)
    for i in range(n):
        q = get_state()[0, 13:25].astype(np.float64)
        step(-k * (q - q0))
        x[i + 1] = get_state()[0, 13 + j] - q0[j]
    up = [i + x[i] / (x[i] - x[i + 1]) for i in range(1, n) if x[i] < 0 <= x[i + 1]]      # upward zero crossings, linear interpolation
    assert len(up) >= n_periods - 1
    T = (up[-1] - up[0]) / (len(up) - 1) * dt
    amp = np.abs(x).max()
    return T, w, amp, eps * np.abs(v[6 + j])


@pytest.mark.parametrize("which", ["oracle_f64", "kernel_arithmetic_f32"])
def test_k4_small_oscillation_periods_match_the_analytic_modes(which):
    """K4 of SURVEY 8c ("single-leg pendulum period vs analytic small-angle").  A floating base in free fall feels no gravity, so the
    restoring force of the known answer is a torsional spring in every joint (zero gravity, tau = -k (q - q0), no damping) instead of
    gravity; what sets the period is the same: the inertia the legs present to their joints, here with the coupling through the floating
    base.  Analytic answer: the generalised eigenproblem K v = w^2 H v with H the 18 x 18 mass matrix at q0 (tests/test_urdf_model.py
    holds the oracle's H to first principles, 2e-6).  Semi-implicit Euler is symplectic: a linear mode keeps its amplitude and runs at
    the discrete frequency sin(w_d dt / 2) = w dt / 2.  The slowest and the fastest elastic mode are excited one at a time on the oracle
    (float64) and on the kernels' own arithmetic (host lane emulation, float32); measured period against the analytic one."""
    o, cfg = make(springs=False)
    o.set_gravity(0.0)
    o.reset()
    s0 = o.get_state()
    s0[0, :3] = [0.0, 0.0, 1.0]
    s0[0, 3:7] = [0, 0, 0, 1]
    s0[0, 7:13] = 0
    s0[0, 25:] = 0
    o.set_state(s0)
    H, _ = o.crba_rnea(0)
    k, dt = 40.0, cfg.dt
    if which == "oracle_f64":
        step, get, put, tol = (lambda tau: o.phys_step(0, tau)), o.get_state, o.set_state, 5e-5
    else:
        from emu.emu import Emu
        cfg.gravity = 0.0
        e = Emu(cfg)
        e.reset()
        step, get, put, tol = (lambda tau: e.phys_step(0, tau)), e.get_state, (lambda s: e.set_state(s.astype(np.float32))), 2e-4
    for mode in (6, 17):
        T, w, amp, amp0 = _normal_mode_period(step, get, put, H, s0, k, mode, dt)
        w_d = 2.0 / dt * np.arcsin(w * dt / 2.0)
        assert abs(T - 2 * np.pi / w_d) < tol * T, (mode, T, 2 * np.pi / w_d)
        assert abs(amp / amp0 - 1.0) < 2e-2, (mode, amp, amp0)        # one pure mode: the joint swings at the amplitude it was given
    assert 2 * np.pi / w < 0.2                                         # (the fastest mode: tens of hertz, ~60 steps per period)
