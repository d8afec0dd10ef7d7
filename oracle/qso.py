"""ctypes loader for the CPU oracle (oracle/qso.h).  TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package."""
import ctypes as C
import os as _os

# the C oracle spreads environments over OpenMP threads; one thread unless the caller (bench.py's cpu_baseline) asks for more
_os.environ.setdefault("OMP_NUM_THREADS", "1")
_os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    targets = [os.path.join(_HERE, n) for n in ("libqso_f64.so", "libqso_f32.so")]
    stale = force or not all(os.path.exists(t) for t in targets)
    if not stale:   # (the sources may be newer than libraries that travelled with the tree)
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
        stale = max(os.path.getmtime(f) for f in srcs) > min(os.path.getmtime(t) for t in targets)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return targets


class Oracle:
    """Batched env on the CPU oracle.  `cfg` is a qs_amd.config.QsConfig (same layout as qso_config)."""

    def __init__(self, cfg, precision="f64"):
        build()
        self.real = np.float64 if precision == "f64" else np.float32
        self._creal = C.c_double if precision == "f64" else C.c_float
        self.lib = C.CDLL(os.path.join(_HERE, f"libqso_{precision}.so"))
        self.lib.qso_last_error.restype = C.c_char_p
        self.lib.qso_pitch_backflip.restype = self._creal
        self.lib.qso_u01.restype = C.c_float
        self.cfg = cfg
        self.n, self.d, self.o = cfg.n_envs, cfg.action_dim, cfg.obs_dim
        self.h = C.c_void_p()
        self._check(self.lib.qso_create(C.byref(cfg), C.byref(self.h)))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.qso_last_error().decode())

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def close(self):
        if self.h:
            self.lib.qso_destroy(self.h)
            self.h = None

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self.lib.qso_reset(self.h, None if m is None else self._p(m)))
        return self.get_obs()

    def get_obs(self):
        obs = np.zeros((self.n, self.o), np.float32)
        self._check(self.lib.qso_get_obs(self.h, self._p(obs)))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n, self.d)
        obs = np.zeros((self.n, self.o), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        self._check(self.lib.qso_step(self.h, self._p(a), self._p(obs), self._p(rew), self._p(done), self._p(trunc)))
        return obs, rew, done.astype(bool), trunc.astype(bool)

    def rollout(self, ring, steps):
        """bench.py's cpu_baseline: `steps` env steps with the action ring [n_ring, n, d], threads not meeting between steps; -> episodes ended."""
        ring = np.ascontiguousarray(ring, np.float32).reshape(-1, self.n, self.d)
        resets = C.c_ulonglong(0)
        self._check(self.lib.qso_rollout(self.h, self._p(ring), int(ring.shape[0]), int(steps), C.byref(resets)))
        return int(resets.value)

    def get_state(self):
        s = np.zeros((self.n, 37), self.real)
        self._check(self.lib.qso_get_state(self.h, self._p(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, self.real).reshape(self.n, 37)
        self._check(self.lib.qso_set_state(self.h, self._p(s)))

    def set_warm(self, w):
        w = np.ascontiguousarray(w, self.real).reshape(self.n, 4)
        self._check(self.lib.qso_set_warm(self.h, self._p(w)))

    def snapshot(self):
        """every environment's whole record (state, parameters, task / filter / wrapper state, counters, outputs) as bytes: restore() puts
        it back, so that a test can ask "what would the step give from THIS state" without disturbing the run (tests/yardstick.py)"""
        self.lib.qso_snapshot_size.restype = C.c_size_t
        buf = C.create_string_buffer(self.lib.qso_snapshot_size(self.h))
        self._check(self.lib.qso_snapshot(self.h, buf))
        return buf

    def restore(self, buf):
        self._check(self.lib.qso_restore(self.h, buf))

    _INFO_DIM = {0: 4, 1: 4, 2: 12, 3: 12, 4: 48, 5: 1, 6: 24, 7: 4, 8: 12, 10: 4}

    def get_info(self, which):
        dim = self.o if which == 9 else self._INFO_DIM[which]
        out = np.zeros((self.n, dim), self.real)
        self._check(self.lib.qso_get_info(self.h, which, self._p(out)))
        return out

    def reset_to(self, states, mask=None):
        st = np.ascontiguousarray(states, self.real).reshape(self.n, 37)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self.lib.qso_reset_to(self.h, None if m is None else self._p(m), self._p(st)))
        obs = np.zeros((self.n, self.o), np.float32)
        self._check(self.lib.qso_get_obs(self.h, self._p(obs)))
        return obs

    def set_demo(self, rows):
        r = np.ascontiguousarray(rows, np.float32).reshape(-1, self.d + 38)
        self._check(self.lib.qso_set_demo(self.h, self._p(r), int(r.shape[0])))

    def set_demo_counter(self, values, mask=None):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, np.int32), (self.n,)))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self.lib.qso_set_demo_counter(self.h, None if m is None else self._p(m), self._p(v)))

    def set_threads(self, n):
        """Spread the environments over n OpenMP threads (default 1)."""
        self._check(self.lib.qso_set_threads(int(n)))

    def set_trace(self, env):
        """Per-substep rows of one environment (layout of qs_set_trace); returns the array the next step() calls fill."""
        self._trace = np.zeros((self.cfg.action_repeat, 70), self.real)
        self._check(self.lib.qso_set_trace(self.h, int(env), self._p(self._trace)))
        return self._trace

    def set_params(self, which, vals):
        v = np.ascontiguousarray(vals, self.real)
        self._check(self.lib.qso_set_params(self.h, which, self._p(v)))

    def set_task(self, task):
        t = np.ascontiguousarray(task, self.real).reshape(self.n, 48)
        self._check(self.lib.qso_set_task(self.h, self._p(t)))

    def eval_reward(self, which):
        out = np.zeros(self.n, self.real)
        self._check(self.lib.qso_eval_reward(self.h, which, self._p(out)))
        return out

    # ---- physics-only entry points
    def phys_step(self, env, tau):
        t = np.ascontiguousarray(tau, self.real)
        self._check(self.lib.qso_phys_step(self.h, env, self._p(t)))

    def contacts(self, env=0, max_n=64):
        """[(bodyA, bodyB, linkA, linkB, distance, normal force)] of the last substep, PyBullet numbering (qso.h)."""
        ids = np.zeros((max_n, 4), np.int32)
        df = np.zeros((max_n, 2), self.real)
        n = self.lib.qso_get_contacts(self.h, int(env), self._p(ids), self._p(df), max_n)
        return [(int(a), int(b), int(c), int(d), float(x), float(f)) for (a, b, c, d), (x, f) in zip(ids[:min(n, max_n)], df[:min(n, max_n)])]

    def boxes_overlap(self, ca, Ra, ha, cb, Rb, hb):
        args = [np.ascontiguousarray(x, self.real) for x in (ca, Ra, ha, cb, Rb, hb)]
        return bool(self.lib.qso_geom_boxes_overlap(*[self._p(x) for x in args]))

    def block(self):
        """payload block as its own body (payload="soft"): dict of pos, quat, v, w, lam [N, .] and gap [N]."""
        out = np.zeros((self.n, 20), self.real)
        self._check(self.lib.qso_get_block(self.h, self._p(out)))
        return dict(pos=out[:, :3], quat=out[:, 3:7], v=out[:, 7:10], w=out[:, 10:13], lam=out[:, 13:19], gap=out[:, 19])

    def set_gravity(self, g):
        self.lib.qso_phys_set_gravity(self.h, self._creal(g))

    def set_manifold(self, mode):
        """0: foot + two support points per leg, three with the foot off the ground (default, what the kernels build); 1: up to four points per
        collision primitive (experiment); 2: mode 0 with warm-started support points; 3: two support points whatever the foot does (rounds 2 - 5)."""
        self._check(self.lib.qso_phys_set_manifold(self.h, int(mode)))

    def crba_rnea(self, env):
        H = np.zeros((18, 18), self.real)
        Cb = np.zeros(18, self.real)
        self._check(self.lib.qso_phys_crba_rnea(self.h, env, self._p(H), self._p(Cb)))
        return H, Cb

    def aba(self, env, tau):
        t = np.ascontiguousarray(tau, self.real)
        acc = np.zeros(18, self.real)
        self._check(self.lib.qso_phys_aba(self.h, env, self._p(t), self._p(acc)))
        return acc

    def energy(self, env):
        out = np.zeros(11, self.real)
        self._check(self.lib.qso_phys_energy(self.h, env, self._p(out)))
        return dict(KE=out[0], PE=out[1], p=out[2:5].copy(), L=out[5:8].copy(), com=out[8:11].copy())

    # ---- stateless helpers
    def action_to_command(self, action):
        a = np.ascontiguousarray(action, self.real)
        out = np.zeros(12, self.real)
        self.lib.qso_action_to_command(C.byref(self.cfg), self._p(a), self._p(out))
        return out

    def command_to_action(self, cmd):
        c = np.ascontiguousarray(cmd, self.real)
        out = np.zeros(self.d, self.real)
        self.lib.qso_command_to_action(C.byref(self.cfg), self._p(c), self._p(out))
        return out

    def pd_torque(self, kp3, kd3, cmd, q, qd):
        args = [np.ascontiguousarray(x, self.real) for x in (kp3, kd3, cmd, q, qd)]
        out = np.zeros(12, self.real)
        self.lib.qso_pd_torque(C.byref(self.cfg), *[self._p(x) for x in args], self._p(out))
        return out

    def spring_torque(self, k3, b3, rest3, q, qd):
        args = [np.ascontiguousarray(x, self.real) for x in (k3, b3, rest3, q, qd)]
        out = np.zeros(12, self.real)
        self.lib.qso_spring_torque(*[self._p(x) for x in args], self._p(out))
        return out

    def leg_fk_jac(self, leg, q3):
        q = np.ascontiguousarray(q3, self.real)
        J = np.zeros((3, 3), self.real)
        p = np.zeros(3, self.real)
        self.lib.qso_leg_fk_jac(self.cfg.leg_len, leg, self._p(q), self._p(J), self._p(p))
        return J, p

    def leg_ik(self, leg, xyz):
        x = np.ascontiguousarray(xyz, self.real)
        q = np.zeros(3, self.real)
        self.lib.qso_leg_ik(self.cfg.leg_len, leg, self._p(x), self._p(q))
        return q

    def filter_step(self, x, xh, yh):
        """xh, yh: [2, d] arrays (row 0 newest), updated in place."""
        x = np.ascontiguousarray(x, self.real)
        y = np.zeros_like(x)
        self.lib.qso_filter_step(self.cfg.filt_b, self.cfg.filt_a, len(x), self._p(x), self._p(xh), self._p(yh), self._p(y))
        return y

    def quat_to_rpy(self, q):
        q = np.ascontiguousarray(q, self.real)
        out = np.zeros(3, self.real)
        self.lib.qso_quat_to_rpy(self._p(q), self._p(out))
        return out

    def pitch_backflip(self, q, switched):
        q = np.ascontiguousarray(q, self.real)
        return float(self.lib.qso_pitch_backflip(self._p(q), int(switched)))

    def cpg_update(self, p5, dt, X8):
        """one HopfNetwork.update(): X8 (r[4], theta[4]) is advanced in place; returns (x[4], z[4])"""
        p = np.ascontiguousarray(p5, self.real)
        x, z = np.zeros(4, self.real), np.zeros(4, self.real)
        self.lib.qso_cpg_update(C.byref(self.cfg), self._p(p), self._creal(dt), self._p(X8), self._p(x), self._p(z))
        return x, z

    def philox(self, seed, env, stream, ctr, blk):
        out = (C.c_uint32 * 4)()
        self.lib.qso_philox(C.c_uint64(seed), C.c_uint32(env), C.c_uint32(stream), C.c_uint32(ctr), C.c_uint32(blk), out)
        return list(out)
