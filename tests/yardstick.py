"""The parity yardstick at a discontinuity of the step map (TEST INFRASTRUCTURE; shared by the GPU parity tests, their CPU twins on the host
lane emulation and tools/fuzz_parity.py).

One env.step from an identical state is held to fixed tolerances (float32 kernel against float64 oracle: pose 5e-6, base velocity 5e-4, q
2e-5 rad, qd 5e-3 rad/s) wherever the step map is smooth.  Where it is not -- a foot touching down inside the step, a trunk / thigh / calf
hitting the floor, a joint running into its stop -- rounding decides on which side of the discontinuity a trajectory falls and no fixed
tolerance means anything: too tight fails on the oracle's own float32 build, too loose (round 5: 0.5 m/s, 2 rad/s) passes a miscompiled
library.  What does mean something is the oracle's OWN sensitivity at that state:

  1. |oracle float32 - oracle float64| from the same state and action  (same algorithm, other rounding),
  2. |oracle float64 from a state 1e-6 away - oracle float64|          (same rounding, other side of whatever switches), 12 draws,

per environment and per GROUP of like quantities (positions with positions, joint rates with joint rates: a joint rate that jumps by 3 rad/s
at an impact must not widen the bound of a position, which moves by dt x that).  A device value may sit `tol + 5 x spread` from the float64
oracle.  The second yardstick is only evaluated where the first one does not cover the device (it costs twelve oracle steps)."""
import numpy as np

TOL_Q, TOL_QD, TOL_BASE_V, TOL_POS = 2e-5, 5e-3, 5e-4, 5e-6
FACTOR = 5.0
STATE_GROUPS = (("pose", 0, 7, TOL_POS), ("base_velocity", 7, 13, TOL_BASE_V), ("q", 13, 25, TOL_Q), ("qd", 25, 37, TOL_QD))
# per-sensor floors for observations: what the strict comparison allows the quantity the sensor reads (sensor keys of qs_amd.config.SENSORS)
SENSOR_TOL = {"Encoder": TOL_Q * 5, "JointVelocity": TOL_QD, "Pitch": 1e-4, "Height": 2e-5, "Base Linear Velocity z direction": TOL_BASE_V,
              "Pitch rate": TOL_QD, "Base Height Velocity X": TOL_BASE_V, "Base Linear Velocity": TOL_BASE_V, "Base Angular Velocity": TOL_QD,
              "FeetPosition": 1e-4, "FeetVelocity": TOL_QD, "Pitch-BackFlip": 1e-4, "Orientation Roll Pitch Yaw": 1e-4, "Quaternion": 1e-4,
              "is landing": 0.0, "is jumping": 0.0, "BoolContatc": 0.0}


def obs_groups(layout):
    """[(sensor key, first column, end column, floor)] of an observation bundle (meta["layout"] of qs_amd.config.build_config)."""
    out, a = [], 0
    for key, dim in zip(layout["keys"], layout["dims"]):
        out.append((key, a, a + dim, SENSOR_TOL.get(key, TOL_QD)))
        a += dim
    return out


def group_spread(x, y, groups):
    """[n, G]: per row, the largest |x - y| inside each group's columns."""
    d = np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64))
    return np.stack([d[:, a:b].max(axis=1) for _, a, b, _ in groups], axis=1)


def bound(spread, groups, width, factor=FACTOR):
    """[n, width]: tol_g + factor x spread_g on the columns of group g."""
    out = np.zeros((spread.shape[0], width))
    for g, (_, a, b, tol) in enumerate(groups):
        out[:, a:b] = tol + factor * spread[:, g:g + 1]
    return out


def perturbed_states(s, rng, scale=1e-6):
    sp = s + scale * rng.standard_normal(s.shape) * np.maximum(np.abs(s), 1.0)
    sp[:, 3:7] /= np.linalg.norm(sp[:, 3:7], axis=1, keepdims=True)
    return sp


class Yardstick:
    """Outputs of one step as a dict of [n, k] arrays (collect), their grouped spread between two evaluations (spread), and what leaves the bound
    `tolerance + 5 x spread` (excess).  `fields`: name -> (groups, width); state, observation and reward are always there, `extra` adds
    single-group fields: name -> (width, tolerance)."""

    def __init__(self, layout, extra=None):
        self.og = obs_groups(layout)
        self.width_obs = self.og[-1][2]
        self.fields = {"state": (STATE_GROUPS, 37), "obs": (self.og, self.width_obs), "reward": ((("reward", 0, 1, 2e-4),), 1)}
        for name, (k, tol) in (extra or {}).items():
            self.fields[name] = (((name, 0, k, tol),), k)

    def collect(self, o, res, extra=None):
        """res = (obs, reward, done, truncated) of o.step(); extra: name -> array"""
        out = {"state": np.asarray(o.get_state(), np.float64), "obs": np.asarray(res[0], np.float64), "reward": np.asarray(res[1], np.float64)[:, None]}
        for k, v in (extra or {}).items():
            out[k] = np.asarray(v, np.float64).reshape(out["state"].shape[0], -1)
        return out

    def spread(self, a, b):
        return {k: group_spread(a[k], b[k], self.fields[k][0]) for k in self.fields if k in a and k in b}

    @staticmethod
    def wider(s1, s2):
        return {k: np.maximum(s1[k], s2[k]) for k in s1}

    def bounds(self, spread, factor=FACTOR):
        return {k: bound(spread[k], self.fields[k][0], self.fields[k][1], factor) for k in spread}

    def excess(self, dev, ref, spread, rows, rtol=None):
        """{field: (worst |dev - ref| / bound, environment, column)} of the fields in which some row of `rows` leaves its bound"""
        bad = {}
        b = self.bounds(spread)
        for k in b:
            if k not in dev:
                continue
            d = np.abs(np.asarray(dev[k], np.float64).reshape(ref[k].shape) - ref[k])
            lim = b[k] + (rtol or {}).get(k, 0.0) * np.abs(ref[k])
            r = d[rows] / np.maximum(lim[rows], 1e-300)
            if r.size and r.max() > 1.0:
                i, j = np.unravel_index(int(np.argmax(r)), r.shape)
                bad[k] = (float(r.max()), int(np.flatnonzero(rows)[i]) if rows.dtype == bool else int(i), int(j), float(d[rows][i, j]), float(lim[rows][i, j]))
        return bad


def oracle_extra(o):
    """the oracle-side values of the optional fields: PD torque, foot forces, the end-of-episode bonus the task would add now"""
    return dict(torque=o.get_info(2), foot_force=o.get_info(0), reward_end=o.eval_reward(1))


EXTRA_FIELDS = dict(torque=(12, 5e-3), foot_force=(4, 0.5), reward_end=(1, 2e-4))
EXTRA_RTOL = dict(foot_force=2e-2, reward_end=1e-3, reward=1e-3)


def what_if(o64, snap, states, action, ys, base, base_done, trials, rng, warm=None, extra=True):
    """Second yardstick: the float64 oracle's step from `trials` states 1e-6 away from `states` (the run's own step has been taken: `base` are
    its outputs, `base_done` its done flags; `snap` is the oracle's snapshot from just before it).  Leaves the oracle as it found it.
    -> (spread dict, [n] bool: some trial ended / did not end the episode where the run's step did the opposite)"""
    after = o64.snapshot()
    spread, flips = None, np.zeros(len(base_done), bool)
    for _ in range(trials):
        o64.restore(snap)
        o64.set_state(perturbed_states(states, rng))
        if warm is not None:
            o64.set_warm(warm)
        res = o64.step(action)
        s = ys.spread(ys.collect(o64, res, oracle_extra(o64) if extra else None), base)
        spread = s if spread is None else ys.wider(spread, s)
        flips |= res[2] != base_done
    o64.restore(after)
    return spread, flips


def lowest_point(s):
    """[n]: height of the robot's lowest surface point above the plane, conservatively, for states [n, 37].  Geometry of go1.urdf
    (tests/golden/urdf_tables.npz: hip joints at (+-0.1881, +-0.04675, 0), thigh joints 0.08 outwards, links 0.213 long, trunk box half
    extents 0.1881 x 0.04675 x 0.057): trunk box corners; hip cylinders (radius 0.046), thigh ends, knees (boxes up to 0.0245 thick) and
    feet (radius 0.02) as spheres."""
    from scipy.spatial.transform import Rotation as Rot
    n = len(s)
    R = Rot.from_quat(s[:, 3:7]).as_matrix()
    pts, margin = [], []
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                pts.append(np.tile([sx * 0.1881, sy * 0.04675, sz * 0.057], (n, 1))); margin.append(0.0)
    for L, (sx, sy) in enumerate(((1, -1), (1, 1), (-1, -1), (-1, 1))):
        q1, q2, q3 = s[:, 13 + 3 * L], s[:, 14 + 3 * L], s[:, 15 + 3 * L]
        hip = np.tile([sx * 0.1881, sy * 0.04675, 0.0], (n, 1))
        c1, s1 = np.cos(q1), np.sin(q1)
        rx = lambda v: np.stack([v[:, 0], c1 * v[:, 1] - s1 * v[:, 2], s1 * v[:, 1] + c1 * v[:, 2]], 1)       # rotation about x by q1
        ry = lambda v, ang: np.stack([np.cos(ang) * v[:, 0] + np.sin(ang) * v[:, 2], v[:, 1], -np.sin(ang) * v[:, 0] + np.cos(ang) * v[:, 2]], 1)
        down = np.tile([0.0, 0.0, -0.213], (n, 1))
        thigh = hip + rx(np.tile([0.0, sy * 0.08, 0.0], (n, 1)))
        knee = thigh + rx(ry(down, q2))
        foot = knee + rx(ry(down, q2 + q3))
        pts += [hip, thigh, knee, foot]; margin += [0.046, 0.025, 0.025, 0.02]
    z = np.stack([s[:, 2] + np.einsum("nj,nj->n", R[:, 2, :], p) - m for p, m in zip(pts, margin)], 1)
    return z.min(axis=1)


def thrown_states(s, rows, rng):
    """`rows` of the states `s` replaced by a robot in a random attitude falling at 0.3 - 1.5 m/s, its lowest point up to 8 mm above the
    ground (no penetration to be pushed out of): within the env step a trunk corner, hip, thigh, knee or foot hits the floor (with a non-foot
    link: the many-rows solve; under a task the episode's last step)"""
    from scipy.spatial.transform import Rotation as Rot
    k = int(rows.sum())
    s = s.copy()
    s[rows, 3:7] = Rot.from_euler("xyz", np.stack([rng.uniform(-1.4, 1.4, k), rng.uniform(-1.2, 1.2, k), rng.uniform(-3.1, 3.1, k)], 1)).as_quat()
    s[rows, 7:10] = rng.normal(size=(k, 3)) * 0.3
    s[rows, 9] -= rng.uniform(0.3, 1.5, k)
    s[rows, 10:13] = rng.normal(size=(k, 3)) * 1.0
    s[rows, 13:25] += rng.uniform(-0.3, 0.3, size=(k, 12))
    s[rows, 25:37] = rng.normal(size=(k, 12)) * 1.0
    s[rows, 2] += rng.uniform(0.0, 0.008, k) - lowest_point(s[rows])
    return s


class EmuDevice:
    """the host lane emulation (tests/emu) behind the device protocol of resynced_parity: the kernels' arithmetic on the CPU"""

    def __init__(self, emu):
        self.e = emu

    def set_state(self, s):
        self.e.set_state(s)

    def get_state(self):
        return self.e.get_state()

    def step(self, a):
        return self.e.step(a)

    def reset(self, mask):
        self.e.reset(mask)

    def reset_to(self, mask, states):
        self.e.reset_to(states, mask)

    def extra(self):
        return dict(torque=self.e.get("R_TAU_PD", 12), foot_force=self.e.get("R_FOOT_FORCE", 4))

    def flags(self):
        return self.e.get("R_FOOT_CONTACT", 4)


class VecEnvDevice:
    """QuadrupedVecEnv (the HIP path through the C ABI) behind the same protocol"""

    def __init__(self, v):
        self.v = v

    def set_state(self, s):
        self.v.set_state(np.asarray(s, np.float32))

    def get_state(self):
        return self.v.get_state().cpu().numpy()

    def step(self, a):
        obs, rew, done, infos = self.v.step(a)
        return obs, rew, done, np.array([inf.get("TimeLimit.truncated", False) for inf in infos])

    def reset(self, mask):
        self.v.reset_tensor(np.asarray(mask, np.uint8))

    def reset_to(self, mask, states):
        self.v.reset_tensor(np.asarray(mask, np.uint8), states=np.asarray(states, np.float32))

    def extra(self):
        out = dict(reward_end=self.v.get_info("reward_end").cpu().numpy()[:, :1])
        if self.v.cfg.info_fields:
            out.update(torque=self.v.get_info("torque").cpu().numpy(), foot_force=self.v.get_info("foot_force").cpu().numpy())
        return out

    def flags(self):
        return self.v.get_info("foot_contact").cpu().numpy() if self.v.cfg.info_fields else None


def rough_action(d):
    """explosive extension: flight, bad landings, falls"""
    if d == 5:
        return np.array([1.0, 1.0, 1.0, 1.0, -1.0], np.float32)
    return (np.tile([0.0, -1.0, 1.0], 4)[:d] if d != 4 else np.tile([-1.0, 1.0], 2)).astype(np.float32)


def resynced_parity(o, o32, dev, cfg, layout, steps=100, thrown_steps=0, seed=1, trials=12):
    """One env.step after another from the ORACLE's state (so that chaotic divergence cannot accumulate): the device against the float64
    oracle `o`, strictly wherever no non-foot link touched the ground inside the step, and under the yardstick of this module where one did
    (the many-rows solve: an impact).  `o32` is the oracle's float32 build, stepped alongside from the same states.  After `steps` steps of
    the scripted hops `thrown_steps` more in which half of the robots start the step thrown at the floor (thrown_states): hundreds of
    impacts instead of the handful a hopping run meets.  Returns the record of what was compared how (the impact steps' |device -
    oracle64| next to the oracle's own float32 / float64 spread)."""
    n, d = cfg.n_envs, cfg.action_dim
    ys = Yardstick(layout, EXTRA_FIELDS)
    rng, prng = np.random.default_rng(seed), np.random.default_rng(seed + 1000)
    o32_ok = np.ones(n, bool)
    names = [g[0] for g in STATE_GROUPS]
    rec = dict(env_steps=0, impact_env_steps=0, second_yardstick_env_steps=0, done_on_one_side_only=0, flag_flips=0, outliers=[],
               impact_dev={k: [] for k in names}, impact_own={k: [] for k in names})
    strict = {k: np.zeros((n, len(g))) for k, (g, _) in ys.fields.items()}
    stance = o.get_state()          # (the caller has reset all three)
    for i in range(steps + thrown_steps):
        a = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
        if i % 40 > 25:
            a[: max(n // 2, 1)] = rough_action(d)
        s = o.get_state()
        if i >= steps:
            s = thrown_states(s, np.arange(n) % 2 == (i % 2), rng)
        o.set_state(s); o32.set_state(s); dev.set_state(s.astype(np.float32))
        snap = o.snapshot()
        r64, r32, rd = o.step(a), o32.step(a), dev.step(a)
        ref, own, got = ys.collect(o, r64, oracle_extra(o)), ys.collect(o32, r32, oracle_extra(o32)), ys.collect(dev, rd, dev.extra())
        hit = o.get_info(5)[:, 0] > 0
        second = None
        # an episode that ends on one side only: allowed where the ORACLE's own verdict flips between its two builds or from a state 1e-6
        # away (a link that reaches its contact range in the step's last substep); such an environment is not compared in this step
        odd = (rd[2] != r64[2]) | (rd[3] != r64[3])
        if odd.any():
            second = what_if(o, snap, s, a, ys, ref, r64[2], trials, prng)
            excused = second[1] | (r32[2] != r64[2])
            assert not (odd & ~excused).any(), f"done / truncated step {i}: environments {np.flatnonzero(odd & ~excused).tolist()} (device {rd[2][odd].tolist()}, oracle {r64[2][odd].tolist()})"
            rec["done_on_one_side_only"] += int(odd.sum())
        cmp_ = ~odd
        rec["env_steps"] += int(cmp_.sum()); rec["impact_env_steps"] += int((hit & cmp_).sum())
        fl = dev.flags()
        same = np.ones((n, 4), bool)
        if fl is not None:
            # a foot whose distance sits within float32 rounding of the 0.727 mm contact range may be flagged on one side only (its force is
            # then a fraction of a newton): allowed for a couple of the smooth steps' flags; an impact step's flags follow its velocities
            same = fl == o.get_info(1)
            rec["flag_flips"] += int((~same)[~hit & cmp_].sum())
            assert rec["flag_flips"] <= 2, f"contact flags step {i}"
            if "foot_force" in got:      # (a foot flagged on one side only carries a fraction of a newton on that side: not compared)
                got["foot_force"] = np.where(same, got["foot_force"], ref["foot_force"])
        if (~hit & cmp_).any():
            bad = ys.excess(got, ref, strict, ~hit & cmp_, EXTRA_RTOL)
            assert not bad, f"step {i}, smooth environments: field -> (|device - oracle64| / tolerance, environment, column, |d|, tolerance) {bad}"
        rows = hit & cmp_
        if rows.any():
            spread = ys.spread(own, ref)
            for k in spread:            # an environment whose float32 oracle has parted ways with the float64 one (another `done`) measures nothing
                spread[k] = np.where(o32_ok[:, None], spread[k], 0.0)
            bad = ys.excess(got, ref, spread, rows, EXTRA_RTOL)
            if bad:
                rec["second_yardstick_env_steps"] += int(rows.sum())
                second = second or what_if(o, snap, s, a, ys, ref, r64[2], trials, prng)
                spread = ys.wider(spread, second[0])
                bad = ys.excess(got, ref, spread, rows, EXTRA_RTOL)
            if bad:     # a two-sided discontinuity shows in few of the draws (round 5, kw19 step 59: the oracle's angular velocity jumps by 0.6 rad/s in 2 of 8): four times as many
                spread = ys.wider(spread, what_if(o, snap, s, a, ys, ref, r64[2], 4 * trials, prng)[0])
                bad = ys.excess(got, ref, spread, rows, EXTRA_RTOL)
            if bad:
                # Still outside 5 x the largest of 62 evaluations of the oracle itself.  The bound is a sample maximum of a heavy-tailed
                # quantity, the device's deviation one more draw of it: a run of thousands of impact rows meets a few such draws.  Recorded,
                # and bounded in number and size by the caller (one per case, within 3 x the bound): a wrong kernel is wrong in every row
                # and fails the percentiles, an outlier is alone.
                rec["outliers"].append(dict(step=i, fields={k: [round(float(x), 6) for x in v] for k, v in bad.items()}))
                assert max(v[0] for v in bad.values()) <= 3.0, (f"step {i}, environments with a link on the ground: field -> (|device - oracle64| / (tolerance + 5 x the "
                                                                f"oracle's own spread over 62 evaluations), environment, column, |d|, bound) {bad}")
            dv, ow = group_spread(got["state"], ref["state"], STATE_GROUPS), group_spread(own["state"], ref["state"], STATE_GROUPS)
            for g, name in enumerate(names):
                rec["impact_dev"][name] += dv[rows, g].tolist(); rec["impact_own"][name] += ow[rows & o32_ok, g].tolist()
        o32_ok &= r32[2] == r64[2]
        end = r64[2] | rd[2]
        if end.any():
            m = end.astype(np.uint8)
            if i >= steps:          # the thrown phase: reference-state initialisation with the settled stance instead of 2500 settle substeps per reset
                o.reset_to(stance, m); o32.reset_to(stance, m); dev.reset_to(m, stance)
            else:
                o.reset(m); o32.reset(m); dev.reset(m)
            o32_ok |= end
    return rec


class FreeEmu:
    """the host lane emulation as a free-running device for terminal_observation_parity (auto_reset handle)"""

    def __init__(self, emu):
        self.e, self.dt = emu, float(emu.cfg.dt)

    def before_step(self):
        return self.e.get_state(), self.e.get("R_FOOT_FORCE", 4) * self.dt

    def step(self, a):
        return self.e.step(a)

    def terminal_obs(self):
        return self.e.get_term_obs()

    def touching(self):
        return self.e.get("R_FOOT_CONTACT", 4) > 0.5

    def state(self):
        return self.e.get_state()


class FreeVecEnv:
    """QuadrupedVecEnv (auto_reset, look-ahead resets) running free: the tensor API, as bench.py drives it"""

    def __init__(self, v):
        self.v, self.dt = v, float(v.cfg.dt)

    def before_step(self):
        return self.v.get_state().cpu().numpy(), self.v.get_info("foot_force").cpu().numpy() * self.dt

    def step(self, a):
        t = self.v.torch
        return tuple(x.cpu().numpy() for x in self.v.step_tensor(t.from_numpy(a).to(self.v.device)))

    def terminal_obs(self):
        return self.v.get_info("terminal_obs").cpu().numpy()

    def touching(self):
        return self.v.get_info("foot_contact").cpu().numpy() > 0.5

    def state(self):
        return self.v.get_state().cpu().numpy()


def oracle_sampled_parity(dev, n, blocks, oracles, oracles32, layout, d, steps=100, rng=None, trials=12):
    """A free-running device (auto-reset, its own resets) with `blocks` = [(first environment, count)] of its environments shadowed by the
    float64 and float32 oracles (already reset by the caller), re-seated before every step in the device's rigid-body state and contact
    warm start; every running environment's state after the step is held to `strict tolerance + 5 x |oracle32 - oracle64|` per group of
    like quantities, widened -- where that does not cover the device -- by the float64 oracle's own step from `trials` states 1e-6 away.
    An env step in which the set of touching feet changes (on the device, or between device and oracle) sits AT a discontinuity of the step
    map: counted as `switching`, by cause.  Observation / reward / flags of the environments away from any discontinuity, the settled
    states and reset observations of the episodes that begin."""
    rng = rng or np.random.default_rng(0)
    ys = Yardstick(layout)
    prng = np.random.default_rng(99)
    finished = strict = switching = 0
    second = dict(switching=0, other=0)
    causes = dict(touch_down=0, lift_off=0, both=0, flags_differ_only=0)
    names = [g[0] for g in STATE_GROUPS]
    sw_dev, sw_own = {k: [] for k in names}, {k: [] for k in names}
    tol_row = bound(np.zeros((1, len(STATE_GROUPS))), STATE_GROUPS, 37)[0]
    rough = rough_action(d)
    for i in range(steps):
        a = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
        if i % 20 > 8:   # explosive extension in half of every block: flight, bad landings, terminations inside the run
            for b, k in blocks:
                a[b:b + k // 2] = rough
        s, warm = dev.before_step()
        touching = dev.touching()
        snaps = []
        for o, p, (b, k) in zip(oracles, oracles32, blocks):
            o.set_state(s[b:b + k]); o.set_warm(warm[b:b + k])
            p.set_state(s[b:b + k]); p.set_warm(warm[b:b + k])
            snaps.append(o.snapshot())
        vo, rv, dv, tv = dev.step(a)
        dv, tv = np.asarray(dv).astype(bool), np.asarray(tv).astype(bool)
        sv, touching_after = dev.state(), dev.touching()
        for o, p, (b, k), snap in zip(oracles, oracles32, blocks, snaps):
            sl = slice(b, b + k)
            r64, r32 = o.step(a[sl]), p.step(a[sl])
            oo, ro, do, to = r64
            d32 = r32[2]
            ref, own = ys.collect(o, r64), ys.collect(p, r32)
            so = ref["state"]
            got = dict(state=sv[sl].astype(np.float64), obs=np.asarray(vo[sl], np.float64), reward=np.asarray(rv[sl], np.float64)[:, None])
            flags_o = o.get_info(1) > 0.5
            down, up = (~touching[sl] & touching_after[sl]).any(axis=1), (touching[sl] & ~touching_after[sl]).any(axis=1)
            switch = down | up | (touching[sl] != flags_o).any(axis=1)
            same_end = dv[sl] == do
            assert same_end[~switch].all(), f"done, step {i} block {b}"
            run = ~do & same_end
            spread = {"state": np.where((d32 == do)[:, None], ys.spread(own, ref)["state"], 0.0)}
            bad = ys.excess(got, ref, spread, run)
            if bad:
                w = what_if(o, snap, s[sl].astype(np.float64), a[sl], ys, ref, do, trials, prng, warm=warm[sl], extra=False)[0]
                over = (np.abs(got["state"] - so) > bound(spread["state"], STATE_GROUPS, 37)).any(axis=1) & run
                second["switching"] += int((over & switch).sum()); second["other"] += int((over & ~switch).sum())
                spread = {"state": np.maximum(spread["state"], w["state"])}
                bad = ys.excess(got, ref, spread, run)
            assert not bad, (f"state, step {i} block {b}: field -> (|device - oracle64| / (tolerance + 5 x the oracle's own spread), environment of the block, "
                             f"column, |d|, bound): {bad}")
            strict += int((run & ~switch).sum()); switching += int((run & switch).sum())
            for key, m in (("both", down & up), ("touch_down", down & ~up), ("lift_off", up & ~down), ("flags_differ_only", switch & ~down & ~up)):
                causes[key] += int((run & m).sum())
            if (run & switch).any():
                dvs, ows = group_spread(got["state"], so, STATE_GROUPS), group_spread(own["state"], so, STATE_GROUPS)
                for g, name in enumerate(names):
                    sw_dev[name] += dvs[run & switch, g].tolist(); sw_own[name] += ows[run & switch & (d32 == do), g].tolist()
            ok = run & ~switch & (np.abs(own["state"] - so) <= tol_row).all(axis=1)          # outputs of the environments away from any discontinuity
            np.testing.assert_array_equal(tv[sl][ok], to[ok], err_msg=f"truncated, step {i} block {b}")
            np.testing.assert_allclose(rv[sl][ok], ro[ok], atol=2e-4, rtol=1e-3, err_msg=f"reward, step {i} block {b}")
            np.testing.assert_allclose(vo[sl][ok], oo[ok], atol=TOL_QD, err_msg=f"observation, step {i} block {b}")
            both = do & same_end                   # (a finished environment holds its NEXT episode's settled state: looser, as every reset)
            np.testing.assert_allclose(sv[sl][both], so[both], atol=1e-3, err_msg=f"settled state of the next episode, step {i} block {b}")
            np.testing.assert_allclose(vo[sl][both], oo[both], atol=TOL_QD, err_msg=f"reset observation, step {i} block {b}")
            finished += int(both.sum())
            if not same_end.all():                 # an episode that ended on one side only (at a discontinuity): bring the oracles' episode along
                m = (dv[sl] & ~do).astype(np.uint8)
                if m.any():
                    o.reset(m); p.reset(m)
                assert not (do & ~dv[sl]).any(), f"the oracle ended an episode the device did not, step {i} block {b}"
            if (do & ~d32).any():
                p.reset((do & ~d32).astype(np.uint8))
    return dict(episodes_finished=finished, strict=strict, switching=switching, switching_by_cause=causes, env_steps_that_needed_the_second_yardstick=second,
                switching_abs_dev={k: dict(device_p50_p90_p99=percentiles(sw_dev[k]), oracle32_p50_p90_p99=percentiles(sw_own[k])) for k in names})


def terminal_observation_parity(dev, n, blocks, make_oracle, layout, d, target=2000, max_steps=600, seed=0, progress=None):
    """infos[i]["terminal_observation"] and the terminal reward of FALL-ENDED episodes, the outputs the body_contacts=True default exists for
    (task_base.py:137-147, quadruped.py:224-258, 533-539; SB3 hands the observation out, load_model.py:133): the device runs free (auto-reset,
    its own look-ahead resets), the float64 and the float32 oracle shadow `blocks` = [(first environment, count)], re-seated before every
    step in the device's rigid-body state and contact warm start.  For every episode that device and float64 oracle end in the same step by
    a fall (done without TimeLimit.truncated): |device - oracle64| per sensor, next to |oracle32 - oracle64| of the same step.  Runs until
    `target` such episodes are in.  -> record with the percentiles of both, per sensor."""
    rng = np.random.default_rng(seed)
    groups = obs_groups(layout)
    o64 = [make_oracle(b, k, "f64") for b, k in blocks]
    o32 = [make_oracle(b, k, "f32") for b, k in blocks]
    for o in o64 + o32:
        o.reset()
    lost = [np.zeros(k, bool) for _, k in blocks]       # the float64 oracle ended an episode the device did not: its episode count has moved on
    lost32 = [np.zeros(k, bool) for _, k in blocks]
    err_dev, err_own = {g[0]: [] for g in groups}, {g[0]: [] for g in groups}
    rew_dev, rew_own = [], []
    falls = dev_only = oracle_only = own_pairs = time_limits = steps = 0
    rough = rough_action(d)
    for i in range(max_steps):
        a = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
        if i % 20 > 8:       # explosive extension in half of every block: flight, bad landings, falls
            for b, k in blocks:
                a[b:b + k // 2] = rough
        s, warm = dev.before_step()
        for o, p, (b, k) in zip(o64, o32, blocks):
            o.set_state(s[b:b + k]); o.set_warm(warm[b:b + k]); p.set_state(s[b:b + k]); p.set_warm(warm[b:b + k])
        vo, rv, dv, tv = dev.step(a)
        dv, tv = np.asarray(dv).astype(bool), np.asarray(tv).astype(bool)
        term = dev.terminal_obs() if dv.any() else None
        steps += 1
        for j, (o, p, (b, k)) in enumerate(zip(o64, o32, blocks)):
            sl = slice(b, b + k)
            oo, ro, do, to = o.step(a[sl])
            _, r32, d32, t32 = p.step(a[sl])
            fall_o, fall_d, fall_p = do & ~to, dv[sl] & ~tv[sl], d32 & ~t32
            both = fall_o & fall_d & ~lost[j]
            time_limits += int((do & to & dv[sl] & tv[sl]).sum())
            if both.any():
                t64, tdev = o.get_info(9)[both].astype(np.float64), term[sl][both].astype(np.float64)
                e = group_spread(tdev, t64, groups)
                for g, (name, _, _, _) in enumerate(groups):
                    err_dev[name] += e[:, g].tolist()
                rew_dev += np.abs(rv[sl][both].astype(np.float64) - ro[both]).tolist()
                falls += int(both.sum())
            pair = fall_o & fall_p & ~lost[j] & ~lost32[j]
            if pair.any():
                e = group_spread(p.get_info(9)[pair].astype(np.float64), o.get_info(9)[pair].astype(np.float64), groups)
                for g, (name, _, _, _) in enumerate(groups):
                    err_own[name] += e[:, g].tolist()
                rew_own += np.abs(r32[pair].astype(np.float64) - ro[pair]).tolist()
                own_pairs += int(pair.sum())
            # episodes that ended on one side only (a link reaching its contact range in the step's last substep on one side): counted; the
            # oracle follows the device's episode where it can
            m = dv[sl] & ~do & ~lost[j]
            if m.any():
                dev_only += int(m.sum()); o.reset(m.astype(np.uint8))
            m = do & ~dv[sl] & ~lost[j]
            oracle_only += int(m.sum()); lost[j] |= m
            m = do & ~d32 & ~lost32[j]
            if m.any():
                p.reset(m.astype(np.uint8))
            lost32[j] |= d32 & ~do
        if progress and i % 20 == 19:
            progress(f"step {i + 1}: {falls} fall-ended episodes compared")
        if falls >= target:
            break
    for o in o64 + o32:
        o.close()
    rec = dict(steps=steps, shadowed_envs=int(sum(k for _, k in blocks)), fall_ended_episodes=falls, float32_oracle_pairs=own_pairs, time_limit_episodes=time_limits,
               ended_on_the_device_only=dev_only, ended_in_the_oracle_only=oracle_only, sensors={}, quantiles=[50, 90, 99])
    for name, _, _, tol in groups:
        rec["sensors"][name] = dict(device=percentiles(err_dev[name]), oracle32=percentiles(err_own[name]), floor=tol,
                                    device_max=float(np.max(err_dev[name])) if err_dev[name] else None, oracle32_max=float(np.max(err_own[name])) if err_own[name] else None)
    rec["terminal_reward"] = dict(device=percentiles(rew_dev), oracle32=percentiles(rew_own), floor=2e-4)
    return rec


def assert_inside_own_spread(rec, factor=2.0):
    """90th and 99th percentile of |device - oracle64| within floor + factor x the float32 oracle's own"""
    for name, r in list(rec["sensors"].items()) + [("terminal_reward", rec["terminal_reward"])]:
        for q in (1, 2):
            assert r["device"][q] <= r["floor"] + factor * r["oracle32"][q], f"{name}: |device - oracle64| p50 / p90 / p99 {r['device']} against the float32 oracle's own {r['oracle32']}"


def usable_cores(cap=16):
    """host threads the oracle may use at once: the affinity mask cut down to the cgroup's CPU quota (the GPU box shows 256 CPUs and grants 16)"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def percentiles(x, qs=(50, 90, 99)):
    x = np.asarray(x, np.float64)
    return [float(np.percentile(x, q)) if x.size else None for q in qs]
