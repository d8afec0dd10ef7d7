#!/usr/bin/env python3
"""What does body_contacts="auto" change against the reference's semantics (body_contacts=True)?   

PyBullet collides every URDF primitive with the plane (quadruped.py:533-539); the task reads the contacts of the LAST substep of an env
step only (task_base.py:137-147, quadruped.py:224-258).  `True` builds that response for trunk / hip / thigh / calf; "auto" leaves it off
under a task, so a falling robot's link sinks into the floor during the episode's last env step(s).

Two handles of the headline workload (Go1 + PEA, JUMPING_IN_PLACE, PPO_BASIC, GROUND_RANDOMIZER, N = 8192, auto-reset with look-ahead
resets), same seed.  Every environment's e-th episode starts from the same settled reset state in both handles (a reset state depends on
(seed, environment, episode) only) and is driven by the same actions: the action of a step is a function of (environment, step of the
episode), NOT of the global step, so that the e-th episodes stay comparable after the two handles' episode schedules have parted.
Observation noise is off (it would mask the physics; it does not act on the motion: the actions do not depend on observations).

Per episode: the step it ended in, whether by the time limit, its terminal observation and its last reward (step reward + end-of-episode
term).  Output: the share of episodes whose `done` step differs, percentiles of |delta terminal observation| per sensor and of |delta
terminal reward|, each against the parity tolerances of tests/test_gpu_parity.py.

    python tools/body_contacts_delta.py [steps=9000] [out.json]        -> markdown on stdout
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "quadruped-springs_amd")):
    sys.path.insert(0, p)

import numpy as np
import torch

from qs_amd.vec_env import QuadrupedVecEnv

TOL_Q, TOL_QD, TOL_BASE_V, TOL_POS = 2e-5, 5e-3, 5e-4, 5e-6     # tests/test_gpu_parity.py
N, RING, E_MAX = 8192, 64, 96
KW = dict(task_env="JUMPING_IN_PLACE", observation_space_mode="PPO_BASIC", env_randomizer_mode="GROUND_RANDOMIZER", enable_springs=True,
          enable_action_filter=True, action_space_mode="SYMMETRIC", motor_control_mode="PD", time_step=0.001, action_repeat=10,
          noise=False, info_fields=True)


class Recorder:
    def __init__(self, body_contacts, dev):
        self.env = QuadrupedVecEnv(num_envs=N, device=0, auto_reset=True, reset_lookahead=16, seed=1234, body_contacts=body_contacts, **KW)
        self.env.reset_tensor()
        o = self.env.obs_dim
        self.ep_step = torch.zeros(N, dtype=torch.long, device=dev)
        self.ep_idx = torch.zeros(N, dtype=torch.long, device=dev)
        self.length = torch.zeros((N, E_MAX), dtype=torch.int32, device=dev)
        self.trunc = torch.zeros((N, E_MAX), dtype=torch.uint8, device=dev)
        self.rew = torch.zeros((N, E_MAX), dtype=torch.float32, device=dev)
        self.term = torch.zeros((N, E_MAX, o), dtype=torch.float32, device=dev)
        self.ids = torch.arange(N, device=dev)

    def step(self, ring):
        a = ring[self.ep_step % RING, self.ids]
        obs, rew, done, trunc = self.env.step_tensor(a)
        d = done.bool()
        if bool(d.any()):
            i = d.nonzero().squeeze(1)
            e = self.ep_idx[i]
            ok = e < E_MAX
            i, e = i[ok], e[ok]
            self.length[i, e] = (self.ep_step[i] + 1).to(torch.int32)
            self.trunc[i, e] = trunc[i]
            self.rew[i, e] = rew[i]
            self.term[i, e] = self.env.get_info("terminal_obs")[i]
        self.ep_step = torch.where(d, torch.zeros_like(self.ep_step), self.ep_step + 1)
        self.ep_idx = self.ep_idx + d.long()


def pct(x, qs=(50, 90, 99, 99.9, 100)):
    return [float(np.percentile(x, q)) for q in qs] if x.size else [float("nan")] * len(qs)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
    out_json = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    ring = torch.rand((RING, N, 6), generator=g, device=dev) * 2 - 1
    a, b = Recorder(True, dev), Recorder("auto", dev)
    for _ in range(steps):
        a.step(ring); b.step(ring)
    torch.cuda.synchronize()
    ne = torch.minimum(a.ep_idx, b.ep_idx).clamp(max=E_MAX).cpu().numpy()            # episodes finished in BOTH handles
    mask = np.arange(E_MAX)[None, :] < ne[:, None]
    la, lb = a.length.cpu().numpy()[mask], b.length.cpu().numpy()[mask]
    ta, tb = a.trunc.cpu().numpy()[mask].astype(bool), b.trunc.cpu().numpy()[mask].astype(bool)
    ra, rb = a.rew.cpu().numpy()[mask], b.rew.cpu().numpy()[mask]
    oa, ob = a.term.cpu().numpy()[mask], b.term.cpu().numpy()[mask]
    n_ep = int(mask.sum())
    fell = ~(ta & tb)                                   # ended by the task in at least one handle
    same_step = la == lb
    lay = a.env.meta["layout"]
    cols, c0 = [], 0
    for key, dim in zip(lay["keys"], lay["dims"]):
        cols.append((str(key), c0, c0 + int(dim))); c0 += int(dim)
    tol_of = lambda name: TOL_Q if "pos" in name.lower() and "joint" in name.lower() or name.lower() in ("encoder",) else (TOL_QD if "vel" in name.lower() else TOL_QD)
    res = dict(steps=steps, n_envs=N, episodes_compared=n_ep, episodes_ended_by_the_task=int(fell.sum()),
               done_step_differs=int((~same_step).sum()), done_step_differs_share=float((~same_step).mean()),
               done_step_differs_share_of_task_ended=float((~same_step)[fell].mean()) if fell.any() else 0.0,
               ended_by_task_in_one_handle_only=int((ta != tb).sum()),
               episodes_longer_with_true=int((la > lb).sum()), episodes_longer_with_auto=int((lb > la).sum()),
               done_step_delta_percentiles=pct(np.abs(la.astype(int) - lb)[~same_step]),
               many_rows_wave_substeps_true=a.env.counter("limit_path_substeps"), resets_true=a.env.counter("resets"), resets_auto=b.env.counter("resets"),
               stalls=(a.env.counter("reset_stalls"), b.env.counter("reset_stalls")))
    cmp_ = same_step & fell                             # task-ended episodes that end in the same step: compare what the learner is handed
    d_obs = np.abs(oa[cmp_] - ob[cmp_])
    d_rew = np.abs(ra[cmp_] - rb[cmp_])
    res["compared_same_step_task_ended"] = int(cmp_.sum())
    res["sensors"] = {}
    for name, lo, hi in cols:
        d = d_obs[:, lo:hi].max(axis=1) if d_obs.size else np.zeros(0)
        res["sensors"][name] = dict(percentiles=pct(d), share_over_tol=float((d > TOL_QD).mean()) if d.size else 0.0)
    res["terminal_reward"] = dict(percentiles=pct(d_rew), share_over_1e_3=float((d_rew > 1e-3).mean()) if d_rew.size else 0.0,
                                  mean_true=float(ra[cmp_].mean()) if cmp_.any() else 0.0, mean_auto=float(rb[cmp_].mean()) if cmp_.any() else 0.0)
    tl = ta & tb & same_step                            # control: episodes that ran to the time limit in both -- nothing should differ
    res["control_time_limit_episodes"] = dict(n=int(tl.sum()), max_obs_delta=float(np.abs(oa[tl] - ob[tl]).max()) if tl.any() else 0.0,
                                              max_reward_delta=float(np.abs(ra[tl] - rb[tl]).max()) if tl.any() else 0.0)
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
    P = "50 / 90 / 99 / 99.9 / 100"
    f = lambda v: " / ".join(f"{x:.3g}" for x in v)
    print(f"# body_contacts=True (the reference's all-links contact response) against \"auto\" (off under a task): what a learner is handed\n")
    print(f"`tools/body_contacts_delta.py {steps}`: headline workload (N = {N}, JUMPING_IN_PLACE, PPO_BASIC, GROUND_RANDOMIZER, U(-1,1) actions as a function of "
          f"(environment, step of the episode), observation noise off), two handles, same seed; the e-th episode of an environment starts from the same "
          f"reset state and sees the same actions in both.\n")
    print(f"* episodes compared (finished in both handles): **{n_ep}**, of them ended by the task (a fall) in at least one handle: **{int(fell.sum())}** "
          f"({100 * fell.mean():.1f} %); ended by the task in ONE handle only: {int((ta != tb).sum())}")
    print(f"* **`done` step differs: {int((~same_step).sum())} episodes = {100 * (~same_step).mean():.2f} % of all, {100 * res['done_step_differs_share_of_task_ended']:.2f} % of the "
          f"task-ended ones** (longer with True: {res['episodes_longer_with_true']}, longer with \"auto\": {res['episodes_longer_with_auto']}; |delta steps| percentiles {P}: {f(res['done_step_delta_percentiles'])})")
    print(f"* control -- {int(tl.sum())} episodes that ran to the time limit in both handles: max |delta terminal observation| {res['control_time_limit_episodes']['max_obs_delta']:.3g}, "
          f"max |delta reward| {res['control_time_limit_episodes']['max_reward_delta']:.3g} (bitwise the same motion: no link ever came into range)")
    print(f"* many-rows wave-substeps of the True handle: {res['many_rows_wave_substeps_true']}; resets {res['resets_true']} / {res['resets_auto']}; stalls {res['stalls']}\n")
    print(f"Task-ended episodes that end in the SAME step in both handles ({int(cmp_.sum())}): |delta| of what the learner receives at that step "
          f"(`infos[i][\"terminal_observation\"]`, the step's reward incl. `_reward_end_episode`), percentiles {P}; parity tolerance of the GPU tests on an "
          f"observation: {TOL_QD} (tests/test_gpu_parity.py TOL_QD)\n")
    print("| sensor | columns | |delta| percentiles | share of episodes over the tolerance |\n|---|---|---|---|")
    for name, lo, hi in cols:
        s = res["sensors"][name]
        print(f"| {name} | {lo}..{hi - 1} | {f(s['percentiles'])} | {100 * s['share_over_tol']:.1f} % |")
    r = res["terminal_reward"]
    print(f"| terminal reward | - | {f(r['percentiles'])} | {100 * r['share_over_1e_3']:.1f} % over 1e-3 (mean {r['mean_true']:.4f} with True, {r['mean_auto']:.4f} with \"auto\") |")


if __name__ == "__main__":
    main()
