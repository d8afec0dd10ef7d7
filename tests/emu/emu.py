"""ctypes driver for the TEST-ONLY host emulation of the kernel arithmetic (tests/emu/qs_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, "libqs_emu.so")
import glob
# every header under csrc/ counts (round 4's hand-kept list lacked qs_rare.h: an edit of the many-rows solver left a stale emulation behind)
_SRC = [os.path.join(_HERE, "qs_emu.cpp")] + sorted(glob.glob(os.path.join(_REPO, "quadruped-springs_amd", "csrc", "*.h"))) + \
       [os.path.join(_REPO, "include", "qs_amd.h")]


def build():
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-ffp-contract=off",
                               "-I" + os.path.join(_REPO, "include"), "-o", _SO, _SRC[0]])
    return _SO


class Emu:
    def __init__(self, cfg):
        self.lib = C.CDLL(build())
        self.lib.qse_create.restype = C.c_void_p
        self.lib.qse_records.restype = C.POINTER(C.c_float)
        self.cfg = cfg
        self.n, self.d, self.o = cfg.n_envs, cfg.action_dim, cfg.obs_dim
        self.h = C.c_void_p(self.lib.qse_create(C.byref(cfg)))
        self.rec_size = self.lib.qse_rec_size()

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def field(self, name):
        return self.lib.qse_field(name.encode())

    def records(self):
        return np.ctypeslib.as_array(self.lib.qse_records(self.h), shape=(self.n, self.rec_size))

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.qse_reset(self.h, None if m is None else self._p(m))
        return self.get_obs()

    def get_obs(self):
        obs = np.zeros((self.n, self.o), np.float32)
        self.lib.qse_get_obs(self.h, self._p(obs))
        return obs

    def get_term_obs(self):
        obs = np.zeros((self.n, self.o), np.float32)
        self.lib.qse_get_term_obs(self.h, self._p(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n, self.d)
        obs = np.zeros((self.n, self.o), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        self.lib.qse_step(self.h, self._p(a), self._p(obs), self._p(rew), self._p(done), self._p(trunc))
        return obs, rew, done.astype(bool), trunc.astype(bool)

    def set_trace(self, env):
        self._trace = np.zeros((self.cfg.action_repeat, 70), np.float32)
        self.lib.qse_set_trace(self.h, int(env), self._p(self._trace))
        return self._trace

    def get_state(self):
        s = np.zeros((self.n, 37), np.float32)
        self.lib.qse_get_state(self.h, self._p(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32).reshape(self.n, 37)
        self.lib.qse_set_state(self.h, self._p(s))

    def phys_step(self, env, tau):
        t = np.ascontiguousarray(tau, np.float32)
        self.lib.qse_phys_step(self.h, env, self._p(t))

    def obb_overlap(self, ca, Ra, ha, cb, Rb, hb):
        a = [np.ascontiguousarray(x, np.float32) for x in (ca, Ra, ha, cb, Rb, hb)]
        return bool(self.lib.qse_obb_overlap(*[self._p(x) for x in a]))

    def get(self, name, dim):
        f = self.field(name)
        return self.records()[:, f:f + dim].copy()

    def block(self):
        """the payload block as its own body (payload="soft"): rows as in Oracle.block()"""
        return self.get("R_BLOCK", 20)

    def set_mu(self, mu):
        self.records()[:, self.field("R_PARAMS")] = mu

    def reset_to(self, states, mask=None):
        st = np.ascontiguousarray(states, np.float32).reshape(self.n, 37)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.qse_reset_to(self.h, None if m is None else self._p(m), self._p(st))
        return self.get_obs()

    def set_demo(self, rows):
        r = np.ascontiguousarray(rows, np.float32).reshape(-1, self.d + 38)
        self.lib.qse_set_demo(self.h, self._p(r), int(r.shape[0]))

    def set_demo_counter(self, values, mask=None):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, np.int32), (self.n,)))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.qse_set_demo_counter(self.h, None if m is None else self._p(m), self._p(v))
