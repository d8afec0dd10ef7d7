#!/usr/bin/env python3
"""qs_norm_step alone, back to back on the same arrays: microseconds per call in training and in evaluation mode, for one or more builds of
the library (A/B of the normalisation kernels without the step around them).
usage: python tools/norm_ab.py new=quadruped-springs_amd/qs_amd/libqs_hip.so [old:f32=path/to/older.so ...]   (":f32": qs_norm_create took floats)"""
import ctypes as C
import sys
import time

import torch

N, O = 8192, 28
obs = torch.randn((N, O), device="cuda"); rew = torch.randn(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    f32 = name.endswith(":f32")
    lib = C.CDLL(path)
    fl = C.c_float if f32 else C.c_double
    lib.qs_norm_create.argtypes = [C.c_int, C.c_int, fl, fl, fl, fl, C.c_int, C.POINTER(C.c_void_p)]
    lib.qs_norm_step.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p] * 2
    lib.qs_norm_destroy.argtypes = [C.c_void_p]; lib.qs_norm_destroy.restype = None
    h = C.c_void_p()
    assert lib.qs_norm_create(N, O, 10.0, 10.0, 0.99, 1e-8, 0, C.byref(h)) == 0
    res = {}
    for mode, training in (("training", 1), ("evaluation", 0)):
        for _ in range(50):
            lib.qs_norm_step(h, p(obs), p(rew), p(done), None, training, 1, 1, None, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000):
            lib.qs_norm_step(h, p(obs), p(rew), p(done), None, training, 1, 1, None, None)
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / 1000 * 1e6
    print(f"{name}: {res['training']:.1f} us per training call, {res['evaluation']:.1f} us per evaluation call")
    lib.qs_norm_destroy(h)
