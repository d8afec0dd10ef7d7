#!/usr/bin/env python3
"""Where the SB3 (numpy) path spends its time per step, and what kind of host buffer the results should land in.

Measures, for the [8192, 30] float32 result block of one step (983 KB): the device-to-host copy plus a host-side read of the data
(numpy copy) for (a) torch's pageable `.cpu()`, (b) torch pinned memory (hipHostMalloc default flags), (c) hipHostMalloc with
hipHostMallocNonCoherent, (d) ... with hipHostMallocNumaUser | NonCoherent.  Run on the GPU box."""
import ctypes as C
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")     # torch has loaded it already: same runtime
N, O = 8192, 30
dev = torch.device("cuda", 0)
src = torch.randn((N, O), device=dev)
nbytes = src.numel() * 4
stream = torch.cuda.current_stream().cuda_stream


def bench(name, fetch, reps=200):
    for _ in range(10):
        fetch()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fetch()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:58s} {dt * 1e6:8.1f} us per step  ({nbytes / dt / 1e9:6.2f} GB/s)   checksum {float(out.sum()):.3f}")


bench("pageable: tensor.cpu().numpy() + copy", lambda: src.cpu().numpy().copy())
pinned = torch.empty((N, O), dtype=torch.float32, pin_memory=True)


def via_pinned():
    pinned.copy_(src, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return pinned.numpy().copy()


bench("torch pinned (hipHostMalloc default): copy_ + sync + copy", via_pinned)
for flag_name, flags in (("hipHostMallocNonCoherent", 0x80000000), ("hipHostMallocNumaUser|NonCoherent", 0xA0000000), ("hipHostMallocCoherent", 0x40000000),
                         ("hipHostMallocDefault", 0)):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), C.c_uint(flags))
    if rc != 0:
        print(flag_name, "hipHostMalloc failed", rc)
        continue
    host = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(N, O))

    def via_host(p=p, host=host):
        hip.hipMemcpyAsync(p, C.c_void_p(src.data_ptr()), C.c_size_t(nbytes), C.c_int(2), C.c_void_p(stream))   # hipMemcpyDeviceToHost
        hip.hipStreamSynchronize(C.c_void_p(stream))
        return host.copy()

    bench(f"{flag_name}: hipMemcpyAsync + sync + copy", via_host)

    def only_copy(p=p):
        hip.hipMemcpyAsync(p, C.c_void_p(src.data_ptr()), C.c_size_t(nbytes), C.c_int(2), C.c_void_p(stream))
        hip.hipStreamSynchronize(C.c_void_p(stream))
        return host[:1]

    bench(f"{flag_name}: hipMemcpyAsync + sync only", only_copy)
    hip.hipHostFree(p)

# (e) ordinary (cached, page-aligned) host memory made DMA-able with hipHostRegister: what the numpy path's result buffers are
import mmap
for flag_name, flags in (("hipHostRegisterDefault", 0), ("hipHostRegisterPortable|Mapped", 3)):
    mm = mmap.mmap(-1, (nbytes + 4095) // 4096 * 4096)
    host = np.frombuffer(mm, dtype=np.float32, count=N * O).reshape(N, O)
    host[:] = 0
    addr = C.c_void_p(C.addressof(C.c_char.from_buffer(mm)))
    rc = hip.hipHostRegister(addr, C.c_size_t(len(mm)), C.c_uint(flags))
    if rc != 0:
        print(flag_name, "hipHostRegister failed", rc)
        continue

    def via_reg(addr=addr, host=host):
        hip.hipMemcpyAsync(addr, C.c_void_p(src.data_ptr()), C.c_size_t(nbytes), C.c_int(2), C.c_void_p(stream))
        hip.hipStreamSynchronize(C.c_void_p(stream))
        return host.copy()

    def reg_only(addr=addr, host=host):
        hip.hipMemcpyAsync(addr, C.c_void_p(src.data_ptr()), C.c_size_t(nbytes), C.c_int(2), C.c_void_p(stream))
        hip.hipStreamSynchronize(C.c_void_p(stream))
        return host[:1]

    bench(f"{flag_name}: hipMemcpyAsync + sync + copy", via_reg)
    bench(f"{flag_name}: hipMemcpyAsync + sync only", reg_only)
    t0 = time.perf_counter()
    for _ in range(200):
        out = host.copy()
    print(f"{flag_name}: host-side numpy copy of the registered block alone {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us")
    hip.hipHostUnregister(addr)
