#!/usr/bin/env python3
"""Build the gfx950 shared library of the batched simulation step (hipcc cross-compiles without a GPU).

    python quadruped-springs_amd/build.py            # -> quadruped-springs_amd/qs_amd/libqs_hip.so
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "qs_hip.hip")
SRC_NORM = os.path.join(HERE, "csrc", "qs_norm.hip")
OUT = os.path.join(HERE, "qs_amd", "libqs_hip.so")


def deps():
    """Everything the library is compiled from: every file under csrc/ and every public header (listed by directory, not by name: round 4's
    qs_rare.h was missing from a hand-kept list, and an edit of the many-rows solver alone left a stale library behind)."""
    import glob
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.hip")) +
                  glob.glob(os.path.join(REPO, "include", "*.h")) + [os.path.abspath(__file__)])


def needs_build(out=None):
    out = out or OUT
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps())


def source_fingerprint():
    """sha256 over everything the library is compiled from (deps(): sources, public headers, this file with its flags) plus the
    environment variables that change the build.  A profile (profiles/r*_pmc.json) carries the fingerprint of the tree it was taken
    on; bench.py reports counter-derived figures only for a matching tree (a rebuilt library on another box still matches, an
    edited kernel does not)."""
    import hashlib
    h = hashlib.sha256()
    for d in deps():
        h.update(os.path.relpath(d, REPO).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for var in ("QS_HIPCC_EXTRA", "QS_MFMA_VGPR_FORM", "QS_OFFLOAD_ARCH"):
        h.update((var + "=" + os.environ.get(var, "")).encode() + b"\0")
    return h.hexdigest()


def fingerprint(path=None):
    """sha256 of the built library: what bench.py and the profile tools use to say which binary a number belongs to."""
    import hashlib
    h = hashlib.sha256()
    with open(path or OUT, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    # -fno-slp-vectorize: packed fp32 (v_pk_*) costs more moves than it saves here.  iterative-ilp: with one wave per SIMD
    # there is no other wave to hide a dependent instruction's latency, so the scheduler should chase ILP, not occupancy
    # (measured +6.5 % env-steps/s over the default strategy, same instructions, same results).
    # -ffinite-math-only -fno-signed-zeros -fno-trapping-math: the step produces no NaN / infinity and tests for none, so the
    # v_max_f32 x, x canonicalisations in front of every fmin / fmax / med3 can go (+5 % measured); reassociation and the rest of
    # -ffast-math are NOT enabled (measured slower, and the summation order is part of the parity with the host emulation).
    # -amdgpu-mfma-vgpr-form: the 4x4x1 MFMAs of the Delassus block write VGPRs, no v_accvgpr_read per result (+1 % at N = 8192).
    # -ffp-contract=on (hipcc's default is "fast"): which multiply of an expression is fused into an FMA is decided by the front end, per
    # source expression, and not by the back end per inlined copy.  Under "fast" the step's two builds (common path / full), the settle
    # loops of k_reset and of the settle lanes, ... are separate inlined copies of the same templates, and for expressions with two
    # candidate products the back end's choice depended on the copy's surroundings: the copies agreed bit for bit under the implicit cone
    # and differed in the last bit under the friction pyramid and with payload="soft" (round 3, tools/diag/r03_lanes.py) -- the
    # look-ahead resets and the independence of wave-mates need them to agree.  Cost 1.7 % on the headline at first (cross-statement fusions); 0.8 % since p + a * s on vectors is one expression (V3s, qs_core.h).
    # (round 4: ROCm 7.2's "AMDGPU Rewrite AGPR-Copy-MFMA" pass, which only has work under -amdgpu-mfma-vgpr-form, segfaults on some
    # register allocations of the step kernels -- eliminateSpillsOfReassignedVGPRs --: seen on k_step<false, false> for two harmless
    # variations of the many-rows solver's source.  The build retries without the flag, loudly; QS_MFMA_VGPR_FORM=0 leaves it out at once.)
    # NOT -greedy-regclass-priority-trumps-globalness=1 any more (end of round 4: about +1 % at N = 8192; round 6, A/B'd again on the final source,
    # three runs each on one box: 69.88 / 69.78 / 69.80 against 69.56 / 69.44 / 69.49 M with the links' response on, 110.07 / 109.85 / 109.95
    # against 110.39 / 110.35 / 110.40 M without -- +0.5 % on one figure, -0.4 % on the other: inside what two boxes differ by.  It was the other
    # half of the flag pair under which round 5's library came out miscompiled; one knob fewer at the edge of the register allocator).
    # NOT -split-spill-mode=size any more (round 4 had it next to the knob above, +0.3 %): round 5's source with the <0, 4> instantiation of the
    # many-rows solver's core came out of it MISCOMPILED -- 17 of 120 fallen-robot fuzz configurations off by 1e-2, the kernel depending on
    # its wave-mates -- while the same source without that option, without the knob above, or with two more instantiations passed, and a
    # build that ran <0, 4> and <0, 6> side by side on the same rows (-DQS_DBG_CORE4, tools/diag/core4_differential.py) found them equal bit
    # for bit: the damage is in the spill code around the solve, not in it.  Without the option the library is as fast (70.5 / 111.5 M
    # against 70.4 / 111.0 M, gpurun_out/ab_tmp2) and every parity gate is green again.
    vgpr_form = os.environ.get("QS_MFMA_VGPR_FORM", "1") != "0"

    def command(with_form):
        c = [hipcc(), "--offload-arch=" + os.environ.get("QS_OFFLOAD_ARCH", "gfx950"), "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize",
             "-ffinite-math-only", "-fno-signed-zeros", "-fno-trapping-math", "-ffp-contract=on",
             "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if with_form else []) + \
            os.environ.get("QS_HIPCC_EXTRA", "").split() + ['-DQS_SOURCE_SHA="' + source_fingerprint() + '"', "-I" + os.path.join(REPO, "include"), "-o", os.environ.get("QS_BUILD_OUT") or OUT, SRC, SRC_NORM]
        if verbose:
            c.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(c))
        return c

    if vgpr_form:
        r = subprocess.run(command(True), stderr=subprocess.PIPE, text=True)
        if r.returncode == 0:
            sys.stderr.write(r.stderr)
            return OUT
        if "Rewrite AGPR-Copy-MFMA" not in r.stderr:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, r.args)
        sys.stderr.write("build.py: WARNING: hipcc crashed in 'AMDGPU Rewrite AGPR-Copy-MFMA'; building WITHOUT -amdgpu-mfma-vgpr-form (about 1 % slower at N = 8192)\n")
    subprocess.check_call(command(False))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
