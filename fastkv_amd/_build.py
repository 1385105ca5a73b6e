"""Builds fastkv_amd/lib/libfastkv_hip.so (the C-ABI library of include/fastkv_hip.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
The .so is built in-tree (git-ignored, shipped by gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# FASTKV_BUILD_DIR: build (and load) the library somewhere else than in-tree -- for instrumented builds that must not replace
# the product's .so (e.g. FASTKV_CXXFLAGS=-DFK_STAMP, -DFK_DEBUG_BOUNDS)
LIBDIR = os.environ.get("FASTKV_BUILD_DIR") or os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfastkv_hip.so")
SOURCES = ["score.hip", "fused.hip", "select.hip", "compact.hip", "sp.hip", "decode.hip", "decode_step.hip", "gemv.hip", "capi.hip", "debug.hip", "prof.hip"]
HEADERS = ["fk_device.h", "fk_hunt.h", "fk_host.h", "prof.h", "rank.h", "mfma_tile.h", os.path.join("..", "..", "include", "fastkv_hip.h")]
# -ffp-contract=off: the arithmetic contract (csrc/fk_device.h) spells every fma out; nothing may be fused or split
# `-target-feature -packed-fp32-ops`: no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in the device code (the same IEEE operations per
# component, issued one by one: +1.4 us on the 77.5 us two-layer scoring launch in the trace, inside the noise of the step).  Why: a workgroup of the fused scoring kernel that
# ran its packed-fp32 phases beside a partner's matrix phase on one compute unit computed wrong values now and then; without the packed
# instructions the strongest reproducer of that goes from 40 % wrong launches to 0 of 600 (docs/HISTORY.md).  The host half of the
# compilation does not know the feature and says so ("not a recognized feature for this target (ignoring feature)"): harmless.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-result",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
OBJDIR = os.path.join(LIBDIR, "obj")


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the fastkv_amd HIP extension cannot be built")
    return exe


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile_one(src: str, extra, verbose: bool) -> str:
    """One translation unit -> object file (kept under lib/obj so that an edit of one kernel file recompiles that file only)."""
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    stamp = obj + ".flags"
    flags = " ".join(HIPCC_FLAGS + list(extra))
    newest = max(os.path.getmtime(d) for d in [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS])
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest and os.path.exists(stamp) and open(stamp).read() == flags:
        return obj
    cmd = [hipcc(), *HIPCC_FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flags)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    # several ranks of one node may get here together: one builds (to a temporary name, then an atomic rename), the others
    # wait on the lock and find the library up to date
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            extra = os.environ.get("FASTKV_CXXFLAGS", "").split()          # measurement builds only (e.g. -DFK_STAMP)
            if extra and not os.environ.get("FASTKV_BUILD_DIR"):
                # needs_build() knows nothing about these flags: an instrumented build in-tree would either silently stay the
                # product's .so or silently replace it
                raise RuntimeError("FASTKV_CXXFLAGS needs FASTKV_BUILD_DIR=<scratch directory>: measurement builds never go in-tree")
            if force:
                for f in os.listdir(OBJDIR):
                    os.remove(os.path.join(OBJDIR, f))
            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
                objs = list(ex.map(lambda s: _compile_one(s, extra, verbose), SOURCES))
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
