"""Builds fastkv_amd/lib/libfastkv_hip.so (the C-ABI library of include/fastkv_hip.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
The .so is built in-tree (git-ignored, shipped by gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfastkv_hip.so")
SOURCES = ["score.hip", "fused.hip", "select.hip", "compact.hip", "capi.hip", "debug.hip", "prof.hip"]
HEADERS = ["fk_device.h", "fk_host.h", "prof.h", "rank.h", "mfma_tile.h", os.path.join("..", "..", "include", "fastkv_hip.h")]
# -ffp-contract=off: the arithmetic contract (csrc/fk_device.h) spells every fma out; nothing may be fused or split
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-result"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the fastkv_amd HIP extension cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # several ranks of one node may get here together: one builds (to a temporary name, then an atomic rename), the others
    # wait on the lock and find the library up to date
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            extra = os.environ.get("FASTKV_CXXFLAGS", "").split()          # measurement builds only (e.g. -DFK_STAMP)
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc(), *HIPCC_FLAGS, *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
