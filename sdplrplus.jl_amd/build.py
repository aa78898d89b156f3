"""hipcc recipe for lib/libsdplr_hip.so (gfx950 only; no torch, no cmake)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "sdplr_hip.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("common.h", "k_dense.h", "k_sparse.h", "k_scalar.h", "k_eig.h", "k_resident.h", "k_group.h")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "sdplr_hip.h"))
OUT = os.path.join(HERE, "lib", "libsdplr_hip.so")


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in [SRC] + DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off",  # keep mul/add as written: results comparable with the CPU oracle
           "-Wall", "-Wno-unused-function", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    for flag in os.environ.get("SDPLR_HIP_EXTRA_CFLAGS", "").split():   # dev builds (-DSDPLR_RS_STAMPS, -DSDPLR_STAMPS …)
        cmd.insert(1, flag)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
