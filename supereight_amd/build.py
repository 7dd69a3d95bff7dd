"""Builds libse_hip.so (the C-ABI library of include/se_hip.h) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libse_hip.so")
SOURCES = ["se_hip_api.hip"]
HEADERS = ["se_device.h", "se_kernels.h", "se_track_kernels.h", "se_mesh_kernels.h", os.path.join("..", "..", "include", "se_hip.h"),
           os.path.join("..", "..", "include", "se_mc_table.h")]
# -ffp-contract=off is part of the numeric contract (no FMA contraction in device or host code)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(SRC_DIR, s) for s in SOURCES]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout, res.stderr)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
