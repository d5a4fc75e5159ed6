"""Builds liblslam_gpu.so (the C-ABI shared library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  -ffp-contract=off is REQUIRED: lattice cells and lookup-table indices are
rounded fp64 expressions whose products/sums the reference rounds separately (csrc/karto_math.hpp).
"""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "liblslam_gpu.so"
SOURCES = ["context.hip", "scan_matcher.hip", "logodds_map.hip", "occupancy_grid.hip", "pool.hip", "deskew.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-result"]


STAMP = PKG / "liblslam_gpu.so.srchash"


def _source_hash() -> str:
    """Content hash of everything the library is built from (mtimes do not survive a snapshot copy)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS + SOURCES).encode())
    deps = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "lslam_gpu.h"]
    for p in deps:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _stale() -> bool:
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != _source_hash()


def build_library(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile csrc/*.hip -> liblslam_gpu.so if missing or out of date.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if LIB.exists():  # a box without the compiler still runs the prebuilt library
            return LIB
        raise RuntimeError("hipcc not found and liblslam_gpu.so has not been built")
    cmd = [hipcc, *FLAGS, *[str(CSRC / s) for s in SOURCES], "-o", str(LIB)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(" ".join(cmd))
        print(r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed building liblslam_gpu.so")
    STAMP.write_text(_source_hash() + "\n")
    return LIB
