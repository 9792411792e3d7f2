"""Build the engine's shared library in-tree with hipcc (gfx950 only).

    python -m nuzero_amd.build

Produces nuzero_amd/csrc/libnuzero_amd.so.  hipcc cross-compiles without a GPU,
so this also runs in the CPU-only build container.  tree.hip and the host
random streams are compiled with -ffp-contract=off: their double-precision
arithmetic must round exactly like the reference's (one operation at a time).
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libnuzero_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("NZ_EXTRA_CXXFLAGS", "").split()   # timing-only ablation builds
UNITS = [
    ("tree.hip", ["-ffp-contract=off"]),
    ("net.hip", ["-ffp-contract=off"]),
    ("selfplay.hip", ["-ffp-contract=off"]),
    ("engine.hip", ["-ffp-contract=off"]),
    ("scs.hip", ["-ffp-contract=off"]),
    ("scs_search.hip", ["-ffp-contract=off"]),
    ("boardnet.hip", []),
    ("replay.hip", ["-ffp-contract=off"]),
    ("loss.hip", []),
    ("rng_host.cpp", ["-ffp-contract=off"]),
]
HEADERS = ["engine.h", "tree_dev.hpp", "net_dev.hpp", "scs_dev.hpp", os.path.join("..", "..", "include", "nuzero_amd.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=True):
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
