#!/usr/bin/env python3
"""Timing-only variants of the library (outputs are wrong): scripts/ablate/lib_<name>.so, selected at
run time with NZ_LIB_PATH.  Usage: python scripts/build_ablations.py NAME=-DFLAG[,-DFLAG...] ...
e.g.  python scripts/build_ablations.py noA=-DNZ_ABLATE_A noB=-DNZ_ABLATE_B"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "nuzero_amd", "csrc")
sys.path.insert(0, REPO)
from nuzero_amd.build import COMMON, HIPCC, UNITS  # noqa: E402

out_dir = os.path.join(REPO, "scripts", "ablate")
os.makedirs(out_dir, exist_ok=True)
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    objs = []
    for src, extra in UNITS:
        o = os.path.join("/tmp", f"abl_{name}_{os.path.splitext(src)[0]}.o")
        subprocess.check_call([HIPCC] + COMMON + extra + flags.split(",") + ["-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    lib = os.path.join(out_dir, f"lib_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)
