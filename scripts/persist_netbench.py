#!/usr/bin/env python3
"""Shader ticks of one network pass of the persistent SCS route (nz_scs_netbench): BASELINE configs[3]'s ConvNet(32 x 8)
on 5x5, at 1 / 64 / 256 workgroups (4 game slots each).  NZ_LIB_PATH selects a timing-only variant of the library."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nuzero_amd._lib import lib                                    # noqa: E402
from nuzero_amd.boardnet import BoardNet                           # noqa: E402
from nuzero_amd.scs import ScsGameConfig                           # noqa: E402
from nuzero_amd.weights import synthetic_weights, convnet_param_shapes   # noqa: E402

cfg = ScsGameConfig(os.path.join(REPO, "tests", "golden", "scs_configs", "mirrored_5x5.yml"))
net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=8, max_batch=1024)
net.set_weights(synthetic_weights(0, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 8)))
phases = os.environ.get("NZ_NETBENCH_PHASES")   # stamped builds: also the ticks of each phase of the pass
names = {0: "K loops", 1: "epilogues", 2: "meetings", 3: "first operands", 4: "layer headers",
         8: "helper K loops", 9: "helper epilogues", 10: "helper meetings", 11: "helper first operands", 12: "helper layer headers"}
for active in (4, 2, 1):               # game slots of a workgroup that run passes (the others stay idle)
    for blocks in (1, 256):
        out = np.zeros(blocks * 4, np.uint64)
        st = lib.nz_scs_netbench(net._h, blocks, 50 | (active << 16), ctypes.c_void_p(out.ctypes.data))
        assert st == 0, st
        out = out.reshape(blocks, 4)[:, :active]
        print(f"{active} of 4 game slots, {blocks} workgroups: ticks per pass min {out.min()} median {int(np.median(out))} max {out.max()}")
        if phases and blocks == 256:
            for ph, name in names.items():
                out = np.zeros(blocks * 4, np.uint64)
                st = lib.nz_scs_netbench(net._h, blocks, 50 | (active << 16) | ((ph + 1) << 24), ctypes.c_void_p(out.ctypes.data))
                assert st == 0, st
                print(f"    {name}: median {int(np.median(out.reshape(blocks, 4)[:, :active]))}")
