#!/usr/bin/env python3
"""BASELINE configs[4]'s network + search alone (bench.py's scs_config5 without importing bench.py): for profilers."""
import json
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from nuzero_amd.boardnet import BoardNet  # noqa: E402
from nuzero_amd.scs import ScsGameConfig, ScsSelfPlay  # noqa: E402
from nuzero_amd.weights import synthetic_weights, recurrent_net_param_shapes  # noqa: E402
games = int(os.environ.get("NZ_CFG5_GAMES", "1024"))
cfg = ScsGameConfig(os.path.join(REPO, "tests", "golden", "scs_configs", "ten_by_ten.yml"))
search = {"Simulation": {"mcts_simulations": int(os.environ.get("NZ_CFG5_SIMS", "400")), "keep_subtree": True},
          "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
          "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04, "epsilon_random_exploration": 0.001,
                          "value_factor": 1, "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                          "root_dist_alpha": 0.15, "root_dist_beta": 1}}
net = BoardNet("recurrent", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=256, num_blocks=2, recall=True,
               value_activation="relu", max_batch=games, device=0)
net.set_weights(synthetic_weights(0, recurrent_net_param_shapes(cfg.channels, cfg.planes, 256, 2, True)), 16)
sp = ScsSelfPlay(cfg, search, games, device=0)
t0 = time.perf_counter()
r = sp.play_native(net, range(games), max_moves=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"expansions_per_s": r["expansions"] / dt, "seconds": dt}
import ctypes  # noqa: E402
import numpy as np  # noqa: E402
from nuzero_amd._lib import lib  # noqa: E402
st = np.zeros(8, np.uint64)
if lib.nz_boardnet_wide_stamps(ctypes.c_void_p(st.ctypes.data)) == 0 and st[5]:
    names = ["loads issued", "first fragments", "MFMAs + staging", "barrier", "loop overhead"]
    out["conv_wide_ticks_per_step"] = {n: float(st[i]) / float(st[5]) for i, n in enumerate(names)}
    out["conv_wide_steps_per_launch"] = float(st[5]) / float(st[6])
print(json.dumps(out))
