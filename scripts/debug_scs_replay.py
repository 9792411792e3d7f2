import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nuzero_amd.boardnet import BoardNet
from nuzero_amd.replay_buffer import ReplayBuffer
from nuzero_amd.replay_device import DeviceReplayBuffer
from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig, scs_game_records
from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
path = "tests/golden/scs_configs/mirrored_5x5.yml"
cfg = ScsGameConfig(path)
net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=6)
net.set_weights(synthetic_weights(4, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 2), 2.0))
search = {"Simulation": {"mcts_simulations": 12, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
          "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04, "epsilon_random_exploration": 0.001, "value_factor": 1,
                          "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2, "root_dist_alpha": 0.2, "root_dist_beta": 1}}
sp = ScsSelfPlay(cfg, search, 6)
r = sp.play_native(net, range(20, 26))
print("lengths", r["lengths"])
host = ReplayBuffer(40, 8)
recs = scs_game_records(sp, r)
for rec in recs: host.save_game(rec, 1)
dev = DeviceReplayBuffer(40, 8, (cfg.channels, cfg.rows, cfg.cols), cfg.num_actions, max_game_length=sp.MAX_MOVES)
dev.save_scs_games(sp, sp.export_device(), 1)
dev.check()
got, want = dev.get_buffer(), host.get_buffer()
print(len(got), len(want))
bad = 0
for i, ((s1, (v1, p1), g1), (s2, (v2, p2), g2)) in enumerate(zip(got, want)):
    if not torch.equal(s1, s2):
        d = (s1 != s2).nonzero()
        if bad < 5: print("pos", i, "n diff", len(d), d[:4].tolist(), s1[tuple(d[0])].item(), s2[tuple(d[0])].item())
        bad += 1
    if not np.array_equal(np.asarray(p1, np.float32), torch.tensor(p2).numpy()): print("policy differs", i)
    if v1 != v2: print("value", i, v1, v2)
print("bad states", bad)
