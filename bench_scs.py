#!/usr/bin/env python3
"""First measurement of the SCS self-play path (BASELINE.json configs[3] shape: SCS 5x5 map,
200 sims/move, 1024 games, 1 GPU).  Not the driver's bench (that is bench.py): the SCS path is
lock-step with the network outside the search kernels -- tree, rules and legal masks on the
device (nz_scs_search_*), a PyTorch conv net (square 3x3 convs: hexagdly is not available, so
the hex form of the reference's ConvNet cannot run or be pinned) evaluating the leaf batch.

    python bench_scs.py [--games 1024] [--sims 200] [--filters 32] [--layers 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--filters", type=int, default=32)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--config", default=os.path.join(REPO, "tests", "golden", "scs_configs", "mirrored_5x5.yml"))
    args = ap.parse_args()
    import torch
    from nuzero_amd.scs import ScsGameConfig, ScsSelfPlay, torch_evaluator
    cfg = ScsGameConfig(args.config)

    class ConvNetSquare(torch.nn.Module):            # ConvNet(in, policy, 3, filters, layers, hex=False) trunk + 1-conv heads
        recurrent = False

        def __init__(self):
            super().__init__()
            layers, c = [], cfg.channels
            for _ in range(args.layers + 1):
                layers += [torch.nn.Conv2d(c, args.filters, 3, padding="same", bias=False), torch.nn.ELU()]
                c = args.filters
            self.trunk = torch.nn.Sequential(*layers)
            self.policy = torch.nn.Conv2d(c, cfg.planes, 3, padding="same", bias=False)
            self.value = torch.nn.Conv2d(c, 1, 3, padding="same", bias=False)

        def forward(self, x):
            t = self.trunk(x)
            return self.policy(t), torch.tanh(self.value(t).mean(dim=(1, 2, 3))).reshape(-1, 1)

    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True        # let MIOpen pick a solver for the (fixed) leaf-batch shape
    net = ConvNetSquare().cuda()
    search = {"Simulation": {"mcts_simulations": args.sims, "keep_subtree": True},
              "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}      # Configs/Search/a1_search_config.yaml
    sp = ScsSelfPlay(cfg, search, args.games, nodes_per_game=1 + args.sims * 40 * 128)
    t0 = time.perf_counter()
    ev = torch_evaluator(net, pad_to=args.games)
    ev(torch.zeros((1, cfg.channels, cfg.rows, cfg.cols), device="cuda"))     # solver search outside the timed region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = sp.play(ev, seeds=range(args.games))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "SCS %dx%d stack %d, %d sims/move, %d concurrent games, torch ConvNet(%d filters, %d layers, "
                                  "square convs) as evaluator" % (cfg.rows, cfg.cols, cfg.stacking, args.sims, args.games,
                                                                  args.filters, args.layers),
                      "games_per_s": args.games / dt, "expansions_per_s": r["expansions"] / dt,
                      "simulations_per_s": r["simulations"] / dt, "seconds": dt,
                      "mean_game_length": float(r["lengths"].mean()),
                      "outcomes": {str(v): int((r["outcomes"] == v).sum()) for v in (-1, 0, 1)}}))


if __name__ == "__main__":
    main()
