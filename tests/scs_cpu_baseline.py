#!/usr/bin/env python3
"""CPU baseline for the SCS self-play path: the oracle (oracle/scs.py rules + oracle/search.py Explorer +
oracle/net.py ConvNet, the restatement of the reference's path) on a bounded sample of bench_scs.py's default
workload -- moves of one self-play game until the time is up, one process, one torch thread.  Lives under tests/
because only test code may use the oracle.

    python tests/scs_cpu_baseline.py [--seconds 15] [--sims 200] [--filters 32] [--layers 8] [--config ...yml]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def cpu_baseline(config_path, weights, layers, search, seconds):
    """The CPU oracle (oracle/scs.py rules + oracle/search.py Explorer + oracle/net.py ConvNet, the
    restatement of the reference's path) on a bounded sample: moves of one self-play game until
    `seconds` have passed, one process, one torch thread."""
    import torch
    from oracle import search as osearch
    from oracle.net import FeedForwardRef
    from oracle.scs import ScsConfig, ScsGame
    torch.set_num_threads(1)
    net = FeedForwardRef(weights, "convnet", layers)
    ev = osearch.net_evaluator(net, None)
    game = ScsGame(ScsConfig(config_path))
    explorer = osearch.Explorer(search, True, np.random.RandomState(0))
    root = osearch.Node(0)
    t0 = time.perf_counter()
    moves = 0
    while not game.is_terminal() and time.perf_counter() - t0 < seconds:
        action, chosen, _ = explorer.run_mcts(game, ev, root)
        game.step_index(action)
        root = chosen
        moves += 1
    dt = time.perf_counter() - t0
    return {"expansions_per_s": explorer.counters.expansions / dt, "simulations_per_s": explorer.counters.simulations / dt,
            "moves_per_s": moves / dt, "cores": 1, "kind": "port",
            "sample": "%d moves of one game (%d simulations) in %.1f s" % (moves, explorer.counters.simulations, dt)}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--filters", type=int, default=32)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--config", default=os.path.join(REPO, "tests", "golden", "scs_configs", "mirrored_5x5.yml"))
    args = ap.parse_args()
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    from oracle.scs import ScsConfig
    cfg = ScsConfig(args.config)
    w = synthetic_weights(0, convnet_param_shapes(cfg.channels, cfg.planes, 3, args.filters, args.layers))
    search = {"Simulation": {"mcts_simulations": args.sims, "keep_subtree": True},
              "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}      # Configs/Search/a1_search_config.yaml
    print(json.dumps(cpu_baseline(args.config, w, args.layers, search, args.seconds)))


if __name__ == "__main__":
    main()
