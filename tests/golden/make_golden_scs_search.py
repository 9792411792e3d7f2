#!/usr/bin/env python3
"""Golden vectors for MCTS self-play on SCS, from the GENUINE reference (Search/Explorer.py +
Games/SCS/SCS_Game.py; import recipe of make_golden_scs.py).  Leaf evaluations come from
tests/scs_eval.py through the reference's cache-hit branch (Explorer.py:146-149), so the
unmodified Explorer runs without a network.  The move loop is Training/Gamer.py:52-92.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_scs_search.py

Output: tests/golden/scs_search_kat.json.gz
"""
import gzip
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Space:
    def __init__(self, *a, **k):
        pass


_stub("termcolor", colored=lambda s, *a, **k: s)
_stub("hexagdly")
_g = _stub("gymnasium")
_g.spaces = _stub("gymnasium.spaces", Discrete=_Space, Box=_Space)
_stub("pettingzoo", AECEnv=object)
_pg = _stub("pygame")
for _sub in ("display", "fastevent", "font", "scrap"):
    setattr(_pg, _sub, _stub("pygame." + _sub, init=lambda *a, **k: None))
_stub("ray")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.chdir(REF)

from Games.SCS.SCS_Game import SCS_Game  # noqa: E402
from Search.Explorer import Explorer  # noqa: E402
from Search.Node import Node  # noqa: E402
from scs_eval import evaluate_image  # noqa: E402


class EvalCache:
    def __init__(self, num_actions):
        self.n = num_actions
        self.calls = 0

    def get(self, state):
        self.calls += 1
        return evaluate_image(state.numpy()[0], self.n)

    def put(self, item):
        raise AssertionError("always hits")


def cfg(sims, eps_s=0, eps_r=0, softmax_moves=0):
    return {"Simulation": {"mcts_simulations": sims, "keep_subtree": True},
            "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
            "Exploration": {"number_of_softmax_moves": softmax_moves, "epsilon_softmax_exploration": eps_s,
                            "epsilon_random_exploration": eps_r, "value_factor": 1,
                            "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                            "root_dist_alpha": 0.2, "root_dist_beta": 1}}


CASES = [
    ("mirrored5_s20", os.path.join(REF, "Games/SCS/Game_configs/mirrored_config_5.yml"), cfg(20), True, [0, 1, 2, 3]),
    ("two_types_s16", os.path.join(HERE, "scs_configs", "two_types_6x5.yml"), cfg(16, eps_s=0.1, eps_r=0.1), True,
     [10, 11, 12]),
    ("late_reinf_eval12", os.path.join(HERE, "scs_configs", "late_reinforcements_5x5.yml"), cfg(12), False, [20]),
]


def play(path, config, training, seed):
    np.random.seed(seed)
    game = SCS_Game(path)
    cache = EvalCache(game.get_num_actions())
    explorer = Explorer(config, training)
    root = Node(0)
    moves = []
    while not game.is_terminal():
        action, chosen, bias = explorer.run_mcts(game, None, root, 2, cache)
        kids = root.children
        moves.append({"action": int(action), "root_visits": int(root.visit_count),
                      "root_value_sum": float(root.value_sum), "bias": float(bias),
                      "child_actions": [int(a) for a in kids],
                      "child_visits": [int(c.visit_count) for c in kids.values()],
                      "child_priors": [float(c.prior) for c in kids.values()],
                      "child_prior_is_f32": [bool(isinstance(c.prior, np.float32)) for c in kids.values()],
                      "child_value_sums": [float(c.value_sum) for c in kids.values()]})
        game.step(game.get_action_coords(action))
        game.store_search_statistics(root)
        root = chosen
    return {"seed": seed, "length": int(game.length), "terminal_value": int(game.terminal_value), "moves": moves,
            "evaluations": cache.calls}


def main():
    out = {}
    for name, path, config, training, seeds in CASES:
        games = [play(path, config, training, s) for s in seeds]
        out[name] = {"config": config, "training": training, "games": games,
                     "config_file": os.path.basename(path)}
        print(name, [(g["length"], g["terminal_value"], g["evaluations"]) for g in games])
    with gzip.open(os.path.join(HERE, "scs_search_kat.json.gz"), "wt", compresslevel=9) as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
