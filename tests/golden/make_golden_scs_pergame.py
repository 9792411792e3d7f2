#!/usr/bin/env python3
"""Golden vectors for SCS games that each have their OWN "Randomized" map (what the reference's SCS presets train on:
Run.py:115 and most others use randomized_config_5.yml; SCS_Game.load_game_from_config draws terrain and victory points
from numpy's global stream when the game object is built, SCS_Game.py:1678-1738, and Gamer builds a new game object per
game, Training/Gamer.py:52).  From the GENUINE reference (import recipe of make_golden_scs.py).

Seeding rule of the harness (the reference never seeds): game with seed s = `np.random.seed(s)`, then `SCS_Game(config)`
(the map draws), then everything the game's play draws -- ONE stream per game, map first.

  scs_pergame_kat.npz             random play on N games, each on its own map (every step: turn machine registers, legal
                                  set, image checksum; images every few steps); the action choices come from a SEPARATE
                                  RandomState so that the map stream is only consumed by the map
  scs_search_pergame_kat.json.gz  MCTS self-play (Explorer + the Gamer loop, Training/Gamer.py:52-92) on 6 such games,
                                  leaf evaluations from tests/scs_eval.py through the cache-hit branch

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_scs_pergame.py
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

CONFIG = os.path.join(REF, "Games/SCS/Game_configs/randomized_config_5.yml")    # = tests/golden/scs_configs/randomized_5x5.yml


def game_map(g):
    terrain = [[[float(t.terrain.attack_modifier), float(t.terrain.defense_modifier), float(t.terrain.cost)]
                for t in row] for row in g.board]
    return terrain, [[list(map(int, p)) for p in side] for side in g.victory_points]


def checksum_weights(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) % 1000003).astype(np.float64) / 1000003.0


def rules_kat(n_games=24, every=6, seed0=5000):
    rows = {k: [] for k in ("game", "player", "sub_phase", "stage", "turn", "action", "n_legal", "checksum")}
    legal, images, image_step, lengths, values, terrains, vps = [], [], [], [], [], [], []
    w = None
    step_global = 0
    for gi in range(n_games):
        np.random.seed(seed0 + gi)
        g = SCS_Game(CONFIG)
        t, v = game_map(g)
        terrains.append(t)
        vps.append(v)
        rs = np.random.RandomState(1000 + gi)              # the action choices: not the game's stream
        while not g.is_terminal():
            mask = g.possible_actions().flatten()
            idx = np.nonzero(mask)[0]
            img = g.generate_network_input().numpy()[0]
            if w is None:
                w = checksum_weights(img.size)
            a = int(rs.choice(idx))
            for k, val in (("game", gi), ("player", g.get_current_player()), ("sub_phase", g.current_sub_phase),
                           ("stage", g.current_stage), ("turn", g.current_turn), ("action", a), ("n_legal", len(idx)),
                           ("checksum", float(np.sum(img.reshape(-1).astype(np.float64) * w)))):
                rows[k].append(val)
            legal.extend(idx.tolist())
            if step_global % every == 0:
                images.append(img.copy())
                image_step.append(step_global)
            step_global += 1
            g.step(g.get_action_coords(a))
        lengths.append(g.get_length())
        values.append(g.get_terminal_value())
        images.append(g.generate_network_input().numpy()[0].copy())
        image_step.append(-(gi + 1))
    out = {"shape": np.array([g.total_action_planes, g.rows, g.columns, g.total_dims, g.stacking_limit, g.turns], np.int32),
           "map_seed": np.arange(seed0, seed0 + n_games, dtype=np.int64),
           "terrain": np.array(terrains, np.float64), "vp": np.array(vps, np.int32)}
    for k in ("game", "player", "sub_phase", "stage", "turn", "action", "n_legal"):
        out[k] = np.array(rows[k], np.int32)
    out["checksum"] = np.array(rows["checksum"], np.float64)
    out["legal"] = np.array(legal, np.int32)
    out["images"] = np.array(images, np.float32)
    out["image_step"] = np.array(image_step, np.int32)
    out["lengths"] = np.array(lengths, np.int32)
    out["values"] = np.array(values, np.int32)
    np.savez_compressed(os.path.join(HERE, "scs_pergame_kat.npz"), **out)
    print("rules: games", n_games, "steps", step_global, "distinct maps", len({json.dumps(t) for t in terrains}),
          "outcomes", {v: values.count(v) for v in (-1, 0, 1)})


class EvalCache:
    def __init__(self, num_actions):
        self.n = num_actions
        self.calls = 0

    def get(self, state):
        self.calls += 1
        return evaluate_image(state.numpy()[0], self.n)

    def put(self, item):
        raise AssertionError("always hits")


def search_cfg(sims, eps_s=0, eps_r=0, softmax_moves=0):
    return {"Simulation": {"mcts_simulations": sims, "keep_subtree": True},
            "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
            "Exploration": {"number_of_softmax_moves": softmax_moves, "epsilon_softmax_exploration": eps_s,
                            "epsilon_random_exploration": eps_r, "value_factor": 1,
                            "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                            "root_dist_alpha": 0.2, "root_dist_beta": 1}}


def play(config, training, seed):
    np.random.seed(seed)                                   # the game's one stream: map first, then the search's draws
    game = SCS_Game(CONFIG)
    terrain, vp = game_map(game)
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
                      "child_value_sums": [float(c.value_sum) for c in kids.values()]})
        game.step(game.get_action_coords(action))
        game.store_search_statistics(root)
        root = chosen
    return {"seed": seed, "length": int(game.length), "terminal_value": int(game.terminal_value), "moves": moves,
            "evaluations": cache.calls, "terrain": terrain, "vp": vp}


def search_kat():
    out = {}
    for name, config, training, seeds in (("randomized5_s16", search_cfg(16, eps_s=0.1, eps_r=0.05), True, [40, 41, 42, 43]),
                                          ("randomized5_eval12", search_cfg(12), False, [50, 51])):
        games = [play(config, training, s) for s in seeds]
        out[name] = {"config": config, "training": training, "games": games, "config_file": "randomized_5x5.yml"}
        print(name, [(g["length"], g["terminal_value"], g["evaluations"]) for g in games])
    with gzip.open(os.path.join(HERE, "scs_search_pergame_kat.json.gz"), "wt", compresslevel=9) as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    rules_kat()
    search_kat()
