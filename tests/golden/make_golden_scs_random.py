#!/usr/bin/env python3
"""Golden vectors for "Randomized" SCS maps and victory points, from the GENUINE reference.

Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_scs_random.py

SCS_Game.load_game_from_config (Games/SCS/SCS_Game.py:1683-1738) draws a "Randomized" map tile by tile with
np.random.choice(terrain_types, p=distribution) and the victory points with np.random.choice(range(...)) from numpy's
GLOBAL stream.  This script seeds that stream, builds the reference's game and records what it drew: per (config, seed)
the terrain of every tile as (attack modifier, defense modifier, cost) and the two victory-point lists.  The configs are
the reference's randomized_config_5.yml / randomized_config_10.yml, stored here as data (parsed and re-emitted) under
tests/golden/scs_configs/.  Output: tests/golden/scs_random_maps.json.  Same import stand-ins as make_golden_scs.py.
"""
import json
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
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
os.chdir(REF)

from Games.SCS.SCS_Game import SCS_Game  # noqa: E402

CASES = [("randomized_5x5", "Games/SCS/Game_configs/randomized_config_5.yml"),
         ("randomized_10x10", "Games/SCS/Game_configs/randomized_config_10.yml")]
SEEDS = [0, 1, 2, 7, 12345]

out = {}
for name, rel in CASES:
    src = os.path.join(REF, rel)
    with open(src) as f:
        data = yaml.safe_load(f)
    with open(os.path.join(HERE, "scs_configs", name + ".yml"), "w") as f:          # the configuration, as data
        yaml.safe_dump(data, f, sort_keys=False)
    out[name] = {}
    for seed in SEEDS:
        np.random.seed(seed)
        g = SCS_Game(src)
        terrain = [[[float(t.terrain.attack_modifier), float(t.terrain.defense_modifier), float(t.terrain.cost)]
                    for t in row] for row in g.board]
        out[name][str(seed)] = {"terrain": terrain,
                                "vp": [[list(map(int, p)) for p in side] for side in g.victory_points]}
with open(os.path.join(HERE, "scs_random_maps.json"), "w") as f:
    json.dump(out, f)
print({k: list(v) for k, v in out.items()})
