#!/usr/bin/env python3
"""Golden vectors for the SCS rules, from the GENUINE reference.

Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_scs.py

Imports /root/reference/Games/SCS/SCS_Game.py with the inert stand-ins of SURVEY.md appendix B
(termcolor, hexagdly, gymnasium.spaces, pettingzoo.AECEnv, pygame init functions, ray) and with
cwd = /root/reference (unit image paths are checked relative to it).  Plays seeded uniformly
random legal games on fully "Detailed" configurations (no map RNG) and records, per step, the
legal action set, the turn machine state, a float64 checksum of the 86-plane state image (the
full image every few steps), and the chosen action; per game the length and terminal value.

Configurations: the reference's mirrored_config_5
(read where they lie; only results are stored) and tests/golden/scs_configs/*.yml
(fixtures written for this repository; the reference's mirrored_plus_config_5 and mirrored_config_10
name a unit image that does not exist and do
not load at HEAD).  Output: tests/golden/scs_kat.npz.
"""
import os
import sys
import types

import numpy as np

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

CASES = [
    # name, config path, games, full image every k steps
    ("mirrored5", os.path.join(REF, "Games/SCS/Game_configs/mirrored_config_5.yml"), 40, 8),
    ("late_reinf", os.path.join(HERE, "scs_configs", "late_reinforcements_5x5.yml"), 30, 8),
    ("ten_by_ten", os.path.join(HERE, "scs_configs", "ten_by_ten.yml"), 10, 24),
    ("two_types", os.path.join(HERE, "scs_configs", "two_types_6x5.yml"), 40, 8),
]


def checksum_weights(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) % 1000003).astype(np.float64) / 1000003.0


def main():
    out = {}
    for name, path, n_games, every in CASES:
        rows = {k: [] for k in ("game", "player", "sub_phase", "stage", "turn", "action", "n_legal", "checksum")}
        legal = []
        images, image_step = [], []
        lengths, values = [], []
        w = None
        step_global = 0
        for gi in range(n_games):
            g = SCS_Game(path)
            rs = np.random.RandomState(1000 + gi)
            while not g.is_terminal():
                mask = g.possible_actions().flatten()
                idx = np.nonzero(mask)[0]
                img = g.generate_network_input().numpy()[0]
                if w is None:
                    w = checksum_weights(img.size)
                a = int(rs.choice(idx))
                rows["game"].append(gi)
                rows["player"].append(g.get_current_player())
                rows["sub_phase"].append(g.current_sub_phase)
                rows["stage"].append(g.current_stage)
                rows["turn"].append(g.current_turn)
                rows["action"].append(a)
                rows["n_legal"].append(len(idx))
                rows["checksum"].append(float(np.sum(img.reshape(-1).astype(np.float64) * w)))
                legal.extend(idx.tolist())
                if step_global % every == 0:
                    images.append(img.copy())
                    image_step.append(step_global)
                step_global += 1
                g.step(g.get_action_coords(a))
            lengths.append(g.get_length())
            values.append(g.get_terminal_value())
            # the terminal position's image too
            images.append(g.generate_network_input().numpy()[0].copy())
            image_step.append(-(gi + 1))
        out[f"{name}_shape"] = np.array([g.total_action_planes, g.rows, g.columns, g.total_dims, g.stacking_limit,
                                         g.turns], np.int32)
        for k in ("game", "player", "sub_phase", "stage", "turn", "action", "n_legal"):
            out[f"{name}_{k}"] = np.array(rows[k], np.int32)
        out[f"{name}_checksum"] = np.array(rows["checksum"], np.float64)
        out[f"{name}_legal"] = np.array(legal, np.int32)
        out[f"{name}_images"] = np.array(images, np.float32)
        out[f"{name}_image_step"] = np.array(image_step, np.int32)
        out[f"{name}_lengths"] = np.array(lengths, np.int32)
        out[f"{name}_values"] = np.array(values, np.int32)
        print(name, "games", n_games, "steps", step_global, "mean length", np.mean(lengths), "max legal",
              max(rows["n_legal"]), "outcomes", {v: values.count(v) for v in (-1, 0, 1)}, "images", len(images))
    np.savez_compressed(os.path.join(HERE, "scs_kat.npz"), **out)


if __name__ == "__main__":
    main()
