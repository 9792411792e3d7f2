"""oracle/scs.py against the SCS golden vectors made from the genuine reference
(tests/golden/make_golden_scs.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle.scs import ScsConfig, ScsGame

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {
    "mirrored5": None,    # the reference's own mirrored_config_5.yml: restated below as data
    "late_reinf": os.path.join(GOLDEN, "scs_configs", "late_reinforcements_5x5.yml"),
    "ten_by_ten": os.path.join(GOLDEN, "scs_configs", "ten_by_ten.yml"),
    "two_types": os.path.join(GOLDEN, "scs_configs", "two_types_6x5.yml"),
}


@pytest.fixture(scope="module")
def kat():
    return dict(np.load(os.path.join(GOLDEN, "scs_kat.npz")))


def checksum_weights(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) % 1000003).astype(np.float64) / 1000003.0


@pytest.mark.parametrize("name", ["late_reinf", "ten_by_ten", "two_types", "mirrored5"])
def test_scs_rules_against_reference(kat, name):
    path = CONFIGS[name] or os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsConfig(path)
    planes, rows, cols, channels, stacking, turns = kat[f"{name}_shape"]
    assert (cfg.planes, cfg.rows, cfg.cols, cfg.channels, cfg.stacking, cfg.turns) == (planes, rows, cols, channels,
                                                                                       stacking, turns)
    game_ids, actions = kat[f"{name}_game"], kat[f"{name}_action"]
    legal, n_legal = kat[f"{name}_legal"], kat[f"{name}_n_legal"]
    images = {int(s): img for s, img in zip(kat[f"{name}_image_step"], kat[f"{name}_images"])}
    w = checksum_weights(channels * rows * cols)
    pos = 0
    g, last = None, -1
    n_images = 0
    for i in range(len(actions)):
        if game_ids[i] != last:
            if g is not None:
                _check_end(kat, name, last, g, images)
            g, last = ScsGame(cfg), int(game_ids[i])
        assert not g.is_terminal()
        assert (g.player, g.sub_phase, g.stage, g.turn) == (kat[f"{name}_player"][i], kat[f"{name}_sub_phase"][i],
                                                           kat[f"{name}_stage"][i], kat[f"{name}_turn"][i]), i
        mask = g.possible_actions()
        assert mask.dtype == np.int8 and mask.shape == (planes, rows, cols)
        idx = np.nonzero(mask.flatten())[0]
        assert idx.tolist() == legal[pos:pos + n_legal[i]].tolist(), i
        pos += n_legal[i]
        img = g.state_image()
        assert img.dtype == np.float32 and img.shape == (1, channels, rows, cols)
        assert float(np.sum(img.reshape(-1).astype(np.float64) * w)) == kat[f"{name}_checksum"][i], i
        if i in images:
            assert np.array_equal(img[0], images[i]), i
            n_images += 1
        g.step_index(int(actions[i]))
    _check_end(kat, name, last, g, images)
    assert n_images > 20


def _check_end(kat, name, gi, g, images):
    assert g.is_terminal()
    assert g.get_length() == kat[f"{name}_lengths"][gi] and g.get_terminal_value() == kat[f"{name}_values"][gi]
    assert np.array_equal(g.state_image()[0], images[-(gi + 1)])


def test_scs_search_against_reference():
    """oracle/search.py on oracle/scs.py vs MCTS games the genuine reference played on SCS
    (float32 priors, float64 after root noise, players 0/1 so Q is never negated)."""
    import gzip
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from scs_eval import evaluate_image
    from oracle import search as osearch
    with gzip.open(os.path.join(GOLDEN, "scs_search_kat.json.gz"), "rt") as f:
        kat = json.load(f)
    files = {"mirrored_config_5.yml": "mirrored_5x5.yml"}
    for name, case in kat.items():
        cfg = ScsConfig(os.path.join(GOLDEN, "scs_configs", files.get(case["config_file"], case["config_file"])))
        ev = lambda game: evaluate_image(game.state_image()[0], cfg.num_actions)
        for ref in case["games"]:
            game = ScsGame(cfg)
            trace = []
            osearch.play_game(game, ev, case["config"], np.random.RandomState(ref["seed"]),
                              training=case["training"], trace=trace)
            assert game.length == ref["length"] and game.terminal_value == ref["terminal_value"], name
            assert len(trace) == len(ref["moves"])
            for mine, theirs in zip(trace, ref["moves"]):
                for key in ("action", "root_visits", "root_value_sum", "child_actions", "child_visits",
                            "child_priors", "child_value_sums"):
                    assert mine[key] == theirs[key], (name, key)


def test_randomized_maps_against_reference():
    """"Randomized" maps and victory points (SCS_Game.py:1678-1738): what the genuine reference builds after
    np.random.seed(seed) (tests/golden/scs_random_maps.json, made by make_golden_scs_random.py) equals what the oracle
    and the product's host-side config build from map_seed=seed, for both of the reference's randomized configs."""
    import json
    from nuzero_amd.scs import ScsGameConfig
    from oracle.scs import ScsConfig
    with open(os.path.join(GOLDEN, "scs_random_maps.json")) as f:
        gold = json.load(f)
    assert set(gold) == {"randomized_5x5", "randomized_10x10"}
    for name, cases in gold.items():
        path = os.path.join(GOLDEN, "scs_configs", name + ".yml")
        with pytest.raises(Exception):
            ScsGameConfig(path)                               # no seed: refused, not guessed
        seen = set()
        for seed, want in cases.items():
            oc = ScsConfig(path, map_seed=int(seed))
            assert [[list(map(float, t)) for t in row] for row in oc.terrain] == want["terrain"], (name, seed)
            assert [list(map(list, side)) for side in oc.vp] == want["vp"], (name, seed)
            pc = ScsGameConfig(path, map_seed=int(seed))
            assert np.array_equal(pc.terrain.reshape(oc.rows, oc.cols, 3), np.array(want["terrain"], np.float32))
            assert pc.vp.tolist() == [list(p) for side in want["vp"] for p in side]
            seen.add(json.dumps(want["terrain"]))
        assert len(seen) == len(cases)                        # the seeds give different maps


def test_games_on_their_own_randomized_maps_against_reference():
    """A new "Randomized" map per game (Gamer.py:52 builds a game object per game; SCS_Game.py:1678-1738 draws its map
    at construction): 24 games the genuine SCS_Game played, each after np.random.seed(5000 + i) (scs_pergame_kat.npz,
    tests/golden/make_golden_scs_pergame.py) -- the oracle on the map it draws from the same seed, step by step."""
    kat = dict(np.load(os.path.join(GOLDEN, "scs_pergame_kat.npz")))
    path = os.path.join(GOLDEN, "scs_configs", "randomized_5x5.yml")
    planes, rows, cols, channels, stacking, turns = kat["shape"]
    w = checksum_weights(channels * rows * cols)
    images = {int(s): img for s, img in zip(kat["image_step"], kat["images"])}
    game_ids, actions, legal, n_legal = kat["game"], kat["action"], kat["legal"], kat["n_legal"]
    pos, g, last, n_images = 0, None, -1, 0

    def end(gi, g):
        assert g.is_terminal() and g.get_length() == kat["lengths"][gi] and g.get_terminal_value() == kat["values"][gi]
        assert np.array_equal(g.state_image()[0], images[-(gi + 1)])

    for i in range(len(actions)):
        if game_ids[i] != last:
            if g is not None:
                end(last, g)
            last = int(game_ids[i])
            cfg = ScsConfig(path, map_seed=int(kat["map_seed"][last]))
            assert np.array_equal(np.array(cfg.terrain, np.float64), kat["terrain"][last])
            assert [[list(p) for p in side] for side in cfg.vp] == kat["vp"][last].tolist()
            g = ScsGame(cfg)
        assert (g.player, g.sub_phase, g.stage, g.turn) == (kat["player"][i], kat["sub_phase"][i], kat["stage"][i], kat["turn"][i]), i
        idx = np.nonzero(g.possible_actions().flatten())[0]
        assert idx.tolist() == legal[pos:pos + n_legal[i]].tolist(), i
        pos += n_legal[i]
        img = g.state_image()
        assert float(np.sum(img.reshape(-1).astype(np.float64) * w)) == kat["checksum"][i], i
        if i in images:
            assert np.array_equal(img[0], images[i]), i
            n_images += 1
        g.step_index(int(actions[i]))
    end(last, g)
    assert n_images > 100 and len(set(kat["terrain"].reshape(24, -1).sum(1).tolist())) > 10


def test_search_on_per_game_maps_against_reference():
    """MCTS self-play where every game has its own map and ONE stream (np.random.seed(s); SCS_Game(config); play): the
    oracle draws the map from RandomState(s) and its Explorer goes on with the same stream."""
    import gzip
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from scs_eval import evaluate_image
    from oracle import search as osearch
    with gzip.open(os.path.join(GOLDEN, "scs_search_pergame_kat.json.gz"), "rt") as f:
        kat = json.load(f)
    path = os.path.join(GOLDEN, "scs_configs", "randomized_5x5.yml")
    for name, case in kat.items():
        for ref in case["games"]:
            rs = np.random.RandomState(ref["seed"])
            cfg = ScsConfig(path, map_seed=rs)
            assert np.array_equal(np.array(cfg.terrain, np.float64), np.array(ref["terrain"]))
            ev = lambda game: evaluate_image(game.state_image()[0], cfg.num_actions)
            game = ScsGame(cfg)
            trace = []
            osearch.play_game(game, ev, case["config"], rs, training=case["training"], trace=trace)
            assert game.length == ref["length"] and game.terminal_value == ref["terminal_value"], name
            assert len(trace) == len(ref["moves"])
            for mine, theirs in zip(trace, ref["moves"]):
                for key in ("action", "root_visits", "root_value_sum", "child_actions", "child_visits", "child_priors",
                            "child_value_sums"):
                    assert mine[key] == theirs[key], (name, key)
