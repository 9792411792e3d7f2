"""Every SCS game on its OWN "Randomized" map -- what the reference's SCS presets train on (Run.py:115;
SCS_Game.py:1678-1738 draws terrain and victory points when a game object is built, Training/Gamer.py:52 builds one per
game) -- on the HIP path, against fixtures the genuine SCS_Game / Explorer made (tests/golden/make_golden_scs_pergame.py):
the device rules and the device search on per-game maps equal them bit for bit; 1024 different maps play in one engine.
Seeding rule: game with seed s = np.random.seed(s); SCS_Game(config); play -- one stream, the map's draws first.
Needs a GPU."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")
PATH = os.path.join(GOLDEN, "scs_configs", "randomized_5x5.yml")


def checksum_weights(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) % 1000003).astype(np.float64) / 1000003.0


def test_device_rules_on_per_game_maps_equal_the_reference():
    """24 games side by side in ONE batch, each on its own map (nz_scs_set_maps), replaying what the genuine SCS_Game
    played: turn machine registers, legal sets, state images of every step."""
    from nuzero_amd.scs import ScsBatch, ScsGameConfig
    kat = dict(np.load(os.path.join(GOLDEN, "scs_pergame_kat.npz")))
    cfg = ScsGameConfig(PATH, per_game=True)
    G = len(kat["lengths"])
    terrain, vp, _, _, _ = cfg.draw_games(kat["map_seed"].tolist())
    assert np.array_equal(terrain.reshape(G, cfg.rows, cfg.cols, 3).astype(np.float64), kat["terrain"])
    assert len({t.tobytes() for t in terrain}) == G                    # 24 different maps
    batch = ScsBatch(cfg, G)
    batch.set_maps(terrain, vp)
    planes, rows, cols, channels, _, _ = kat["shape"]
    w = checksum_weights(channels * rows * cols)
    first = np.concatenate([[0], np.cumsum(kat["lengths"])[:-1]])         # every game's first row in the step arrays
    legal_at = np.concatenate([[0], np.cumsum(kat["n_legal"])])
    images = {int(s): img for s, img in zip(kat["image_step"], kat["images"])}
    n_images = 0
    for m in range(int(kat["lengths"].max()) + 1):
        st = batch.status().cpu().numpy()
        mask = batch.legal_mask().cpu().numpy().reshape(G, -1)
        img = batch.state_image().cpu().numpy()
        actions = np.full(G, -1, np.int32)
        for g in range(G):
            if m >= kat["lengths"][g]:
                assert st[g, 4] == 1 and st[g, 5] == kat["values"][g] and st[g, 6] == kat["lengths"][g]
                assert np.array_equal(img[g], images[-(g + 1)])
                continue
            i = int(first[g]) + m
            assert tuple(st[g, :4]) == (kat["player"][i], kat["sub_phase"][i], kat["stage"][i], kat["turn"][i]), (g, m)
            assert np.nonzero(mask[g])[0].tolist() == kat["legal"][legal_at[i]:legal_at[i + 1]].tolist(), (g, m)
            assert float(np.sum(img[g].reshape(-1).astype(np.float64) * w)) == kat["checksum"][i], (g, m)
            if i in images:
                assert np.array_equal(img[g], images[i]), (g, m)
                n_images += 1
            actions[g] = kat["action"][i]
        if (actions < 0).all():
            break
        batch.step(actions)
    assert n_images > 100
    batch.close()


def test_device_search_on_per_game_maps_equals_the_reference():
    """MCTS self-play, every game on its own map and one stream per game (map first): the games the genuine Explorer
    played (scs_search_pergame_kat.json.gz) -- every root statistic of every move."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from test_gpu_scs import _host_evaluator
    with gzip.open(os.path.join(GOLDEN, "scs_search_pergame_kat.json.gz"), "rt") as f:
        kat = json.load(f)
    for name, case in kat.items():
        cfg = ScsGameConfig(PATH, per_game=True)
        games = case["games"]
        sp = ScsSelfPlay(cfg, case["config"], len(games), training=case["training"])
        r = sp.play(_host_evaluator(cfg.num_actions), [g["seed"] for g in games])
        for g, ref in enumerate(games):
            assert np.array_equal(sp.game_maps[0][g].reshape(cfg.rows, cfg.cols, 3), np.array(ref["terrain"], np.float32))
            assert r["lengths"][g] == ref["length"] and r["outcomes"][g] == ref["terminal_value"], (name, g)
            for m, mv in enumerate(ref["moves"]):
                k = len(mv["child_actions"])
                assert r["actions"][g, m] == mv["action"], (name, g, m)
                assert r["tree_size"][g, m] == mv["root_visits"] and r["n_children"][g, m] == k
                assert r["bias"][g, m] == mv["bias"] and r["root_value_sum"][g, m] == mv["root_value_sum"]
                assert r["child_action"][g, m, :k].tolist() == mv["child_actions"]
                assert r["child_visit"][g, m, :k].tolist() == mv["child_visits"], (name, g, m)
                assert r["child_prior"][g, m, :k].tolist() == mv["child_priors"], (name, g, m)
                assert r["child_value_sum"][g, m, :k].tolist() == mv["child_value_sums"]
        assert r["expansions"] == sum(g["evaluations"] for g in games)
        sp.close()


def _properties_on_own_maps(r, seeds, games):
    """Replay through the oracle rules, every game on the map the oracle draws from its own seed."""
    from oracle.scs import ScsConfig, ScsGame
    for g in games:
        og = ScsGame(ScsConfig(PATH, map_seed=np.random.RandomState(int(seeds[g]))))
        n = int(r["lengths"][g])
        for m in range(n):
            legal = np.nonzero(og.possible_actions().reshape(-1))[0]
            k = int(r["n_children"][g, m])
            assert r["child_action"][g, m, :k].tolist() == legal.tolist(), (g, m)
            assert int(r["child_visit"][g, m, :k].sum()) == int(r["tree_size"][g, m]) - 1, (g, m)
            og.step_index(int(r["actions"][g, m]))
        assert og.terminal and og.terminal_value == r["outcomes"][g], g


def test_1024_different_maps_in_one_engine_both_routes_and_refill():
    """The library's move loop with the native network, 1024 games on 1024 different maps: the persistent route and the
    wave-by-wave route play the same games (bit for bit), a sample replays legally through the oracle on its own maps, and
    a round of more games than trees (a tree's next game brings its own map) equals the same games one per tree."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from test_gpu_scs_configs import a1_search, _net, _same_games
    cfg = ScsGameConfig(PATH, per_game=True)
    G = 1024
    net, _ = _net(cfg, "convnet", 32, 3, seed=21, gain=2.0, max_batch=G)
    search = a1_search(24)
    seeds = list(range(9000, 9000 + G))
    sp = ScsSelfPlay(cfg, search, G)
    sp.persistent(1)
    rp = sp.play_native(net, seeds)
    assert sp.persistent() is True
    assert len({t.tobytes() for t in sp.game_maps[0]}) > 1000
    assert rp["simulations"] == 24 * int(rp["lengths"].sum())
    _properties_on_own_maps(rp, seeds, range(0, G, 64))
    sp.persistent(0)
    rw = sp.play_native(net, seeds)
    assert sp.persistent() is False
    _same_games(rp, rw, [(g, g) for g in range(G)], "per-game maps")
    sp.close()
    small = ScsSelfPlay(cfg, search, 96)
    rr = small.play_round(net, seeds[:300])
    _same_games(rr, rp, [(g, g) for g in range(300)], "refill on own maps")
    small.close(); net.close()


def test_gamer_gives_every_game_its_own_map_and_fills_the_buffers():
    """Gamer(..., SCS_Game, [randomized config], ...): game i of a round = np.random.seed(base_seed + i); SCS_Game(config);
    play.  A round of more games than trees, with the inference cache on (the reference's default presets); the host
    records (states regenerated on each game's own map) equal the oracle's images along the recorded actions, and the
    device replay buffer holds the same positions."""
    import torch
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.replay_device import DeviceReplayBuffer
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    from oracle.scs import ScsConfig, ScsGame
    from test_gpu_scs_configs import a1_search

    class SCS_Game:
        pass

    shapes = convnet_param_shapes(86, 21, 3, 32, 2)
    model = {k: torch.from_numpy(v) for k, v in synthetic_weights(5, shapes, 2.0).items()}
    nm = Network_Manager(model)
    host = ReplayBuffer(100, 16)
    N, base = 10, 7700
    g = Gamer(host, nm, SCS_Game, [PATH], 3, a1_search(16), 1, "keyless", size_estimate=4096, num_games=N,
              concurrent_games=4, base_seed=base)
    assert g.scs_config.per_game
    records, stats = g.play_games()
    assert len(records) == N and len(stats) == N and g.engine.persistent()
    maps = g.engine.game_maps[0]
    assert len({m.tobytes() for m in maps}) == N
    for i, rec in enumerate(records):
        og = ScsGame(ScsConfig(PATH, map_seed=np.random.RandomState(base + i)))
        assert np.array_equal(np.array(og.cfg.terrain, np.float32).reshape(-1, 3), maps[i])
        for m in range(rec.length):
            assert np.array_equal(rec.get_state_from_history(m).numpy(), og.state_image()), (i, m)
            pol = np.asarray(rec.make_target(m)[1])
            legal = og.possible_actions().reshape(-1) != 0
            assert abs(pol.sum() - 1.0) < 1e-9 and not pol[~legal].any()
            assert legal[rec.action_history[m]]
            og.step_index(rec.action_history[m])
        assert og.terminal and og.terminal_value == rec.terminal_value
    # the same round into a device buffer: same positions
    dev = DeviceReplayBuffer(100, 16, (86, 5, 5), 525, max_game_length=g.engine.MAX_MOVES)
    g2 = Gamer(dev, nm, SCS_Game, [PATH], 3, a1_search(16), 1, "keyless", size_estimate=4096, num_games=N,
               concurrent_games=4, base_seed=base, records=False)
    _, stats2 = g2.play_games()
    assert stats2 == stats
    assert dev.len() == host.len() == sum(rec.length for rec in records)
    a, b = dev.get_buffer(), host.get_buffer()
    for x, y in zip(a, b):
        assert torch.equal(x[0], y[0]) and x[1][0] == y[1][0] and x[2] == y[2]
        assert x[1][1] == np.asarray(y[1][1], np.float32).tolist()       # (the device buffer keeps policies as float32)
    dev.close(); g.engine.close(); g2.engine.close()
