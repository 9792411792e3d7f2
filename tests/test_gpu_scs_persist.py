"""The persistent SCS self-play kernel (nz_scs_search_persistent: one wavefront per game, a move's whole search --
descent, rules, the network for its own leaf, expansion, backup -- in one launch) against
  * the CPU oracle (oracle/search.py + oracle/scs.py, pinned to the genuine reference) replaying the same games with
    the leaf evaluations the kernel ITSELF recorded (digest of the planes, probabilities, value per expansion): every
    action, visit count, float32 / float64 prior and value sum bit-identical;
  * the oracle network on the recorded leaves: probabilities and value within 1e-5 (tolerance of BASELINE.json's
    north_star);
  * the wave-by-wave route (wave_kernel + one network launch per simulation wave): the same games, bit for bit.
Needs a GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")

from test_gpu_scs_configs import a1_search, _net, _same_games, _game_properties   # noqa: E402


def _replay_recorded(path, search, seeds, r, recs, games, label, training=True):
    from scs_replay import replay_games, assert_trace_equals_device
    jobs = [(path, search, seeds[g], training) + recs[g] + (None,) for g in games]
    outs = replay_games(jobs)
    moves = 0
    for g, out in zip(games, outs):
        assert out["evaluations_used"] == out["evaluations_recorded"], (label, g)
        assert out["length"] == r["lengths"][g] and out["terminal"], (label, g)
        assert out["terminal_value"] == r["outcomes"][g], (label, g)
        moves += assert_trace_equals_device(r, g, out, label)
    return moves


def _oracle_net_on_leaves(path, w, arch, depth, hexnet, r, g, recs, n_check):
    """Replay game g through the oracle rules and compare the recorded evaluation of each root position (the first
    evaluation of a move whose root is not yet expanded is the root itself: only move 0; so instead every recorded
    leaf is matched by digest against the positions along the game) -- simpler and sufficient: the root positions of
    the game are leaves of the previous move's search, so each is looked up by digest among the recorded leaves."""
    from scipy.special import softmax
    from oracle.net import FeedForwardRef, HexNetRef
    from oracle.scs import ScsConfig, ScsGame
    from scs_replay import image_mix_digest
    ref = HexNetRef(w, arch, depth) if hexnet else FeedForwardRef(w, arch, depth)
    dig, probs, values = recs[g]
    index = {(int(d[0]), int(d[1])): i for i, d in enumerate(dig)}
    og = ScsGame(ScsConfig(path))
    worst_p = worst_v = 0.0
    checked = 0
    for m in range(int(r["lengths"][g])):
        if checked < n_check:
            img = og.state_image()
            d = image_mix_digest(img[0])
            i = index.get((int(d[0]), int(d[1])))
            if i is not None:
                p, v = ref.inference(img, None)
                worst_p = max(worst_p, float(np.max(np.abs(softmax(p.reshape(-1)) - probs[i]))))
                worst_v = max(worst_v, abs(float(v.reshape(-1)[0]) - float(values[i])))
                checked += 1
        og.step_index(int(r["actions"][g, m]))
    assert checked >= min(n_check, 3), checked
    assert worst_p < 1e-5 and worst_v < 1e-5, (worst_p, worst_v)
    return checked, worst_p, worst_v


@pytest.mark.parametrize("arch,hexnet", [("convnet", False), ("resnet", False), ("convnet", True)])
def test_persistent_route_equals_the_oracle_replay_and_the_wave_route(arch, hexnet):
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "late_reinforcements_5x5.yml" if arch == "convnet" else "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G = 12
    depth = 3 if arch == "convnet" else 2
    net, w = _net(cfg, arch, 32, depth, seed=43, gain=2.5, hexnet=hexnet, max_batch=G)
    search = a1_search(48, number_of_softmax_moves=4, epsilon_softmax_exploration=0.1, epsilon_random_exploration=0.05,
                       root_exploration_fraction=0.25, root_dist_alpha=0.3)
    seeds = list(range(1900, 1900 + G))
    sp = ScsSelfPlay(cfg, search, G)
    sp.persistent(1)                                   # a play fails if the route is not available
    sp.record(range(G), 48 * (sp.MAX_MOVES + 1))
    rp = sp.play_native(net, seeds)
    assert sp.persistent() is True
    recs = sp.records()
    assert (rp["lengths"] > 10).all()
    assert rp["simulations"] == 48 * int(rp["lengths"].sum())
    assert rp["expansions"] == sum(len(v[2]) for v in recs.values())
    _game_properties(path, rp, range(G))
    moves = _replay_recorded(path, search, seeds, rp, recs, range(G), arch)
    assert moves == int(rp["lengths"].sum())
    for g in (0, G - 1):
        _oracle_net_on_leaves(path, w, arch, depth, hexnet, rp, g, recs, 12)
    # the wave-by-wave route plays the same games
    sp.record([], 0)
    sp.persistent(0)
    rw = sp.play_native(net, seeds)
    assert sp.persistent() is False
    _same_games(rp, rw, [(g, g) for g in range(G)], arch)
    assert rw["expansions"] == rp["expansions"] and rw["simulations"] == rp["simulations"]
    sp.close(); net.close()


@pytest.mark.parametrize("name,sims,softmax_moves", [("many_units_5x6", 16, 30)])
def test_persistent_route_with_more_than_64_children(name, sims, softmax_moves):
    """Positions with 70-100 legal actions on a board small enough for the persistent kernel (19 units against 4 on
    5 x 6 cells: the kernel variant that holds a node's children in chunks of 64 lanes): the persistent route
    replays exactly on the oracle from its own recorded evaluations, and plays the games of the wave-by-wave route."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", name + ".yml")
    cfg = ScsGameConfig(path)
    G = 6
    net, w = _net(cfg, "convnet", 32, 2, seed=47, gain=2.5, max_batch=G)
    search = a1_search(sims, number_of_softmax_moves=softmax_moves, epsilon_softmax_exploration=0.1,
                       epsilon_random_exploration=0.05, root_exploration_fraction=0.25, root_dist_alpha=0.3)
    seeds = list(range(2300, 2300 + G))
    sp = ScsSelfPlay(cfg, search, G)
    assert sp.MAX_CHILDREN > 64
    sp.persistent(1)
    sp.record(range(G), sims * (sp.MAX_MOVES + 1))
    rp = sp.play_native(net, seeds)
    assert sp.persistent() is True
    assert int(rp["n_children"].max()) > 64
    recs = sp.records()
    _game_properties(path, rp, range(G))
    moves = _replay_recorded(path, search, seeds, rp, recs, range(G), name)
    assert moves == int(rp["lengths"].sum())
    sp.record([], 0)
    sp.persistent(0)
    rw = sp.play_native(net, seeds)
    assert sp.persistent() is False
    _same_games(rp, rw, [(g, g) for g in range(G)], name)
    sp.close(); net.close()


def test_persistent_route_on_baseline_config_4():
    """BASELINE.json configs[3] at full size on the persistent route: 1024 games x 200 simulations per move, ConvNet(32
    filters x 8 layers).  Properties of all 1024 games through the oracle rules; exact oracle replay of 4 games on the
    evaluations the kernel recorded; the oracle network on recorded leaves within 1e-5."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G = 1024
    net, w = _net(cfg, "convnet", 32, 8, seed=16, gain=2.0, max_batch=G)
    search = a1_search(200)
    seeds = list(range(4000, 4000 + G))
    sample = [0, 341, 682, 1023]
    sp = ScsSelfPlay(cfg, search, G)
    sp.persistent(1)
    sp.record(sample, 200 * (sp.MAX_MOVES + 1))
    rp = sp.play_native(net, seeds)
    assert sp.persistent() is True
    assert rp["simulations"] == 200 * int(rp["lengths"].sum())
    _game_properties(path, rp, range(G))
    recs = sp.records()
    moves = _replay_recorded(path, search, seeds, rp, recs, sample, "config4-persistent")
    assert moves == int(rp["lengths"][sample].sum())
    _oracle_net_on_leaves(path, w, "convnet", 8, False, rp, sample[1], recs, 16)
    sp.close(); net.close()


def test_persistent_route_says_why_it_is_not_available():
    from nuzero_amd._lib import NzError
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    net, _ = _net(cfg, "recurrent", 32, 1, seed=3, gain=1.0, iters=2, max_batch=4)
    sp = ScsSelfPlay(cfg, a1_search(8), 4)
    sp.persistent(1)
    with pytest.raises(NzError, match="feed-forward"):
        sp.play_native(net, [1, 2, 3, 4], max_moves=1)
    sp.persistent(-1)
    sp.play_native(net, [1, 2, 3, 4], max_moves=1)      # the default falls back to the wave-by-wave route
    assert sp.persistent() is False
    sp.close(); net.close()


@pytest.mark.parametrize("arch", ["convnet", "resnet"])
def test_heads_side_by_side_play_the_same_games_as_layer_by_layer(arch, monkeypatch):
    """The persistent route runs the policy head on a game's leader wavefront and the value head on its helper, side by
    side (boardnet_wave_program's solo chains); NZ_SCS_PERSIST_NO_SOLO keeps every head layer split between the two with
    a meeting after each.  Same arithmetic either way: the same games, bit for bit."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G = 8
    search = a1_search(32, number_of_softmax_moves=3, epsilon_softmax_exploration=0.1, root_exploration_fraction=0.25,
                       root_dist_alpha=0.3)
    seeds = list(range(700, 700 + G))
    out = []
    for no_solo in (False, True):
        if no_solo:
            monkeypatch.setenv("NZ_SCS_PERSIST_NO_SOLO", "1")      # read when a net's per-wavefront program is built
        net, _ = _net(cfg, arch, 32, 2, seed=5, gain=2.0, max_batch=G)
        sp = ScsSelfPlay(cfg, search, G)
        sp.persistent(1)
        out.append(sp.play_native(net, seeds, max_moves=12))
        assert sp.persistent() is True
        sp.close(); net.close()
    _same_games(out[0], out[1], [(g, g) for g in range(G)], arch)
    assert out[0]["expansions"] == out[1]["expansions"] and out[0]["simulations"] == out[1]["simulations"]
