"""BASELINE.json configs[3] and configs[4] on the HIP path, and the composed exactness of SCS search + native
network: the device search (tree, rules, masks, network all on the GPU, through the C ABI) against the CPU oracle
(oracle/search.py + oracle/scs.py, both pinned to the genuine reference by tests/test_scs_oracle.py) replaying the
same games with the DEVICE network's own leaf evaluations -- every action, visit count, float32/float64 prior and
value sum of every move bit-identical.  The network itself is held to 1e-5 against the oracle nets / the reference's
outputs in tests/test_gpu_boardnet.py (cases K and L of net_kat3.npz are the two configs' networks); the two
statements compose to the end-to-end one, as for Tic-Tac-Toe (tests/test_gpu_parity.py).

hex=True runs are labelled PARITY UNPINNED where they appear: hexagdly is not installed in the build container, so
the hexagonal convolution could only be tied to the oracle's restatement.  The tree search on
top of it is checked exactly all the same (it does not depend on what the network computes).  Needs a GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")

# Configs/Search/a1_search_config.yaml of the reference, with the simulation count of the BASELINE config
def a1_search(sims, **exploration):
    cfg = {"Simulation": {"mcts_simulations": sims, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
           "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                           "epsilon_random_exploration": 0.001, "value_factor": 1,
                           "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                           "root_dist_alpha": 0.15, "root_dist_beta": 1}}
    cfg["Exploration"].update(exploration)
    return cfg


def _net(cfg, arch, width, depth, seed, gain, hexnet=False, recall=True, vact="tanh", iters=1, max_batch=16):
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, convnet_param_shapes, resnet_param_shapes,
                                    recurrent_net_param_shapes)
    if arch == "convnet":
        shapes = convnet_param_shapes(cfg.channels, cfg.planes, 3, width, depth)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cfg.channels, cfg.planes, width, depth)
    else:
        shapes = recurrent_net_param_shapes(cfg.channels, cfg.planes, width, depth, recall)
    w = synthetic_weights(seed, hex_param_shapes(shapes) if hexnet else shapes, gain)
    net = BoardNet(arch, cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=width, num_blocks=depth, recall=recall,
                   value_activation=vact, kernel_size=3, max_batch=max_batch, hex=hexnet)
    net.set_weights(w, iters)
    return net, w


def _lockstep_recorded(path, search, net, seeds, fixed_batch, max_moves=None, training=True):
    """ScsSelfPlay.play (one host round trip per wave) with the native network, recording every leaf evaluation."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from scs_replay import LeafRecorder
    sp = ScsSelfPlay(ScsGameConfig(path), search, len(seeds), training=training)
    rec = LeafRecorder(net.evaluator(fixed_batch=fixed_batch))
    r = sp.play(rec, seeds, max_moves=max_moves)
    sp.close()
    return r, rec


def _replay_and_compare(path, search, seeds, r, rec, games, label, max_moves=None, training=True):
    from scs_replay import replay_games, assert_trace_equals_device
    jobs = [(path, search, seeds[g], training) + rec.arrays(g) + (max_moves,) for g in games]
    outs = replay_games(jobs)
    moves = 0
    for g, out in zip(games, outs):
        assert out["evaluations_used"] == out["evaluations_recorded"], (label, g)
        assert out["length"] == r["lengths"][g], (label, g)
        if out["terminal"]:
            assert out["terminal_value"] == r["outcomes"][g], (label, g)
        moves += assert_trace_equals_device(r, g, out, label)
    return moves


def _same_games(ra, rb, games, label):
    """Two device exports hold the same games (records past a game's end / a root's children are not defined)."""
    for g_a, g_b in games:
        n = ra["lengths"][g_a]
        assert n == rb["lengths"][g_b] and ra["outcomes"][g_a] == rb["outcomes"][g_b], (label, g_a)
        for k in ("actions", "tree_size", "n_children", "bias", "root_value_sum"):
            assert np.array_equal(ra[k][g_a, :n], rb[k][g_b, :n]), (label, k, g_a)
        for m in range(n):
            c = ra["n_children"][g_a, m]
            for k in ("child_action", "child_visit", "child_prior", "child_value_sum"):
                assert np.array_equal(ra[k][g_a, m, :c], rb[k][g_b, m, :c]), (label, k, g_a, m)


def _game_properties(path, r, games, full_games=True):
    """Every recorded game replays legally through the oracle rules: the root's children are exactly the legal
    actions, their visits add up to the root's minus one, and the game ends where the device says."""
    from oracle.scs import ScsConfig, ScsGame
    ocfg = ScsConfig(path)
    for g in games:
        og = ScsGame(ocfg)
        n = int(r["lengths"][g])
        assert n > 0
        for m in range(n):
            legal = np.nonzero(og.possible_actions().reshape(-1))[0]
            k = int(r["n_children"][g, m])
            assert r["child_action"][g, m, :k].tolist() == legal.tolist(), (g, m)
            assert int(r["child_visit"][g, m, :k].sum()) == int(r["tree_size"][g, m]) - 1, (g, m)
            assert int(r["actions"][g, m]) in legal, (g, m)
            og.step_index(int(r["actions"][g, m]))
        if full_games:
            assert og.terminal and og.terminal_value == r["outcomes"][g], g
        assert (r["actions"][g, n:] == -1).all()


@pytest.mark.parametrize("arch,hexnet", [("resnet", False), ("convnet", False), ("recurrent", True)])
def test_scs_search_with_the_native_network_equals_the_oracle_replay(arch, hexnet):
    """(a) of the composed statement, on whole games with exploration switched up (softmax moves, epsilon moves, noise):
    lock-step device search + BoardNet == oracle replay on the recorded evaluations, and the in-library move loop
    (nz_scs_search_play: leaf count on the device, terminal budget per wave) plays the same games.
    The hex=True case checks the search on top of the (parity-unpinned) hexagonal network."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "late_reinforcements_5x5.yml" if arch == "convnet" else "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G = 12
    net, _ = _net(cfg, arch, 32, 2, seed=41, gain=2.5, hexnet=hexnet, iters=2, max_batch=G)
    search = a1_search(48, number_of_softmax_moves=4, epsilon_softmax_exploration=0.1, epsilon_random_exploration=0.05,
                       root_exploration_fraction=0.25, root_dist_alpha=0.3)
    seeds = list(range(900, 900 + G))
    r, rec = _lockstep_recorded(path, search, net, seeds, fixed_batch=G)
    assert (r["lengths"] > 10).all()
    moves = _replay_and_compare(path, search, seeds, r, rec, range(G), arch)
    assert moves == int(r["lengths"].sum())
    sp = ScsSelfPlay(cfg, search, G)
    rn = sp.play_native(net, seeds)
    _same_games(r, rn, [(g, g) for g in range(G)], arch)
    assert rn["expansions"] == r["expansions"] == sum(len(v) for v in rec.records.values())
    sp.close(); net.close()


@pytest.mark.parametrize("hexnet", [False, True])
def test_baseline_config_4_scs_5x5_convnet_200_sims_1024_games(hexnet):
    """BASELINE.json configs[3]: SCS small map (the reference's mirrored_config_5.yml), ConvNet with 32 filters x 8
    layers (System_Tests/Neural_Networks/ConvNet_test.py:16), 200 simulations per move, 1024 concurrent self-play games
    on one GPU, whole pipeline in the library (nz_scs_search_play).  hex=False is the pinned form of the network,
    hex=True the form the config names (network parity UNPINNED; the search on top of it is exact either way).
    All 1024 games: legality / visit-sum / outcome properties through the oracle rules.  A sample of 8 games: exact
    oracle replay on the device network's own evaluations (recorded from an 8-game lock-step run with the same seeds
    and the same launch shapes -- games do not interact, so they are the same games)."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G, SAMPLE = 1024, 8
    net, w = _net(cfg, "convnet", 32, 8, seed=16, gain=2.0, hexnet=hexnet, max_batch=G)
    search = a1_search(200)
    seeds = list(range(4000, 4000 + G))
    sp = ScsSelfPlay(cfg, search, G)
    rn = sp.play_native(net, seeds)
    sp.close()
    assert rn["simulations"] == 200 * int(rn["lengths"].sum())
    _game_properties(path, rn, range(G))
    sample = list(range(0, G, G // SAMPLE))
    r, rec = _lockstep_recorded(path, search, net, [seeds[g] for g in sample], fixed_batch=G)
    _same_games(rn, r, [(g, i) for i, g in enumerate(sample)], "config4")
    moves = _replay_and_compare(path, search, [seeds[g] for g in sample], r, rec, range(SAMPLE), "config4")
    assert moves == int(r["lengths"].sum())
    # the first evaluation of every game is the opening position: the oracle network agrees within 1e-5
    from scipy.special import softmax
    from oracle.net import FeedForwardRef, HexNetRef
    from oracle.scs import ScsConfig, ScsGame
    ref = HexNetRef(w, "convnet", 8) if hexnet else FeedForwardRef(w, "convnet", 8)
    p, v = ref.inference(ScsGame(ScsConfig(path)).state_image(), None)
    _, probs0, value0 = rec.records[0][0]
    assert np.max(np.abs(softmax(p).reshape(-1) - probs0)) < 1e-5 and abs(float(v.reshape(-1)[0]) - float(value0)) < 1e-5
    net.close()


def test_baseline_config_5_scs_10x10_recurrent_256_wide_16_iterations_400_sims():
    """BASELINE.json configs[4] on one GPU's share: 10 x 10 map, RecurrentNet(86, 21, 256 filters, 2 blocks, recall,
    relu value head) (Run.py:148; square convs: the pinned form -- case K of net_kat3.npz holds the reference's outputs
    of exactly this network, tests/test_gpu_boardnet.py), 16 recurrent iterations, 400 simulations per move.  Bounded:
    16 games, the first 3 decisions of each (a full game is ~120 decisions x 400 evaluations x 9 GFLOP).  The first
    moves of a game do not depend on later ones, so what is checked is what a full game would check: the in-library loop
    against the lock-step route and both against the exact oracle replay (4 games), properties on all 16."""
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    path = os.path.join(GOLDEN, "scs_configs", "ten_by_ten.yml")
    cfg = ScsGameConfig(path)
    G, MOVES, SAMPLE = 16, 3, 4
    net, _ = _net(cfg, "recurrent", 256, 2, seed=15, gain=1.8, recall=True, vact="relu", iters=16, max_batch=G)
    search = a1_search(400)
    seeds = list(range(7000, 7000 + G))
    sp = ScsSelfPlay(cfg, search, G)
    rn = sp.play_native(net, seeds, max_moves=MOVES)
    sp.close()
    assert (rn["lengths"] == MOVES).all() and rn["simulations"] == 400 * MOVES * G
    _game_properties(path, rn, range(G), full_games=False)
    sample = list(range(0, G, G // SAMPLE))
    r, rec = _lockstep_recorded(path, search, net, [seeds[g] for g in sample], fixed_batch=G, max_moves=MOVES)
    _same_games(rn, r, [(g, i) for i, g in enumerate(sample)], "config5")
    moves = _replay_and_compare(path, search, [seeds[g] for g in sample], r, rec, range(SAMPLE), "config5", max_moves=MOVES)
    assert moves == MOVES * SAMPLE
    net.close()


def test_config_5_network_on_the_wide_kernel_equals_the_oracle():
    """The configs[4] network at a batch size where its 256-filter layers run on conv_wide_kernel (split-bf16 MFMA,
    LDS-staged tiles): 16 iterations, 10 x 10, 1e-5 against oracle/net.py (pinned to the reference by case K)."""
    import torch
    from scipy.special import softmax
    from conftest import NETS3, nets3_inputs, nets3_oracle, nets3_weights
    from nuzero_amd.boardnet import BoardNet
    arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, _, gain = NETS3["K"]
    n = 136                                   # 9 position groups: tiles at least half full -> wide kernel (boardnet.hip)
    net = BoardNet(arch, cin, planes, rows, cols, width=width, num_blocks=depth, recall=recall, value_activation=vact,
                   max_batch=n)
    net.set_weights(nets3_weights("K"), iters)
    x = nets3_inputs("K", n, offset=3)
    probs, value, logits = net.forward(torch.from_numpy(x).cuda(), want_logits=True)
    p, v = nets3_oracle("K").inference(x, iters)
    p = p.reshape(n, -1)
    scale = np.abs(p).max(axis=1, keepdims=True) + 1.0
    assert np.max(np.abs(logits.cpu().numpy() - p) / scale) < 1e-5
    assert np.max(np.abs(probs.cpu().numpy() - softmax(p, axis=1))) < 1e-5
    assert np.max(np.abs(value.cpu().numpy() - v.reshape(-1))) < 1e-5
    net.close()
