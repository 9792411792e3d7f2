"""SCS rules on the device (nz_scs_*: step, legal mask, state image) against the oracle
(oracle/scs.py, pinned to the reference by tests/test_scs_oracle.py), step by step on random
games.  Needs a GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,n_games", [("mirrored_5x5", 48), ("late_reinforcements_5x5", 48),
                                          ("two_types_6x5", 48), ("ten_by_ten", 12), ("wide_arrival_10x10", 8),
                                          ("reference_test_config", 12),
                                          ("randomized_5x5@7", 24), ("randomized_10x10@2", 8)])
def test_scs_device_rules_equal_oracle(name, n_games):
    """(name@seed: a "Randomized" configuration of the reference with the map it draws after np.random.seed(seed) --
    the maps themselves are pinned by tests/test_scs_oracle.py::test_randomized_maps_against_reference)"""
    from nuzero_amd.scs import ScsBatch, ScsGameConfig
    from oracle.scs import ScsConfig, ScsGame
    name, _, map_seed = name.partition("@")
    map_seed = int(map_seed) if map_seed else None
    path = os.path.join(GOLDEN, "scs_configs", name + ".yml")
    ocfg = ScsConfig(path, map_seed=map_seed)
    cfg = ScsGameConfig(path, map_seed=map_seed)
    assert (cfg.planes, cfg.channels, cfg.num_actions) == (ocfg.planes, ocfg.channels, ocfg.num_actions)
    batch = ScsBatch(cfg, n_games)
    games = [ScsGame(ocfg) for _ in range(n_games)]
    rs = np.random.RandomState(7)
    steps = 0
    while True:
        st = batch.status().cpu().numpy()
        mask = batch.legal_mask().cpu().numpy()
        img = batch.state_image().cpu().numpy()
        actions = np.full(n_games, -1, np.int32)
        for g, og in enumerate(games):
            want = (og.player, og.sub_phase, og.stage, og.turn, int(og.terminal), og.terminal_value, og.length)
            assert tuple(st[g]) == want, (steps, g, tuple(st[g]), want)
            assert np.array_equal(img[g], og.state_image()[0]), (steps, g)
            if og.terminal:
                assert not mask[g].any()
                continue
            om = og.possible_actions()
            assert np.array_equal(mask[g], om), (steps, g)
            actions[g] = rs.choice(np.nonzero(om.reshape(-1))[0])
            og.step_index(int(actions[g]))
        if (actions < 0).all():
            break
        batch.step(actions)
        steps += 1
    assert steps > 20
    outcomes = [g.terminal_value for g in games]
    assert len(outcomes) == n_games
    batch.close()


def _host_evaluator(num_actions):
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from scs_eval import evaluate_image

    def ev(images):
        imgs = images.cpu().numpy()
        out = [evaluate_image(im, num_actions) for im in imgs]
        return (torch.from_numpy(np.stack([o[0] for o in out])), torch.from_numpy(np.array([o[1] for o in out], np.float32)))
    return ev


def test_scs_device_search_equals_reference():
    """MCTS self-play on SCS with tree + rules on the device and evaluations injected from the host:
    every root statistic of every move must equal the games the genuine reference played
    (tests/golden/make_golden_scs_search.py): float32 priors, float64 after noise, exploration moves."""
    import gzip
    import json
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    with gzip.open(os.path.join(GOLDEN, "scs_search_kat.json.gz"), "rt") as f:
        kat = json.load(f)
    files = {"mirrored_config_5.yml": "mirrored_5x5.yml"}
    for name, case in kat.items():
        cfg = ScsGameConfig(os.path.join(GOLDEN, "scs_configs", files.get(case["config_file"], case["config_file"])))
        games = case["games"]
        sp = ScsSelfPlay(cfg, case["config"], len(games), training=case["training"])
        r = sp.play(_host_evaluator(cfg.num_actions), [g["seed"] for g in games])
        for g, ref in enumerate(games):
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
            assert (r["actions"][g, ref["length"]:] == -1).all()
        assert r["expansions"] == sum(g["evaluations"] for g in games)
        sp.close()


def test_scs_selfplay_with_a_torch_network_and_records():
    """End to end on SCS: a PyTorch conv net (square 3x3 convs, the hex=False form of the reference's
    nets) evaluates the leaves on the GPU, tree/rules/masks run on the device; the finished games
    come back as replay-buffer records whose targets follow SCS_Game.store_search_statistics."""
    import torch
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig, torch_evaluator, scs_game_records
    from nuzero_amd.replay_buffer import ReplayBuffer
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)

    class TinyNet(torch.nn.Module):
        recurrent = False

        def __init__(self, cin, planes):
            super().__init__()
            self.trunk = torch.nn.Sequential(torch.nn.Conv2d(cin, 16, 3, padding="same", bias=False), torch.nn.ReLU(),
                                             torch.nn.Conv2d(16, 16, 3, padding="same", bias=False), torch.nn.ReLU())
            self.policy = torch.nn.Conv2d(16, planes, 3, padding="same", bias=False)
            self.value = torch.nn.Conv2d(16, 1, 3, padding="same", bias=False)

        def forward(self, x):
            t = self.trunk(x)
            return self.policy(t), torch.tanh(self.value(t).mean(dim=(1, 2, 3))).reshape(-1, 1)

    torch.manual_seed(0)
    net = TinyNet(cfg.channels, cfg.planes).cuda()
    search = {"Simulation": {"mcts_simulations": 12, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.2, "root_dist_beta": 1}}
    sp = ScsSelfPlay(cfg, search, 6)
    r = sp.play(torch_evaluator(net), seeds=range(100, 106))
    assert (r["lengths"] > 10).all() and r["expansions"] == sp.evaluations
    # every recorded action must be legal when replayed through the oracle, and end where the device did
    ocfg = ScsConfig(path)
    for g in range(6):
        og = ScsGame(ocfg)
        for m in range(r["lengths"][g]):
            a = int(r["actions"][g, m])
            legal = np.nonzero(og.possible_actions().reshape(-1))[0]
            k = r["n_children"][g, m]
            assert r["child_action"][g, m, :k].tolist() == legal.tolist()
            assert r["child_visit"][g, m, :k].sum() == r["tree_size"][g, m] - 1
            og.step_index(a)
        assert og.terminal and og.terminal_value == r["outcomes"][g]
    recs = scs_game_records(sp, r)
    rb = ReplayBuffer(100, 8)
    for rec in recs:
        rb.save_game(rec, 0)
    assert rb.len() == int(r["lengths"].sum())
    state, (value, policy), idx = rb.get_buffer()[0]
    assert tuple(state.shape) == (1, cfg.channels, cfg.rows, cfg.cols) and len(policy) == cfg.num_actions
    assert abs(sum(policy) - 1.0) < 1e-12 and value in (-1, 0, 1)
    og = ScsGame(ocfg)
    assert np.array_equal(recs[0].get_state_from_history(0).numpy(), og.state_image())
    sp.close()


@pytest.mark.parametrize("hexnet", [False, True])
def test_gamer_and_network_manager_surface_for_scs(hexnet):
    """The reference's worker surface on SCS: Gamer(buffer, storage, SCS_Game, [config], ...) plays a round on the
    device with the model's own weights (state_dict names recognised, hex=True models by their kernel0/kernel1
    parameters -- parity unpinned), fills the replay buffer and returns the six statistics;
    Network_Manager.inference evaluates board-sized batches on the device (against the oracle nets)."""
    import torch
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.weights import synthetic_weights, resnet_param_shapes, hex_param_shapes
    from oracle.net import FeedForwardRef, HexNetRef
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    ocfg = ScsConfig(path)
    shapes = resnet_param_shapes(ocfg.channels, ocfg.planes, 32, 2)
    w = synthetic_weights(17, hex_param_shapes(shapes) if hexnet else shapes, 2.0)
    nm = Network_Manager(w)
    assert (nm.spec().arch, nm.spec().width, nm.spec().num_blocks, nm.spec().hex) == ("resnet", 32, 2, hexnet)

    class SCS_Game:                      # only the class name is looked at
        pass

    search = {"Simulation": {"mcts_simulations": 10, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.2, "root_dist_beta": 1}}
    rb = ReplayBuffer(100, 8)
    gamer = Gamer(rb, nm, SCS_Game, [path], 3, search, 1, "disabled", num_games=5)
    records, stats = gamer.play_games()
    assert len(records) == 5 and rb.len() == sum(r.length for r in records)
    assert set(stats[0]) == {"number_of_moves", "average_children", "average_tree_size", "final_tree_size",
                             "average_bias_value", "final_bias_value"}
    state, (value, policy), idx = rb.get_buffer()[0]
    assert tuple(state.shape) == (1, ocfg.channels, ocfg.rows, ocfg.cols) and len(policy) == ocfg.num_actions and idx == 3
    assert np.array_equal(records[0].get_state_from_history(0).numpy(), ScsGame(ocfg).state_image())
    st, cache = gamer.play_game()
    assert st["number_of_moves"] > 10 and cache.get_hit_ratio() == 0.0
    # Network_Manager.inference on a board-sized batch
    x = np.stack([r.get_state_from_history(min(3, r.length - 1)).numpy()[0] for r in records])
    logits, value = nm.inference(torch.from_numpy(x), False)
    ref = HexNetRef(w, "resnet", 2) if hexnet else FeedForwardRef(w, "resnet", 2)
    p, v = ref.inference(x, None)
    assert tuple(logits.shape) == (5, ocfg.planes, ocfg.rows, ocfg.cols)
    scale = float(np.abs(p).max()) + 1.0
    assert np.max(np.abs(logits.cpu().numpy() - p)) / scale < 1e-5 and np.max(np.abs(value.cpu().numpy() - v)) < 1e-5


@pytest.mark.parametrize("name,sims,n_games", [("wide_arrival_10x10", 24, 4), ("reference_test_config", 8, 3),
                                               ("many_units_10x10", 12, 3)])
def test_scs_search_limits_come_from_the_game_description(name, sims, n_games):
    """No fixed limits on the search: the reference's own test_config.yml (23 units, stacking 3: games of ~290 decisions,
    past the 256 a record used to hold) and a map whose opening positions have 90 legal actions (more than the 64 lanes
    of a wavefront: children are handled in chunks) play on the device exactly as on the oracle -- root noise, float32
    pairwise sums and PUCT ties included; so does a map with 27 units on one side, whose movement phase has more than 128 legal
    actions (three chunks; with softmax moves up to there, select_action's softmax sums more than the 128 values numpy
    adds in one pairwise block)."""
    import torch
    from scs_eval import evaluate_image
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from oracle import search as osearch
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", name + ".yml")
    cfg, ocfg = ScsGameConfig(path), ScsConfig(path)
    A = cfg.num_actions

    def device_ev(images):
        out = [evaluate_image(im, A) for im in images.cpu().numpy()]
        return (torch.from_numpy(np.stack([o[0] for o in out])), torch.from_numpy(np.array([o[1] for o in out], np.float32)))

    search = {"Simulation": {"mcts_simulations": sims, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 40 if name == "many_units_10x10" else 2,
                              "epsilon_softmax_exploration": 0.1,
                              "epsilon_random_exploration": 0.05, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.3, "root_dist_beta": 1}}
    seeds = list(range(70, 70 + n_games))
    sp = ScsSelfPlay(cfg, search, n_games)
    assert sp.MAX_CHILDREN % 64 == 0
    assert sp.MAX_CHILDREN >= {"wide_arrival_10x10": 128, "many_units_10x10": 192}.get(name, 64)
    r = sp.play(device_ev, seeds)
    sp.close()
    most_children, longest, softmax_over_128 = 0, 0, 0
    for g in range(n_games):
        game, trace = ScsGame(ocfg), []
        osearch.play_game(game, lambda gm: evaluate_image(gm.state_image()[0], A), search, np.random.RandomState(seeds[g]),
                          trace=trace)
        assert r["lengths"][g] == game.length and r["outcomes"][g] == game.terminal_value
        longest = max(longest, game.length)
        for m, mv in enumerate(trace):
            k = len(mv["child_actions"])
            most_children = max(most_children, k)
            softmax_over_128 += k > 128 and m < search["Exploration"]["number_of_softmax_moves"]
            assert r["actions"][g, m] == mv["action"], (g, m)
            assert r["child_action"][g, m, :k].tolist() == mv["child_actions"]
            assert r["child_visit"][g, m, :k].tolist() == mv["child_visits"], (g, m)
            assert r["child_prior"][g, m, :k].tolist() == mv["child_priors"], (g, m)
            assert r["child_value_sum"][g, m, :k].tolist() == mv["child_value_sums"]
    if name == "wide_arrival_10x10":
        assert most_children > 64
    elif name == "many_units_10x10":
        assert most_children > 128 and softmax_over_128 > 0
    else:
        assert longest > 150 and sp.MAX_MOVES > 256


@pytest.mark.parametrize("entries", [1 << 16, 64])
def test_scs_inference_cache_is_results_neutral(entries):
    """The device inference cache (KeylessCache semantics: index bits + stored hash, newest entry replaces) for the
    library's move loop: the games of a round are the same with and without it, also with a table so small that entries
    are replaced all the time; hits happen (games share positions) and the statistics add up."""
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    G = 48
    w = synthetic_weights(6, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 2), 2.0)
    net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=G)
    net.set_weights(w)
    search = {"Simulation": {"mcts_simulations": 30, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}
    seeds = list(range(300, 300 + G))
    plain = ScsSelfPlay(cfg, search, G)
    ra = plain.play_native(net, seeds)
    cached = ScsSelfPlay(cfg, search, G)
    cached.cache(entries)
    rb = cached.play_native(net, seeds)
    for k in ("lengths", "outcomes", "actions", "tree_size", "n_children"):
        assert np.array_equal(ra[k], rb[k]), k
    for g in range(G):
        n = ra["lengths"][g]
        assert np.array_equal(ra["root_value_sum"][g, :n], rb["root_value_sum"][g, :n])
        for m in range(n):
            c = ra["n_children"][g, m]
            for k in ("child_action", "child_visit", "child_prior", "child_value_sum"):
                assert np.array_equal(ra[k][g, m, :c], rb[k][g, m, :c]), (k, g, m)
    st = cached.cache_stats()
    assert st["size"] == entries and st["hits"] > 0 and st["hits"] + st["misses"] == rb["expansions"] == ra["expansions"]
    assert 0 < st["entries"] <= min(st["size"], st["misses"])
    # a second round on an emptied table: same again; a kept table: more hits, same games
    cached.cache_clear()
    assert cached.cache_stats()["entries"] == 0
    rc = cached.play_native(net, seeds)
    # (which of two leaves of one wave takes a shared entry depends on timing, so the hit count may differ a little
    # between two runs; the games may not)
    assert np.array_equal(rc["actions"], ra["actions"]) and abs(cached.cache_stats()["hits"] - st["hits"]) <= 0.2 * st["hits"]
    rd = cached.play_native(net, seeds)
    assert np.array_equal(rd["actions"], ra["actions"])
    if entries >= 1 << 16:
        assert cached.cache_stats()["hits"] > 2 * st["hits"]
    plain.close(); cached.close(); net.close()
    # the reference's surface: Gamer(..., cache_choice="keyless", cache_max) hands back a cache with these statistics
    class SCS_Game:
        pass
    gamer = Gamer(None, Network_Manager(w), SCS_Game, [path], 0, search, 1, "keyless", 1 << 12, num_games=8, records=False)
    stats, cache = gamer.play_game()
    assert 0.0 < cache.get_hit_ratio() < 1.0 and 0 < cache.length() <= 1 << 12 and 0.0 < cache.get_fill_ratio() <= 1.0
    assert cache.get_update_threshold() == 0.8 and cache.update(cache) is None


def test_scs_round_refills_finished_slots():
    """nz_scs_search_play_round: 22 games over 6 slots (a slot whose game has ended starts the round's next game; the
    games end at different moves, so slots restart at different times) are, game for game and record for record, the 22
    games a 22-slot engine plays from the same seeds; so is a second round on the same handle, and a round of exactly
    one game per slot."""
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    N, G = 22, 6
    w = synthetic_weights(9, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 2), 2.0)
    net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=N)
    net.set_weights(w)
    search = {"Simulation": {"mcts_simulations": 20, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 4, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.01, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.25,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}
    seeds = list(range(700, 700 + N))
    wide = ScsSelfPlay(cfg, search, N)
    ra = wide.play_native(net, seeds)
    assert len(set(ra["lengths"].tolist())) > 2, "the slots must restart at different moves for this test to bite"
    narrow = ScsSelfPlay(cfg, search, G)
    for attempt in range(2):
        rb = narrow.play_round(net, seeds)
        assert rb["actions"].shape[0] == N
        for k in ("lengths", "outcomes", "actions", "tree_size", "n_children"):
            assert np.array_equal(ra[k], rb[k]), (k, attempt)
        for g in range(N):
            n = ra["lengths"][g]
            assert np.array_equal(ra["root_value_sum"][g, :n], rb["root_value_sum"][g, :n])
            assert np.array_equal(ra["bias"][g, :n], rb["bias"][g, :n])
            for m in range(n):
                c = ra["n_children"][g, m]
                for k in ("child_action", "child_visit", "child_prior", "child_value_sum"):
                    assert np.array_equal(ra[k][g, m, :c], rb[k][g, m, :c]), (k, g, m)
        assert rb["simulations"] == ra["simulations"] and rb["expansions"] == ra["expansions"]
        assert rb["waves"] < ra["waves"] * (N / G)          # fewer waves than N / G rounds of one game per slot
    rc = narrow.play_round(net, seeds[:G])                  # exactly one game per slot: the plain move loop
    assert np.array_equal(rc["actions"], ra["actions"][:G]) and np.array_equal(rc["lengths"], ra["lengths"][:G])
    from nuzero_amd._lib import NzError
    with pytest.raises(NzError):
        narrow.play_round(net, seeds[:G - 1])
    wide.close(); narrow.close(); net.close()


def test_scs_round_larger_than_the_concurrent_trees():
    """Gamer(num_games=10, concurrent_games=4) on SCS: the round is played in batches of 4 trees (the reference's
    ActorPool: num_actors workers over num_games_per_step games); every game equals the one a 10-tree engine plays."""
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    from oracle.scs import ScsConfig
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    ocfg = ScsConfig(path)
    nm = Network_Manager(synthetic_weights(8, convnet_param_shapes(ocfg.channels, ocfg.planes, 3, 32, 2), 2.0))

    class SCS_Game:
        pass

    search = {"Simulation": {"mcts_simulations": 10, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.2, "root_dist_beta": 1}}
    out = {}
    for conc in (10, 4):
        rb = ReplayBuffer(100, 8)
        g = Gamer(rb, nm, SCS_Game, [path], 1, search, 1, "disabled", num_games=10, concurrent_games=conc, base_seed=40)
        records, stats = g.play_games()
        assert len(records) == len(stats) == 10 and rb.len() == sum(r.length for r in records)
        out[conc] = (records, stats)
    for ra, rb_ in zip(out[10][0], out[4][0]):
        assert ra.length == rb_.length and ra.terminal_value == rb_.terminal_value and ra.child_policy == rb_.child_policy
    assert out[10][1] == out[4][1]


@pytest.mark.parametrize("sims1,sims2", [(12, 20), (25, 9)])
def test_scs_evaluation_match_between_two_mcts_agents(sims1, sims2):
    """Tester.Test_using_agents with two MctsAgents that keep their subtrees, on SCS: two device engines
    (training=False), both search every decision, the mover's action is applied to both (nz_scs_search_apply).
    The action sequence equals oracle/agents.py (MctsAgent.py:28-39, Tester.py:62-118: agent 1 plays player index 1)."""
    import torch
    from scs_eval import evaluate_image
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from oracle.agents import MctsAgentRef, play_match
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", "late_reinforcements_5x5.yml")
    cfg, ocfg = ScsGameConfig(path), ScsConfig(path)
    A = cfg.num_actions

    def search_cfg(sims):
        return {"Simulation": {"mcts_simulations": sims, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
                "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                                "epsilon_random_exploration": 0.001, "value_factor": 1,
                                "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                                "root_dist_alpha": 0.15, "root_dist_beta": 1}}

    def device_ev(images):
        out = [evaluate_image(im, A) for im in images.cpu().numpy()]
        return (torch.from_numpy(np.stack([o[0] for o in out])), torch.from_numpy(np.array([o[1] for o in out], np.float32)))

    ev = lambda gm: evaluate_image(gm.state_image()[0], A)
    game = ScsGame(ocfg)
    want = play_match(game, MctsAgentRef(search_cfg(sims1), ev), MctsAgentRef(search_cfg(sims2), ev))
    e1 = ScsSelfPlay(cfg, search_cfg(sims1), 2, training=False)
    e2 = ScsSelfPlay(cfg, search_cfg(sims2), 2, training=False)
    e1.reset(); e2.reset()
    got = []
    while e1.status()[0, 4] == 0:
        player = int(e1.status()[0, 0])
        mover, other = (e1, e2) if player == 1 else (e2, e1)
        mover.search(device_ev)
        other.search(device_ev)               # update_subtree: the opponent searches the same position
        mover.apply()                          # choose_action (max_action)
        a = mover.last_actions()
        other.apply(actions=a)
        got.append(int(a[0]))
        assert int(a[0]) == int(a[1])
    assert got == want
    s1, s2 = e1.status(), e2.status()
    assert np.array_equal(s1, s2) and s1[0, 5] == game.terminal_value and s1[0, 6] == game.length
    from nuzero_amd._lib import NzError
    e3 = ScsSelfPlay(cfg, search_cfg(4), 1, training=False)
    e3.reset(); e3.search(device_ev)
    with pytest.raises(NzError):              # an action that is not legal in the position
        e3.apply(actions=[A - 1])
    e1.close(); e2.close(); e3.close()
