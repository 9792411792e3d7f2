"""The oracle (oracle/) against the golden vectors made from the genuine
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import full_table
from oracle import ttt as ottt
from oracle import search as osearch
from oracle.net import RecurrentNetRef, param_shapes
from nuzero_amd.weights import synthetic_recurrent_net_weights, recurrent_net_param_shapes


def test_rules_against_reference(rules_kat):
    k = rules_kat
    g = None
    last_game = -1
    for i in range(len(k["game"])):
        if k["game"][i] != last_game:
            g = ottt.TicTacToe()
            last_game = k["game"][i]
        assert g.board == k["board"][i].tolist()
        assert g.get_current_player() == k["player"][i]
        assert g.possible_actions().dtype == np.float64
        assert np.array_equal(g.possible_actions().flatten(), k["mask"][i])
        img = g.state_image()
        assert img.dtype == np.float32 and img.shape == (1, 2, 3, 3)
        assert np.array_equal(img.reshape(-1), k["image"][i])
        assert int(g.is_terminal()) == k["terminal"][i]
        assert g.get_terminal_value() == k["value"][i]
        assert g.get_length() == k["length"][i]
        if k["action"][i] >= 0:
            g.step_index(int(k["action"][i]))


def test_reachable_counts():
    codes = ottt.reachable_positions()
    assert len(codes) == 5478
    n_term = 0
    for c in codes:
        g = ottt.TicTacToe()
        g.board = ottt.board_from_code(c)
        g.length = sum(1 for x in g.board if x)
        g._check_terminal()
        n_term += g.terminal
    assert len(codes) - n_term == 4520


def test_param_shapes_agree():
    assert param_shapes(2, 1, 64, 2) == recurrent_net_param_shapes(2, 1, 64, 2)
    assert sum(int(np.prod(s)) for _, s in param_shapes(2, 1, 64, 2)) == 251568


@pytest.mark.parametrize("name,seed,width,gain", [("A", 0, 64, 1.0), ("B", 1, 64, 3.0), ("C", 2, 16, 2.0)])
def test_net_against_reference(net_kat, name, seed, width, gain):
    w = synthetic_recurrent_net_weights(seed, 2, 1, width, 2, True, gain)
    net = RecurrentNetRef(w, 2, 1, width, 2)
    codes = net_kat["codes"]

    def images(sel):
        out = np.zeros((len(sel), 2, 3, 3), np.float32)
        for i, c in enumerate(sel):
            g = ottt.TicTacToe()
            g.board = ottt.board_from_code(int(c))
            out[i] = g.state_image()[0]
        return out

    sub = net_kat["sub_index"]
    for iters, sel in ((2, np.arange(len(codes))), (1, sub), (16, sub)):
        if name == "B" and iters == 16:
            continue        # gain-3 net diverges (|logit| ~ 2e7) after 16 iterations
        x = images(codes[sel])
        # batch-1 calls, as the reference makes them: must be bit-identical
        for j in range(0, len(sel), max(1, len(sel) // 40)):
            p, v = net.inference(x[j:j + 1], iters)
            assert np.array_equal(p.reshape(-1), net_kat[f"{name}_i{iters}_logits"][j])
            assert np.float32(v.reshape(-1)[0]) == net_kat[f"{name}_i{iters}_value"][j]
        # one big batch: conv kernels may pick another algorithm -> tolerance
        p, v = net.inference(x, iters)
        np.testing.assert_allclose(p.reshape(len(sel), 9), net_kat[f"{name}_i{iters}_logits"],
                                   rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(v.reshape(-1), net_kat[f"{name}_i{iters}_value"], atol=1e-5)
    b7 = net_kat["batch7_index"]
    p, v = net.inference(images(codes[b7]), 2)
    assert np.array_equal(p.reshape(7, 9), net_kat[f"{name}_batch7_logits"])
    assert np.array_equal(v.reshape(7), net_kat[f"{name}_batch7_value"])


def test_select_cases(unit_kat):
    ex = osearch.Explorer(osearch.DEFAULT_SEARCH_CONFIG | {"UCT": {"pb_c_base": 5000, "pb_c_init": 1.15}}, True)
    for case in unit_kat["select"]:
        parent = osearch.Node(0)
        parent.visit_count, parent.to_play = case["parent_visits"], case["to_play"]
        for a, prior, n, vsum in case["children"]:
            c = osearch.Node(prior, int(a))
            c.visit_count, c.value_sum = int(n), vsum
            parent.children.append(c)
        scores = [ex.score(parent, c) for c in parent.children]
        assert scores == case["scores"]
        assert ex.select_child(parent).action == case["chosen"]
    for case in unit_kat["max_action"]:
        root = osearch.Node(0)
        for v, a in case["visit_action"]:
            c = osearch.Node(0.0, a)
            c.visit_count = v
            root.children.append(c)
        assert ex.max_action(root) == case["chosen"]


def test_expand_cases(unit_kat):
    ex = osearch.Explorer(osearch.DEFAULT_SEARCH_CONFIG, True)
    for case in unit_kat["expand"]:
        g = ottt.TicTacToe()
        g.board = ottt.board_from_code(case["code"])
        g.length = sum(1 for x in g.board if x)
        g.player = g.length % 2 + 1
        node = osearch.Node(0)
        probs = np.array(case["probs"], np.float32)
        v = ex.evaluate(node, g, lambda game: (probs, 0.125))
        assert v == case["value"] and node.to_play == case["to_play"]
        assert [c.action for c in node.children] == case["child_actions"]
        assert [float(c.prior) for c in node.children] == case["child_priors"]


def _check_game(ref, game, trace, stats):
    assert game.length == ref["length"]
    assert game.terminal_value == ref["terminal_value"]
    assert len(trace) == len(ref["moves"])
    for mine, theirs in zip(trace, ref["moves"]):
        for key in ("action", "root_visits", "root_value_sum", "child_actions", "child_visits",
                    "child_priors", "child_value_sums"):
            assert mine[key] == theirs[key], key
    assert [[float(x) for x in row] for row in game.child_policy] == ref["child_policy"]
    assert [s.reshape(-1).astype(int).tolist() for s in game.state_history] == ref["states"]
    for k, v in ref["stats"].items():
        assert float(stats[k]) == v, k


def test_full_search_against_reference(search_kat, net_kat):
    n = 0
    for name, case in search_kat.items():
        table = full_table(net_kat, case["table"])
        ev = osearch.table_evaluator(table)
        for ref in case["games"]:
            game = ottt.TicTacToe()
            trace = []
            stats, _ = osearch.play_game(game, ev, case["config"], np.random.RandomState(ref["seed"]),
                                         training=case["training"], trace=trace)
            _check_game(ref, game, trace, stats)
            n += 1
    assert n >= 180


def test_network_route_equals_reference(search_kat):
    """End to end with the real network (oracle net + scipy softmax) on the
    reference's own games: legacy config, net A, seeds 0-2."""
    w = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True, 1.0)
    net = RecurrentNetRef(w, 2, 1, 64, 2)
    ev = osearch.net_evaluator(net, 2)
    case = search_kat["legacy100_A"]
    for ref in case["games"][:3]:
        game = ottt.TicTacToe()
        trace = []
        stats, _ = osearch.play_game(game, ev, case["config"], np.random.RandomState(ref["seed"]),
                                     trace=trace)
        _check_game(ref, game, trace, stats)


def test_c_oracle_against_reference(search_kat, net_kat):
    """oracle/c/mcts_ref.c (used for full-size GPU checks) against the reference's golden games."""
    from oracle import cref
    for name, case in search_kat.items():
        games = case["games"]
        r = cref.play_games(full_table(net_kat, case["table"]), case["config"], [g["seed"] for g in games],
                            training=case["training"])
        for g, ref in enumerate(games):
            assert r["lengths"][g] == ref["length"] and r["outcomes"][g] == ref["terminal_value"]
            for m, mv in enumerate(ref["moves"]):
                assert r["actions"][g, m] == mv["action"]
                assert r["tree_size"][g, m] == mv["root_visits"] and r["bias"][g, m] == mv["bias"]
                assert r["root_value_sum"][g, m] == mv["root_value_sum"]
                assert r["visits"][g, m][mv["child_actions"]].tolist() == mv["child_visits"]
                assert r["child_prior"][g, m][mv["child_actions"]].tolist() == mv["child_priors"]
                assert r["child_value_sum"][g, m][mv["child_actions"]].tolist() == mv["child_value_sums"]


@pytest.mark.parametrize("name", ["D", "E", "F"])
def test_feedforward_nets_against_reference(net_kat2, name):
    """ResNet / ConvNet (hex=False) oracle vs the reference classes' outputs, bit for bit at batch 1."""
    from conftest import NETS2, nets2_weights
    from oracle.net import FeedForwardRef
    arch, seed, width, depth, k, gain = NETS2[name]
    net = FeedForwardRef(nets2_weights(name), arch, depth)
    codes = net_kat2["codes"]
    for j in range(0, len(codes), 25):
        g = ottt.TicTacToe()
        g.board = ottt.board_from_code(int(codes[j]))
        p, v = net.inference(g.state_image())
        assert np.array_equal(p.reshape(-1), net_kat2[f"{name}_logits"][j])
        assert np.float32(v.reshape(-1)[0]) == net_kat2[f"{name}_value"][j]


@pytest.mark.parametrize("name", ["G", "H", "I", "J", "K", "L"])
def test_board_sized_nets_against_reference(net_kat3, name):
    """The oracle nets on SCS-sized inputs (86/105 planes, 5x5 .. 10x10 boards, 21/30 policy planes)
    against the reference's RecurrentNet / ResNet / ConvNet (hex=False) through Network_Manager.inference
    (tests/golden/make_golden.py gen_nets3)."""
    from conftest import NETS3, nets3_inputs, nets3_oracle
    from scipy.special import softmax
    kat = net_kat3
    iters = NETS3[name][10]
    net = nets3_oracle(name)
    x = nets3_inputs(name)
    for i in range(len(x)):
        p, v = net.inference(x[i:i + 1], iters)
        assert np.array_equal(p.reshape(-1), kat[f"{name}_logits"][i])
        assert np.array_equal(softmax(p).reshape(-1), kat[f"{name}_probs"][i])
        assert np.float32(v.reshape(-1)[0]) == kat[f"{name}_value"][i]


@pytest.mark.parametrize("name", ["G", "H", "I", "J", "K", "L"])
def test_hex_nets_against_reference(name):
    """Pins oracle/net.py HexNetRef to the reference's hex=True nets -- once tests/golden/net_kat_hex.npz exists
    (tests/golden/make_golden_hex.py needs the real hexagdly package, which the build container lacks: until
    then hex parity is UNPINNED and this test is skipped)."""
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "net_kat_hex.npz")
    if not os.path.exists(path):
        pytest.skip("net_kat_hex.npz missing: hexagdly was not available to generate it (hex parity unpinned)")
    from conftest import NETS3, nets3_inputs
    from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, recurrent_net_param_shapes, resnet_param_shapes,
                                    convnet_param_shapes)
    from oracle.net import HexNetRef
    kat = np.load(path)
    arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain = NETS3[name]
    if arch == "recurrent":
        shapes = recurrent_net_param_shapes(cin, planes, width, depth, recall)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cin, planes, width, depth)
    else:
        shapes = convnet_param_shapes(cin, planes, 3, width, depth)
    w = synthetic_weights(100 + seed, hex_param_shapes(shapes), gain)
    p, v = HexNetRef(w, arch, depth, recall, vact).inference(nets3_inputs(name), iters)
    scale = np.abs(kat[f"{name}_logits"]).max(axis=1, keepdims=True) + 1.0
    assert np.max(np.abs(p.reshape(n, -1) - kat[f"{name}_logits"]) / scale) < 1e-5
    assert np.max(np.abs(v.reshape(-1) - kat[f"{name}_value"])) < 1e-5
