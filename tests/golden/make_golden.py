#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the GENUINE reference.

Run in the build container only (the reference does not travel to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from /root/reference (read-only, nothing is copied):
  Search.Explorer.Explorer, Search.Node.Node,
  Games.Tic_Tac_Toe.tic_tac_toe.tic_tac_toe,
  Neural_Networks.Network_Manager.Network_Manager,
  Neural_Networks.Architectures.RecurrentNet.RecurrentNet (hex=False).
Import recipe: inert stand-ins for `termcolor` (string
rendering only) and `hexagdly` (only referenced under hex=True), and the HEAD
drift shim ``tic_tac_toe.generate_network_input = generate_state_image``.
``Training.Gamer`` needs Ray, which is absent; its 40-line move loop
(Gamer.py:52-92) is driven here around the imported Explorer.

The "table network" vectors use the reference's own cache-hit branch
(Explorer.py:147-149): a cache-shaped object that always hits returns
post-softmax probabilities and a value for the position, so the genuine
Explorer code runs unmodified and no softmax/NN arithmetic is involved.

Outputs (all small):
  ttt_rules.npz   rules KATs        rng_kat.npz    legacy RandomState KATs
  net_kat.npz     NN KATs           unit_kat.json  select/expand corner cases
  search_kat.json.gz full-search KATs  meta.json      versions + recipe
"""
import gzip
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

_tc = types.ModuleType("termcolor")
_tc.colored = lambda s, *a, **k: s
sys.modules["termcolor"] = _tc
sys.modules["hexagdly"] = types.ModuleType("hexagdly")

import scipy  # noqa: E402
import torch  # noqa: E402
from scipy.special import softmax  # noqa: E402

from Search.Explorer import Explorer  # noqa: E402
from Search.Node import Node  # noqa: E402
from Games.Tic_Tac_Toe.tic_tac_toe import tic_tac_toe  # noqa: E402
from Neural_Networks.Network_Manager import Network_Manager  # noqa: E402
from Neural_Networks.Architectures.RecurrentNet import RecurrentNet  # noqa: E402
from Neural_Networks.Architectures.ResNet import ResNet  # noqa: E402
from Neural_Networks.Architectures.ConvNet import ConvNet  # noqa: E402

from nuzero_amd.weights import (synthetic_recurrent_net_weights, synthetic_weights,  # noqa: E402
                                resnet_param_shapes, convnet_param_shapes, recurrent_net_param_shapes)

tic_tac_toe.generate_network_input = tic_tac_toe.generate_state_image  # HEAD drift shim
torch.set_num_threads(1)


def code_of(board):
    k = 0
    for a in range(8, -1, -1):
        k = k * 3 + board[a // 3][a % 3]
    return k


def set_board(game, code):
    n = 0
    for a in range(9):
        c = code % 3
        code //= 3
        game.board[a // 3][a % 3] = c
        n += c != 0
    game.length = n
    game.agent_selection = (n % 2) + 1


def reachable():
    seen, stack = {}, [tic_tac_toe()]
    while stack:
        g = stack.pop()
        k = code_of(g.board)
        if k in seen:
            continue
        seen[k] = g.terminal
        if g.terminal:
            continue
        for a in range(9):
            if g.board[a // 3][a % 3] == 0:
                h = g.shallow_clone()
                h.step(g.get_action_coords(a))
                stack.append(h)
    return seen


# --------------------------------------------------------------------------- rules
def gen_rules(n_games=300):
    rnd = random.Random(1234)
    rows = {k: [] for k in ("game", "action", "board", "player", "mask", "image",
                            "terminal", "value", "length")}
    for gi in range(n_games):
        g = tic_tac_toe()
        while True:
            mask = g.possible_actions().flatten()
            for k, v in (("game", gi), ("board", [c for r in g.board for c in r]),
                         ("player", g.get_current_player()), ("mask", mask),
                         ("image", g.generate_state_image().numpy().reshape(-1)),
                         ("terminal", int(g.is_terminal())), ("value", g.get_terminal_value()),
                         ("length", g.get_length())):
                rows[k].append(v)
            if g.is_terminal():
                rows["action"].append(-1)
                break
            a = rnd.choice([i for i in range(9) if mask[i]])
            rows["action"].append(a)
            g.step(g.get_action_coords(a))
    np.savez_compressed(
        os.path.join(HERE, "ttt_rules.npz"),
        game=np.array(rows["game"], np.int32), action=np.array(rows["action"], np.int8),
        board=np.array(rows["board"], np.int8), player=np.array(rows["player"], np.int8),
        mask=np.array(rows["mask"], np.float64), image=np.array(rows["image"], np.float32),
        terminal=np.array(rows["terminal"], np.int8), value=np.array(rows["value"], np.int8),
        length=np.array(rows["length"], np.int8))
    reach = reachable()
    return {"plies": len(rows["game"]), "reachable": len(reach),
            "reachable_nonterminal": sum(1 for t in reach.values() if not t)}


# --------------------------------------------------------------------------- rng
def gen_rng():
    out = {}
    pats = []
    for seed in (0, 1, 7, 12345, 2**31 - 1):
        for alpha, beta in ((0.15, 1.0), (0.2, 1.0), (0.3, 0.5), (1.0, 1.0), (1.3, 2.0), (4.5, 0.25)):
            rs = np.random.RandomState(seed)
            seq = []
            for n in (9, 0, 8, 1, 7, 40):
                seq.extend(rs.gamma(alpha, beta, n).tolist())
                seq.append(rs.random_sample())
                seq.append(rs.random_sample())
                p = np.arange(1, 10, dtype=np.float64)
                p /= p.sum()
                seq.append(float(rs.choice(9, p=p)))
            key = f"s{seed}_a{alpha}_b{beta}"
            out[key] = np.array(seq, np.float64)
            pats.append(key)
    # raw generator outputs
    rs = np.random.RandomState(42)
    out["raw_u32_seed42"] = np.frombuffer(rs.bytes(4 * 1300), dtype="<u4").copy()
    rs = np.random.RandomState(42)
    out["double_seed42"] = rs.random_sample(1300)
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), **out)
    return {"patterns": pats, "draw_pattern": "for n in (9,0,8,1,7,40): gamma(a,b,n); random_sample(); "
            "random_sample(); choice(9, p=arange(1,10)/45)"}


# --------------------------------------------------------------------------- nets
NETS = {
    # name: (seed, width, gain)   -- RecurrentNet(2, 1, width, 2, recall=True, hex=False)
    "A": (0, 64, 1.0),     # the bench network
    "B": (1, 64, 3.0),     # sharper policies / larger values
    "C": (2, 16, 2.0),     # small
}


def build_ref_net(name):
    seed, width, gain = NETS[name]
    w = synthetic_recurrent_net_weights(seed, 2, 1, width, 2, True, gain)
    net = RecurrentNet(2, 1, width, 2, recall=True, hex=False)
    sd = net.state_dict()
    assert list(sd.keys()) == list(w.keys()), (list(sd.keys()), list(w.keys()))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return Network_Manager(net)


def ref_eval(nm, code, iters):
    g = tic_tac_toe()
    set_board(g, code)
    state = g.generate_network_input()
    p, v = nm.inference(state, False, iters)
    probs = softmax(p)
    return p.numpy().reshape(-1), probs.reshape(-1), np.float32(v.item())


def gen_nets(reach):
    codes = np.array(sorted(k for k, t in reach.items() if not t), np.int32)
    out = {"codes": codes}
    tables = {}
    for name in NETS:
        nm = build_ref_net(name)
        for iters, sel in ((2, codes), (1, codes[::71]), (16, codes[::71])):
            logits = np.zeros((len(sel), 9), np.float32)
            probs = np.zeros((len(sel), 9), np.float32)
            vals = np.zeros((len(sel),), np.float32)
            for i, c in enumerate(sel):
                logits[i], probs[i], vals[i] = ref_eval(nm, int(c), iters)
            out[f"{name}_i{iters}_logits"] = logits
            out[f"{name}_i{iters}_probs"] = probs
            out[f"{name}_i{iters}_value"] = vals
            if iters == 2:
                tables[name] = (probs, vals)
        # batched forward must equal batch-1 rows (reference accepts B > 1)
        g = tic_tac_toe()
        batch = []
        for c in codes[::600][:7]:
            set_board(g, int(c))
            batch.append(g.generate_network_input())
        p, v = nm.inference(torch.cat(batch, 0), False, 2)
        out[f"{name}_batch7_logits"] = p.numpy().reshape(7, 9)
        out[f"{name}_batch7_value"] = v.numpy().reshape(7)
    out["sub_index"] = np.arange(len(codes))[::71].astype(np.int32)
    out["batch7_index"] = np.arange(len(codes))[::600][:7].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "net_kat.npz"), **out)
    return codes, tables


# non-recurrent square nets: name -> (arch, seed, width, depth, kernel_size, gain)
NETS2 = {
    "D": ("resnet", 3, 32, 3, 3, 2.0),
    "E": ("convnet", 4, 32, 3, 3, 2.0),
    "F": ("convnet", 5, 48, 2, 1, 2.0),
}


def gen_nets2(reach):
    """ResNet / ConvNet with hex=False through Network_Manager.inference's non-recurrent branch."""
    codes = np.array(sorted(k for k, t in reach.items() if not t), np.int32)[::9]
    out = {"codes": codes}
    for name, (arch, seed, width, depth, k, gain) in NETS2.items():
        if arch == "resnet":
            w = synthetic_weights(seed, resnet_param_shapes(2, 1, width, depth), gain)
            net = ResNet(2, 1, num_filters=width, num_blocks=depth, hex=False)
        else:
            w = synthetic_weights(seed, convnet_param_shapes(2, 1, k, width, depth), gain)
            net = ConvNet(2, 1, kernel_size=k, num_filters=width, num_layers=depth, hex=False)
        sd = net.state_dict()
        assert list(sd.keys()) == list(w.keys()), (list(sd.keys()), list(w.keys()))
        net.load_state_dict({kk: torch.from_numpy(v) for kk, v in w.items()})
        nm = Network_Manager(net)
        logits = np.zeros((len(codes), 9), np.float32)
        probs = np.zeros((len(codes), 9), np.float32)
        vals = np.zeros((len(codes),), np.float32)
        g = tic_tac_toe()
        for i, c in enumerate(codes):
            set_board(g, int(c))
            p, v = nm.inference(g.generate_network_input(), False)
            logits[i], probs[i], vals[i] = p.numpy().reshape(-1), softmax(p).reshape(-1), v.item()
        out[f"{name}_logits"], out[f"{name}_probs"], out[f"{name}_value"] = logits, probs, vals
    np.savez_compressed(os.path.join(HERE, "net_kat2.npz"), **out)
    return {k: list(v) for k, v in NETS2.items()}


# board-sized nets (the SCS shapes): name -> (arch, seed, in, planes, rows, cols, width, depth, recall,
# value_activation, iters, positions, gain)
NETS3 = {
    "G": ("recurrent", 11, 86, 21, 5, 5, 32, 2, True, "tanh", 2, 6, 2.0),
    "H": ("resnet", 12, 105, 30, 6, 5, 48, 2, False, "relu", 1, 4, 2.0),
    "I": ("convnet", 13, 86, 21, 10, 10, 32, 3, False, "tanh", 1, 3, 2.0),
    "J": ("recurrent", 14, 86, 21, 5, 5, 64, 1, False, "relu", 3, 5, 2.0),
    # BASELINE configs[4] network (Run.py:148, square form): 256 filters, 2 blocks, recall, relu value head,
    # 16 recurrent iterations on a 10x10 board; configs[3] network (ConvNet_test.py:16, square form): 32 x 8
    "K": ("recurrent", 15, 86, 21, 10, 10, 256, 2, True, "relu", 16, 3, 1.8),
    "L": ("convnet", 16, 86, 21, 5, 5, 32, 8, False, "tanh", 1, 5, 2.0),
}


def nets3_inputs(name):
    """Sparse 0/1 planes plus a few fractional ones, like SCS state images."""
    _, seed, cin, _, rows, cols, *_rest = NETS3[name]
    n = NETS3[name][11]
    rs = np.random.RandomState(1000 + seed)
    x = (rs.random_sample((n, cin, rows, cols)) < 0.15).astype(np.float32)
    x[:, -3:] = rs.random_sample((n, 3, rows, cols)).astype(np.float32)
    return x


def gen_nets3():
    """hex=False RecurrentNet / ResNet / ConvNet on SCS-sized inputs through Network_Manager.inference."""
    out = {}
    for name, (arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain) in NETS3.items():
        if arch == "recurrent":
            w = synthetic_weights(seed, recurrent_net_param_shapes(cin, planes, width, depth, recall), gain)
            net = RecurrentNet(cin, planes, width, depth, recall=recall, value_activation=vact, hex=False)
        elif arch == "resnet":
            w = synthetic_weights(seed, resnet_param_shapes(cin, planes, width, depth), gain)
            net = ResNet(cin, planes, num_filters=width, num_blocks=depth, value_activation=vact, hex=False)
        else:
            w = synthetic_weights(seed, convnet_param_shapes(cin, planes, 3, width, depth), gain)
            net = ConvNet(cin, planes, kernel_size=3, num_filters=width, num_layers=depth, hex=False)
        sd = net.state_dict()
        assert list(sd.keys()) == list(w.keys()), (list(sd.keys()), list(w.keys()))
        net.load_state_dict({kk: torch.from_numpy(v) for kk, v in w.items()})
        nm = Network_Manager(net)
        x = nets3_inputs(name)
        logits = np.zeros((n, planes * rows * cols), np.float32)
        probs = np.zeros_like(logits)
        vals = np.zeros((n,), np.float32)
        for i in range(n):
            state = torch.from_numpy(x[i:i + 1])
            p, v = nm.inference(state, False, iters) if arch == "recurrent" else nm.inference(state, False)
            assert tuple(p.shape) == (1, planes, rows, cols)
            logits[i], probs[i], vals[i] = p.numpy().reshape(-1), softmax(p).reshape(-1), v.item()
        out[f"{name}_logits"], out[f"{name}_probs"], out[f"{name}_value"] = logits, probs, vals
    np.savez_compressed(os.path.join(HERE, "net_kat3.npz"), **out)
    return {k: list(v) for k, v in NETS3.items()}


# --------------------------------------------------------------------------- search
class TableCache:
    """Cache-shaped object that always hits (Explorer.py:146-149)."""

    def __init__(self, codes, probs, vals):
        self.rows = {int(c): (probs[i], vals[i]) for i, c in enumerate(codes)}
        self.hits = 0

    def get(self, state):
        s = state.numpy().reshape(2, 9)
        k = 0
        for a in range(8, -1, -1):
            k = k * 3 + (1 if s[0, a] else 2 if s[1, a] else 0)
        self.hits += 1
        p, v = self.rows[k]
        return p.reshape(1, 1, 3, 3).copy(), np.float32(v)

    def put(self, item):
        raise AssertionError("table cache must always hit")


def play_reference_game(search_config, training, seed, network, cache, iters=2):
    """Training/Gamer.py:52-92 around the imported Explorer; per-move trace."""
    np.random.seed(seed)            # legacy global stream == RandomState(seed)
    explorer = Explorer(search_config, training)
    game = tic_tac_toe()
    keep = search_config["Simulation"]["keep_subtree"]
    root = Node(0)
    moves = []
    stats = {"average_children": 0, "average_tree_size": 0, "final_tree_size": 0,
             "average_bias_value": 0, "final_bias_value": 0}
    while not game.is_terminal():
        game.store_state(game.generate_network_input())
        action, chosen, bias = explorer.run_mcts(game, network, root, iters, cache)
        moves.append({
            "action": int(action),
            "root_visits": int(root.visit_count),
            "root_value_sum": float(root.value_sum),
            "child_actions": [int(a) for a in root.children],
            "child_visits": [int(c.visit_count) for c in root.children.values()],
            "child_priors": [float(c.prior) for c in root.children.values()],
            "child_value_sums": [float(c.value_sum) for c in root.children.values()],
            "bias": float(bias),
        })
        tree_size, n_children = root.get_visit_count(), root.num_children()
        game.step(game.get_action_coords(action))
        game.store_search_statistics(root)
        if keep:
            root = chosen
        stats["average_children"] += n_children
        stats["average_tree_size"] += tree_size
        stats["final_tree_size"] = tree_size
        stats["average_bias_value"] += bias
        stats["final_bias_value"] = bias
    stats["number_of_moves"] = game.length
    for k in ("average_children", "average_tree_size", "average_bias_value"):
        stats[k] /= game.length
    return {
        "seed": seed, "moves": moves, "length": int(game.length),
        "terminal_value": int(game.terminal_value),
        "child_policy": [[float(x) for x in row] for row in game.child_policy],
        "states": [s.numpy().reshape(-1).astype(int).tolist() for s in game.state_history],
        "targets": [[int(game.make_target(i)[0])] + [float(x) for x in game.make_target(i)[1]]
                    for i in range(len(game.state_history))],
        "stats": {k: float(v) for k, v in stats.items()},
    }


def cfg(sims, base=5000, init=1.15, softmax_moves=0, eps_s=0, eps_r=0, vf=1, frac=0.2,
        alpha=0.15, beta=1):
    return {"Simulation": {"mcts_simulations": sims, "keep_subtree": True},
            "UCT": {"pb_c_base": base, "pb_c_init": init},
            "Exploration": {"number_of_softmax_moves": softmax_moves,
                            "epsilon_softmax_exploration": eps_s,
                            "epsilon_random_exploration": eps_r, "value_factor": vf,
                            "root_exploration_distribution": "gamma",
                            "root_exploration_fraction": frac,
                            "root_dist_alpha": alpha, "root_dist_beta": beta}}


SEARCH_CASES = [
    # name, table, config, training, seeds
    ("legacy100_A", "A", cfg(100), True, list(range(48))),
    ("legacy25_A", "A", cfg(25), True, list(range(100, 132))),
    ("legacy100_B", "B", cfg(100), True, list(range(200, 232))),
    ("explore50_B", "B", cfg(50, base=10000, softmax_moves=2, eps_s=0.3, eps_r=0.25, vf=0.75,
                             frac=0.25, alpha=0.3, beta=0.5), True, list(range(300, 348))),
    ("alpha_ge1_A", "A", cfg(30, alpha=1.3, beta=2.0, frac=0.1), True, list(range(400, 416))),
    ("eval40_B", "B", cfg(40), False, [500]),
    ("sims2_A", "A", cfg(2), True, list(range(600, 608))),
    ("sims400_C", "C", cfg(400, base=19652, init=1.25), True, list(range(700, 704))),
]


def gen_search(codes, tables):
    out = {}
    for name, tab, config, training, seeds in SEARCH_CASES:
        probs, vals = tables[tab]
        games = []
        for s in seeds:
            cache = TableCache(codes, probs, vals)
            games.append(play_reference_game(config, training, s, None, cache))
        out[name] = {"table": tab, "config": config, "training": training, "games": games}
    # the table route must equal the genuine network route (no cache) exactly
    nm = build_ref_net("A")
    probs, vals = tables["A"]
    for s in (0, 1, 2):
        real = play_reference_game(cfg(100), True, s, nm, None)
        tab = play_reference_game(cfg(100), True, s, None, TableCache(codes, probs, vals))
        assert real == tab, "table route differs from network route"
    with gzip.open(os.path.join(HERE, "search_kat.json.gz"), "wt", compresslevel=9) as f:
        json.dump(out, f, separators=(",", ":"))
    return {k: len(v["games"]) for k, v in out.items()}


# --------------------------------------------------------------------------- unit cases
def gen_unit():
    cases = {"select": [], "expand": [], "max_action": []}
    ex = Explorer(cfg(10), True)

    def build(parent_visits, to_play, kids):
        parent = Node(0)
        parent.visit_count, parent.to_play = parent_visits, to_play
        for a, (prior, n, vsum) in kids.items():
            c = Node(prior)
            c.visit_count, c.value_sum = n, vsum
            parent.children[a] = c
        return parent

    rnd = np.random.RandomState(5)
    trees = [
        (5, 1, {0: (0.25, 0, 0), 3: (0.25, 0, 0), 4: (0.25, 0, 0), 8: (0.25, 0, 0)}),   # exact tie -> 8
        (0, 1, {1: (0.7, 0, 0), 2: (0.3, 0, 0)}),                                         # N_p=0: all scores 0 -> 2
        (9, 2, {0: (0.5, 4, 2.0), 5: (0.5, 4, -2.0)}),                                    # negation for player 2
        (9, 1, {0: (0.5, 4, 2.0), 5: (0.5, 4, -2.0)}),
        (17, 2, {2: (0.1, 3, 0.0), 6: (0.1, 3, -0.0), 7: (0.1, 3, 0.0)}),               # +-0 tie
    ]
    for _ in range(40):
        k = rnd.randint(1, 10)
        acts = sorted(rnd.choice(9, k, replace=False).tolist())
        pri = rnd.dirichlet(np.ones(k))
        kids = {}
        for a, p in zip(acts, pri):
            n = int(rnd.randint(0, 6))
            kids[a] = (float(p), n, float(np.float32(rnd.uniform(-1, 1))) * n)
        trees.append((int(rnd.randint(0, 200)), int(rnd.randint(1, 3)), kids))
    for pv, tp, kids in trees:
        parent = build(pv, tp, kids)
        action, _ = ex.select_child(parent)
        cases["select"].append({
            "parent_visits": pv, "to_play": tp,
            "children": [[a, kids[a][0], kids[a][1], kids[a][2]] for a in kids],
            "scores": [float(ex.score(parent, parent.children[a])) for a in kids],
            "chosen": int(action)})
        vc = [(c.visit_count, a) for a, c in parent.children.items()]
        cases["max_action"].append({"visit_action": [[int(v), int(a)] for v, a in vc],
                                    "chosen": int(ex.max_action(vc))})

    class OneShot:
        def __init__(self, probs, v):
            self.p, self.v = probs, v

        def get(self, state):
            return self.p.reshape(1, 1, 3, 3).copy(), np.float32(self.v)

    boards = [0, 1 + 2 * 3, 1 + 2 * 3 + 9 + 2 * 27, 3 ** 4 + 2 * 3 ** 8 + 3 ** 2 + 2 * 3 ** 6]
    for code in boards:
        for kind in ("random", "zero_on_legal", "all_zero", "tiny"):
            g = tic_tac_toe()
            set_board(g, code)
            mask = g.possible_actions().flatten()
            if kind == "random":
                p = softmax(rnd.normal(size=9).astype(np.float32))
            elif kind == "zero_on_legal":
                p = ((1 - mask) / max(1, (1 - mask).sum())).astype(np.float32)
            elif kind == "all_zero":
                p = np.zeros(9, np.float32)
            else:
                p = softmax((rnd.normal(size=9) * 30).astype(np.float32))
            node = Node(0)
            v = ex.evaluate(node, g, OneShot(p, 0.125))
            cases["expand"].append({
                "code": code, "kind": kind, "probs": [float(x) for x in p],
                "value": float(v), "to_play": int(node.to_play),
                "child_actions": [int(a) for a in node.children],
                "child_priors": [float(c.prior) for c in node.children.values()]})
    with open(os.path.join(HERE, "unit_kat.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    return {k: len(v) for k, v in cases.items()}


E2E_CASES = [
    # name, net, config, seeds: the REAL network inside the genuine Explorer (no cache, no table)
    ("e2e25_A", "A", cfg(25), list(range(800, 816))),
    ("e2e100_A", "A", cfg(100), list(range(820, 836))),
    ("e2e25_B", "B", cfg(25), list(range(840, 856))),
    ("e2e100_B", "B", cfg(100), list(range(860, 876))),
]


def gen_e2e():
    """SURVEY.md section 8c KAT 6: seeded self-play games with the real RecurrentNet(2,1,64,2) evaluated by the genuine
    Explorer through Network_Manager.inference (Explorer.py:157-162), 2 recurrent iterations: per-move visit counts,
    priors, value sums.  The device's network differs from torch's in the last float32 bits, so these games are
    compared with an agreement rule (tests/test_gpu_parity.py::test_end_to_end_agreement_with_the_real_network)."""
    out = {}
    for name, net, config, seeds in E2E_CASES:
        nm = build_ref_net(net)
        out[name] = {"net": net, "config": config, "training": True,
                     "games": [play_reference_game(config, True, s, nm, None) for s in seeds]}
    with gzip.open(os.path.join(HERE, "e2e_kat.json.gz"), "wt") as f:
        json.dump(out, f)
    return {k: len(v["games"]) for k, v in out.items()}


def main():
    if "--only-e2e" in sys.argv:
        print(json.dumps(gen_e2e(), indent=1))
        return
    if "--only-nets2" in sys.argv:          # added later; leaves the other fixtures byte-identical
        print(json.dumps(gen_nets2(reachable()), indent=1))
        return
    if "--only-nets3" in sys.argv:
        print(json.dumps(gen_nets3(), indent=1))
        return
    meta = {"numpy": np.__version__, "scipy": scipy.__version__, "torch": torch.__version__,
            "python": sys.version.split()[0],
            "shims": ["termcolor stand-in", "hexagdly stand-in (hex=False only)",
                      "tic_tac_toe.generate_network_input = generate_state_image"],
            "nets": {k: {"seed": v[0], "width": v[1], "gain": v[2]} for k, v in NETS.items()}}
    meta["rules"] = gen_rules()
    meta["rng"] = gen_rng()
    reach = reachable()
    codes, tables = gen_nets(reach)
    meta["nets2"] = gen_nets2(reach)
    meta["nets3"] = gen_nets3()
    meta["search"] = gen_search(codes, tables)
    meta["unit"] = gen_unit()
    meta["e2e"] = gen_e2e()
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
