"""Corner cases of Explorer.evaluate / max_action on the HIP path (through the C ABI):

* the `total == 0` fallback of the expansion (Search/Explorer.py:171-174: a network that puts no mass on any legal
  move gives uniform priors over the legal moves), for Tic-Tac-Toe (expand_row, tree_dev.hpp) and SCS (wave_kernel,
  scs_search.hip);
* the expansion cases of tests/golden/unit_kat.json -- priors the GENUINE reference computed (random / zero on legal /
  all zero / tiny probabilities on four positions), reproduced by the device bit for bit;
* max_action ties -> the lowest action (Explorer.py:183-185), select ties -> the largest (Explorer.py:100).

Everything is compared bit-exactly with oracle/search.py (pinned to the reference by tests/test_oracle_golden.py,
which runs the same unit_kat cases on the oracle).  Needs a GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")


def _engine(config, n_games, training):
    from nuzero_amd.engine import SelfPlayEngine
    return SelfPlayEngine(config, n_games, training=training)


def _oracle_games(table, config, seeds, training):
    from oracle import search as osearch, ttt as ottt
    ev = osearch.table_evaluator(table)
    out = []
    for s in seeds:
        game, trace = ottt.TicTacToe(), []
        osearch.play_game(game, ev, config, np.random.RandomState(int(s)), training=training, trace=trace)
        out.append((game, trace))
    return out


def _compare(r, oracle_games):
    ties = 0
    for g, (game, trace) in enumerate(oracle_games):
        assert r["lengths"][g] == game.length and r["outcomes"][g] == game.terminal_value, g
        for m, mv in enumerate(trace):
            acts = mv["child_actions"]
            want = np.zeros(9, np.int64)
            want[acts] = mv["child_visits"]
            assert np.array_equal(r["visits"][g, m], want), (g, m)
            assert r["actions"][g, m] == mv["action"], (g, m)
            assert r["child_prior"][g, m][acts].tolist() == mv["child_priors"], (g, m)
            assert r["child_value_sum"][g, m][acts].tolist() == mv["child_value_sums"], (g, m)
            assert r["root_value_sum"][g, m] == mv["root_value_sum"] and r["tree_size"][g, m] == mv["root_visits"]
            top = max(mv["child_visits"])
            if mv["child_visits"].count(top) > 1:
                ties += 1
    return ties


def _zero_mass_table(kind):
    """[19683, 10] table whose probabilities avoid the legal moves.  'all_zero': every probability 0;
    'occupied': all mass on the occupied cells (none on the empty board); 'mixed': half of the positions like
    'occupied', the others a proper distribution.  Values vary with the position so that backups matter."""
    t = np.zeros((3 ** 9, 10), np.float32)
    codes = np.arange(3 ** 9)
    cells = (codes[:, None] // 3 ** np.arange(9)[None, :]) % 3
    occ = (cells != 0).astype(np.float32)
    n_occ = occ.sum(1, keepdims=True)
    if kind in ("occupied", "mixed"):
        t[:, :9] = np.where(n_occ > 0, occ / np.maximum(n_occ, 1), 0.0)
    if kind == "mixed":
        rs = np.random.RandomState(3)
        proper = rs.dirichlet(np.ones(9), size=3 ** 9).astype(np.float32)
        pick = (codes % 2 == 1)
        t[pick, :9] = proper[pick]
    t[:, 9] = (((codes * 7919) % 201) - 100).astype(np.float32) / 400.0 if kind != "all_zero" else 0.0
    return t


@pytest.mark.parametrize("route", ["persistent", "lockstep"])
@pytest.mark.parametrize("kind", ["all_zero", "occupied", "mixed"])
def test_ttt_zero_mass_expansion_equals_the_oracle(kind, route):
    """`total == 0` on the TTT path, both routes, training mode (noise on top of the uniform priors, 25 and 100 sims)."""
    from nuzero_amd.search_config import legacy_ttt_search_config
    table = _zero_mass_table(kind)
    for sims, G in ((25, 24), (100, 8)):
        cfg = legacy_ttt_search_config(sims)
        eng = _engine(cfg, G, training=True)
        eng.set_table(table)
        (eng.play if route == "persistent" else eng.play_lockstep)(base_seed=300)
        r = eng.export(trace=True)
        _compare(r, _oracle_games(table, cfg, range(300, 300 + G), True))
        if kind == "all_zero":               # uniform over the legal moves, before the noise of the NEXT move is mixed in
            k = r["n_children"][:, 0]
            assert (k == 9).all()
        eng.close()


def test_ttt_max_action_ties_go_to_the_lowest_action():
    """Evaluation mode (max_action, no randomness) on a table with uniform priors and value 0 everywhere: the root's
    children tie on their visit counts at move 0 and the first maximum -- the lowest action -- is played
    (Explorer.py:183-185), while ties on the PUCT score inside the search go to the LARGEST action (Explorer.py:100)."""
    from nuzero_amd.search_config import legacy_ttt_search_config
    table = _zero_mass_table("all_zero")
    total_ties = 0
    for sims in (10, 19, 100):
        cfg = legacy_ttt_search_config(sims)
        eng = _engine(cfg, 1, training=False)
        eng.set_table(table)
        eng.play_lockstep(base_seed=0)
        r = eng.export(trace=True)
        oracle = _oracle_games(table, cfg, [0], False)
        total_ties += _compare(r, oracle)
        v0 = r["visits"][0, 0]
        if (v0 == v0.max()).sum() > 1:
            assert r["actions"][0, 0] == int(np.argmax(v0))       # np.argmax: first maximum = lowest action
        eng.close()
    assert total_ties > 0


def test_unit_kat_expansion_cases_on_the_device(unit_kat):
    """tests/golden/unit_kat.json `expand`: (position, network probabilities) -> children priors as the genuine
    reference computed them.  Each case's position is reached with forced moves (nz_engine_apply), searched with a
    table that holds the case's probabilities for that position, and the root's priors of that move must be the
    reference's doubles exactly (evaluation mode: no noise on top)."""
    from oracle import ttt as ottt
    from nuzero_amd.search_config import legacy_ttt_search_config
    cases = unit_kat["expand"]
    kinds = sorted(set(c["kind"] for c in cases))
    assert {"zero_on_legal", "all_zero", "tiny", "random"} <= set(kinds)
    for kind in kinds:
        sel = [c for c in cases if c["kind"] == kind]
        G = len(sel)
        table = np.zeros((3 ** 9, 10), np.float32)
        table[:, :9] = 1.0 / 9.0
        paths = []
        for c in sel:
            table[c["code"], :9] = np.array(c["probs"], np.float32)
            table[c["code"], 9] = np.float32(c["value"])
            board = ottt.board_from_code(c["code"])
            xs = [a for a in range(9) if board[a] == 1]
            os_ = [a for a in range(9) if board[a] == 2]
            path = []
            for i in range(len(xs)):
                path.append(xs[i])
                if i < len(os_):
                    path.append(os_[i])
            assert len(path) == len(xs) + len(os_)
            paths.append(path)
        eng = _engine(legacy_ttt_search_config(12), G, training=False)
        eng.set_table(table)
        eng.reset()
        depth = max(len(p) for p in paths)
        for step in range(depth + 1):
            eng.search()
            forced = np.array([p[step] if step < len(p) else -1 for p in paths], np.int32)   # -1: the search's own choice
            eng.apply(forced)
        r = eng.export(trace=True)
        for g, c in enumerate(sel):
            m = len(paths[g])
            acts = c["child_actions"]
            assert r["n_children"][g, m] == len(acts), (kind, g)
            assert r["child_prior"][g, m][acts].tolist() == c["child_priors"], (kind, c["code"])
            assert (r["visits"][g, m][[a for a in range(9) if a not in acts]] == 0).all()
        eng.close()


def test_scs_zero_mass_expansion_equals_the_oracle():
    """`total == 0` on the SCS path: an evaluator that returns all-zero probabilities for some leaves (and proper ones
    for the others); float32 priors 1/k from `probs += mask` (Explorer.py:171-174) and everything downstream must equal
    the oracle's games with the same evaluator."""
    import torch
    from scs_eval import evaluate_image
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from oracle import search as osearch
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg, ocfg = ScsGameConfig(path), ScsConfig(path)
    A = cfg.num_actions

    zeroed = [0, 0]

    def evaluate(img):
        p, v = evaluate_image(img, A)
        zero = int(abs(float(v)) * 1000) % 3 != 1
        zeroed[int(zero)] += 1
        if zero:
            p = np.zeros_like(p)               # no mass anywhere -> total == 0 at this leaf
        return p, v

    def device_ev(images):
        out = [evaluate(im) for im in images.cpu().numpy()]
        return (torch.from_numpy(np.stack([o[0] for o in out])), torch.from_numpy(np.array([o[1] for o in out], np.float32)))

    search = {"Simulation": {"mcts_simulations": 24, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 3, "epsilon_softmax_exploration": 0.1,
                              "epsilon_random_exploration": 0.05, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.3, "root_dist_beta": 1}}
    G = 6
    seeds = list(range(60, 60 + G))
    sp = ScsSelfPlay(cfg, search, G)
    r = sp.play(device_ev, seeds)
    sp.close()
    for g in range(G):
        game, trace = ScsGame(ocfg), []
        osearch.play_game(game, lambda gm: evaluate(gm.state_image()[0]), search, np.random.RandomState(seeds[g]), trace=trace)
        assert r["lengths"][g] == game.length and r["outcomes"][g] == game.terminal_value
        for m, mv in enumerate(trace):
            k = len(mv["child_actions"])
            assert r["actions"][g, m] == mv["action"], (g, m)
            assert r["child_action"][g, m, :k].tolist() == mv["child_actions"]
            assert r["child_visit"][g, m, :k].tolist() == mv["child_visits"], (g, m)
            assert r["child_prior"][g, m, :k].tolist() == mv["child_priors"], (g, m)
            assert r["child_value_sum"][g, m, :k].tolist() == mv["child_value_sums"], (g, m)
    assert zeroed[0] > 100 and zeroed[1] > 100          # both branches of the expansion ran, many times
