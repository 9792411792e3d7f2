"""Parity of the HIP path (through the C ABI) against the golden vectors made
from the genuine reference and against the oracle.  Needs a GPU."""
import numpy as np
import pytest

from conftest import full_table

pytestmark = pytest.mark.gpu


def _engine(config, n_games, training=True, n_slots=None):
    from nuzero_amd.engine import SelfPlayEngine
    return SelfPlayEngine(config, n_games, training=training, n_slots=n_slots)


def _images(codes):
    from oracle import ttt as ottt
    out = np.zeros((len(codes), 2, 3, 3), np.float32)
    for i, c in enumerate(codes):
        g = ottt.TicTacToe()
        g.board = ottt.board_from_code(int(c))
        out[i] = g.state_image()[0]
    return out


def _compare_with_reference_games(r, games):
    """r: engine export with trace; games: golden game dicts (same order)."""
    for g, ref in enumerate(games):
        L = ref["length"]
        assert r["lengths"][g] == L, (g, r["lengths"][g], L)
        assert r["outcomes"][g] == ref["terminal_value"]
        for m, mv in enumerate(ref["moves"]):
            assert r["actions"][g, m] == mv["action"], (g, m)
            assert r["tree_size"][g, m] == mv["root_visits"]
            assert r["n_children"][g, m] == len(mv["child_actions"])
            assert r["bias"][g, m] == mv["bias"]
            assert r["root_value_sum"][g, m] == mv["root_value_sum"]
            want_visits = np.zeros(9, np.int64)
            want_visits[mv["child_actions"]] = mv["child_visits"]
            assert np.array_equal(r["visits"][g, m], want_visits), (g, m)
            assert r["child_prior"][g, m][mv["child_actions"]].tolist() == mv["child_priors"], (g, m)
            assert r["child_value_sum"][g, m][mv["child_actions"]].tolist() == mv["child_value_sums"], (g, m)
            # policy target exactly as the reference computes it (tic_tac_toe.py:177-182)
            v = r["visits"][g, m].astype(np.int64)
            pol = [int(x) / int(v.sum()) if x else 0 for x in v]
            assert pol == ref["child_policy"][m]
            assert r["states"][g, m].reshape(-1).astype(int).tolist() == ref["states"][m]
        assert (r["actions"][g, L:] == -1).all()
        assert (r["states"][g, L:] == 0).all()


@pytest.mark.parametrize("case_name", ["legacy100_A", "legacy25_A", "legacy100_B", "explore50_B",
                                       "alpha_ge1_A", "eval40_B", "sims2_A", "sims400_C"])
def test_table_search_matches_reference(search_kat, net_kat, case_name):
    """Tree kernels alone (leaf evaluations injected from a table): every visit
    count, prior, value sum, action and outcome must equal the reference's."""
    case = search_kat[case_name]
    games = case["games"]
    eng = _engine(case["config"], len(games), training=case["training"])
    eng.set_table(full_table(net_kat, case["table"]))
    eng.play_with_numpy_rng([g["seed"] for g in games])
    _compare_with_reference_games(eng.export(trace=True), games)
    eng.close()


@pytest.mark.parametrize("route", ["persistent", "lockstep"])
def test_engine_random_streams_equal_numpy(search_kat, net_kat, route):
    """nz_engine_play (persistent self-play kernel, randomness pre-drawn from the
    library's host MT19937 streams) and nz_engine_play_lockstep reproduce the
    reference games too: seeds base..base+G-1."""
    for case_name in ("legacy100_A", "legacy25_A", "legacy100_B", "explore50_B", "alpha_ge1_A", "sims2_A",
                      "sims400_C"):
        case = search_kat[case_name]
        games = case["games"]
        eng = _engine(case["config"], len(games), training=True)
        eng.set_table(full_table(net_kat, case["table"]))
        if route == "persistent":
            eng.play(base_seed=games[0]["seed"])
        else:
            eng.play_lockstep(base_seed=games[0]["seed"])
        _compare_with_reference_games(eng.export(trace=True), games)
        c = eng.counters()
        assert c["simulations"] == sum(g["length"] for g in games) * case["config"]["Simulation"]["mcts_simulations"]
        eng.close()


@pytest.mark.parametrize("name,seed,width,gain", [("A", 0, 64, 1.0), ("B", 1, 64, 3.0), ("C", 2, 16, 2.0)])
def test_network_matches_reference(net_kat, name, seed, width, gain):
    """MFMA network kernel vs the reference's outputs on all 4,520 non-terminal
    positions: priors and values within 1e-5 (north-star tolerance)."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    w = synthetic_recurrent_net_weights(seed, 2, 1, width, 2, True, gain)
    eng = _engine(legacy_ttt_search_config(), 16)
    codes = net_kat["codes"]
    x = _images(codes)
    sub = net_kat["sub_index"]
    for iters, sel in ((2, np.arange(len(codes))), (1, sub), (16, sub)):
        if name == "B" and iters == 16:
            continue        # diverged net (|logit| ~ 2e7); see tests/test_oracle_golden.py
        eng.set_weights(w, width=width, recurrent_iterations=iters)
        logits, value, probs = eng.net_forward(x[sel])
        logits, value, probs = logits.cpu().numpy(), value.cpu().numpy(), probs.cpu().numpy()
        want_l = net_kat[f"{name}_i{iters}_logits"]
        scale = max(1.0, float(np.abs(want_l).max()))
        np.testing.assert_allclose(logits, want_l, rtol=0, atol=2e-6 * scale)
        np.testing.assert_allclose(probs, net_kat[f"{name}_i{iters}_probs"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(value, net_kat[f"{name}_i{iters}_value"], rtol=0, atol=1e-5)
    eng.close()


def test_network_batch_slot_invariance(net_kat):
    """A position's outputs must not depend on its batch slot or neighbours."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    eng = _engine(legacy_ttt_search_config(), 16)
    eng.set_weights(synthetic_recurrent_net_weights(1, 2, 1, 64, 2, True, 3.0))
    x = _images(net_kat["codes"][:1000])
    l0, v0, _ = eng.net_forward(x)
    perm = np.random.RandomState(0).permutation(1000)
    l1, v1, _ = eng.net_forward(x[perm][:333])
    assert np.array_equal(l0.cpu().numpy()[perm][:333], l1.cpu().numpy())
    assert np.array_equal(v0.cpu().numpy()[perm][:333], v1.cpu().numpy())
    l2, v2, _ = eng.net_forward(x[7:8])
    assert np.array_equal(l0.cpu().numpy()[7:8], l2.cpu().numpy())
    eng.close()


@pytest.mark.parametrize("n_slots", [16, 20, 1])
def test_round_larger_than_concurrency(search_kat, net_kat, n_slots):
    """A round of 48 games on fewer concurrent slots: finished slots take the
    next game of the round; every game still equals the reference's."""
    for case_name in ("legacy100_A", "explore50_B"):
        case = search_kat[case_name]
        games = case["games"]
        eng = _engine(case["config"], len(games), training=True, n_slots=n_slots)
        assert eng.n_slots == n_slots and eng.n_games == len(games)
        eng.set_table(full_table(net_kat, case["table"]))
        eng.play(base_seed=games[0]["seed"])
        _compare_with_reference_games(eng.export(trace=True), games)
        assert eng.live_games() == 0
        eng.close()


def test_eval_mode_persistent(search_kat, net_kat):
    """training=False (MctsAgent's use of run_mcts): no noise, max action."""
    case = search_kat["eval40_B"]
    eng = _engine(case["config"], 1, training=False)
    eng.set_table(full_table(net_kat, case["table"]))
    eng.play(base_seed=0)
    _compare_with_reference_games(eng.export(trace=True), case["games"])
    eng.close()


def test_desynced_games_are_replayed(net_kat):
    """With 3 simulations per move and always-random move selection the chosen
    child is often unvisited, so the next root is unexpanded and the persistent
    kernel's pre-drawn randomness does not fit: those games must come back
    through the lock-step replay, and every game must equal the oracle's."""
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import ttt as ottt, search as osearch
    cfg = legacy_ttt_search_config(3)
    cfg["Exploration"]["epsilon_random_exploration"] = 1.0
    table = full_table(net_kat, "B")
    eng = _engine(cfg, 40)
    eng.set_table(table)
    eng.play(base_seed=77)
    assert 0 < eng.desync_count() <= 40
    r = eng.export(trace=True)
    ev = osearch.table_evaluator(table)
    for g in range(40):
        game = ottt.TicTacToe()
        trace = []
        osearch.play_game(game, ev, cfg, np.random.RandomState(77 + g), trace=trace)
        assert r["lengths"][g] == game.length and r["outcomes"][g] == game.terminal_value
        for m, mv in enumerate(trace):
            assert r["actions"][g, m] == mv["action"]
            want = np.zeros(9, np.int64)
            want[mv["child_actions"]] = mv["child_visits"]
            assert np.array_equal(r["visits"][g, m], want)
            assert r["child_prior"][g, m][mv["child_actions"]].tolist() == mv["child_priors"]
    eng.close()


def _gpu_table(eng):
    """Evaluate the engine's own network on every position code -> [19683,10]."""
    codes = np.arange(3 ** 9)
    _, value, probs = eng.net_forward(_images(codes))
    t = np.zeros((3 ** 9, 10), np.float32)
    t[:, :9] = probs.cpu().numpy()
    t[:, 9] = value.cpu().numpy()
    return t


@pytest.mark.parametrize("route", ["persistent", "lockstep"])
@pytest.mark.parametrize("sims,n_games,net", [(100, 96, "A"), (25, 72, "B")])
def test_fused_search_equals_oracle_on_same_evaluations(sims, n_games, net, route):
    """End to end with the network fused into the search.  The oracle is given
    the GPU network's own outputs as a table, so every tree decision must match
    exactly; the network's closeness to the reference is test_network_matches_reference."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import ttt as ottt, search as osearch
    seed, gain = {"A": (0, 1.0), "B": (1, 3.0)}[net]
    cfg = legacy_ttt_search_config(sims)
    eng = _engine(cfg, n_games)
    eng.set_weights(synthetic_recurrent_net_weights(seed, 2, 1, 64, 2, True, gain))
    table = _gpu_table(eng)
    if route == "persistent":
        eng.play(base_seed=1000)
    else:
        eng.play_lockstep(base_seed=1000)
    r = eng.export(trace=True)
    counters = eng.counters()
    ev = osearch.table_evaluator(table)
    n_exp = 0
    for g in range(n_games):
        game = ottt.TicTacToe()
        trace = []
        _, cnt = osearch.play_game(game, ev, cfg, np.random.RandomState(1000 + g), trace=trace)
        n_exp += cnt.expansions
        assert r["lengths"][g] == game.length and r["outcomes"][g] == game.terminal_value
        for m, mv in enumerate(trace):
            assert r["actions"][g, m] == mv["action"], (g, m)
            want = np.zeros(9, np.int64)
            want[mv["child_actions"]] = mv["child_visits"]
            assert np.array_equal(r["visits"][g, m], want), (g, m)
            assert r["child_prior"][g, m][mv["child_actions"]].tolist() == mv["child_priors"]
            assert r["child_value_sum"][g, m][mv["child_actions"]].tolist() == mv["child_value_sums"]
    assert counters["expansions"] == n_exp
    eng.close()


def test_gamer_surface_matches_reference(search_kat, net_kat):
    """The Gamer-shaped adapter: records, targets and the six statistics must be
    what the reference's Gamer.play_game hands to ReplayBuffer / the trainer."""
    import torch
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.replay_buffer import ReplayBuffer

    class tic_tac_toe:      # the game class is only identified by name
        pass

    case = search_kat["legacy100_A"]
    games = case["games"]
    rb = ReplayBuffer(window_size=1000, batch_size=8)
    gamer = Gamer(rb, None, tic_tac_toe, [], 3, case["config"], 2, "disabled", num_games=len(games),
                  concurrent_games=16, base_seed=games[0]["seed"])
    gamer.engine.set_table(full_table(net_kat, "A"))
    records, stats = gamer.play_games()
    for rec, st, ref in zip(records, stats, games):
        assert rec.length == ref["length"] and rec.terminal_value == ref["terminal_value"]
        for i in range(rec.length):
            s = rec.get_state_from_history(i)
            assert s.dtype == torch.float32 and tuple(s.shape) == (1, 2, 3, 3) and not s.is_cuda
            assert s.reshape(-1).int().tolist() == ref["states"][i]
            v, pol = rec.make_target(i)
            assert [v] + [float(x) for x in pol] == ref["targets"][i]
        assert {k: float(v) for k, v in st.items()} == ref["stats"]
    assert rb.len() == sum(g["length"] for g in games) and rb.played_games() == len(games)
    state, (value, policy), idx = rb.get_buffer()[0]
    assert idx == 3 and len(policy) == 9
    gamer.engine.close()


def test_network_manager_inference(net_kat):
    """Network_Manager.inference on the GPU == the reference's outputs (1e-5)."""
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    nm = Network_Manager(synthetic_recurrent_net_weights(2, 2, 1, 16, 2, True, 2.0))
    assert nm.is_recurrent() and nm.spec().width == 16
    codes = net_kat["codes"][net_kat["sub_index"]]
    p, v = nm.inference(_images(codes), False, 16)
    assert tuple(p.shape) == (len(codes), 1, 3, 3) and tuple(v.shape) == (len(codes), 1)
    np.testing.assert_allclose(p.cpu().numpy().reshape(-1, 9), net_kat["C_i16_logits"], atol=2e-6)
    np.testing.assert_allclose(v.cpu().numpy().reshape(-1), net_kat["C_i16_value"], atol=1e-5)


def test_full_size_round_equals_c_oracle():
    """BASELINE configs[1] at full size: 4096 concurrent games, 100 simulations/move, network
    fused into the search (persistent kernel, 8192-game round).  The C oracle replays every game
    from the GPU network's own outputs; all visit counts, priors, actions and outcomes must be
    identical, plus size-independent invariants of the search."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import cref
    cfg = legacy_ttt_search_config(100)
    n_games, n_slots, base = 8192, 4096, 31337
    eng = _engine(cfg, n_games, n_slots=n_slots)
    eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
    table = _gpu_table(eng)
    eng.play(base_seed=base)
    r = eng.export(trace=True)
    c = eng.counters()
    assert eng.desync_count() == 0 and eng.live_games() == 0
    o = cref.play_games(table, cfg, [base + g for g in range(n_games)])
    for k in ("lengths", "outcomes", "actions", "visits", "tree_size", "n_children"):
        assert np.array_equal(r[k], o[k]), k
    for k in ("bias", "child_prior", "child_value_sum", "root_value_sum"):
        assert np.array_equal(r[k], o[k]), k
    assert c["simulations"] == o["simulations"] == int(r["lengths"].sum()) * 100
    assert c["expansions"] == o["expansions"]
    # invariants: a move's root children hold all but one of the root's visits; kept subtrees add up
    L = r["lengths"]
    for m in range(9):
        live = L > m
        vs = r["visits"][live, m].sum(1)
        assert np.array_equal(vs, r["tree_size"][live, m] - 1)
    assert (r["tree_size"][:, 0] == 100).all()
    states = r["states"]
    assert ((states == 0) | (states == 1)).all() and (states[:, 0] == 0).all()
    stones = states.reshape(n_games, 9, -1).sum(2)
    for m in range(9):
        assert (stones[L > m, m] == m).all()
    eng.close()


def test_bench_round_of_65536_games():
    """The round bench.py times by default: 65,536 games on 4096 concurrent trees (16 games per tree in one launch of the
    persistent kernel).  Size-independent invariants on every game; every eighth game replayed exactly by the C oracle
    from the GPU network's own outputs (each game has its own random stream, so any subset can be replayed)."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import cref
    cfg = legacy_ttt_search_config(100)
    n_games, n_slots, base = 65536, 4096, 900001
    eng = _engine(cfg, n_games, n_slots=n_slots)
    eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
    table = _gpu_table(eng)
    eng.play(base_seed=base)
    r = eng.export(trace=True)
    c = eng.counters()
    assert eng.desync_count() == 0 and eng.live_games() == 0
    L = r["lengths"]
    assert L.min() >= 5 and L.max() <= 9
    assert c["simulations"] == int(L.sum()) * 100
    for m in range(9):
        live = L > m
        assert np.array_equal(r["visits"][live, m].sum(1), r["tree_size"][live, m] - 1)
        legal = 9 - m                                             # a root's children are its legal moves
        assert (r["n_children"][live, m] == legal).all()
    assert (r["tree_size"][:, 0] == 100).all()
    states = r["states"]
    assert ((states == 0) | (states == 1)).all() and (states[:, 0] == 0).all()
    stones = states.reshape(n_games, 9, -1).sum(2)
    for m in range(9):
        assert (stones[L > m, m] == m).all()
    sample = np.arange(0, n_games, 8)
    o = cref.play_games(table, cfg, [base + int(g) for g in sample])
    for k in ("lengths", "outcomes", "actions", "visits", "tree_size", "n_children", "bias", "child_prior",
              "child_value_sum", "root_value_sum"):
        assert np.array_equal(r[k][sample], o[k]), k
    eng.close()


@pytest.mark.parametrize("width,iters,vact,sims,n_games,n_slots", [
    (16, 16, "tanh", 30, 40, 24),      # small net, 16 recurrent iterations, ragged tile
    (64, 1, "relu", 400, 20, 20),      # relu value head, 400 simulations (BASELINE configs[2] search depth)
    (32, 3, "tanh", 64, 1, 1),         # a single game
])
def test_fused_search_other_shapes(width, iters, vact, sims, n_games, n_slots):
    """Other network shapes / search depths through the persistent kernel, against the C oracle
    fed with the GPU network's outputs; network vs the torch fp32 oracle within 1e-5."""
    from scipy.special import softmax
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import cref
    from oracle.net import RecurrentNetRef
    cfg = legacy_ttt_search_config(sims)
    w = synthetic_recurrent_net_weights(11, 2, 1, width, 2, True, 1.5)
    eng = _engine(cfg, n_games, n_slots=n_slots)
    eng.set_weights(w, width=width, recurrent_iterations=iters, value_activation=vact)
    table = _gpu_table(eng)
    ref = RecurrentNetRef(w, 2, 1, width, 2, value_activation=vact)
    codes = np.arange(0, 3 ** 9, 37)
    p_ref, v_ref = ref.inference(_images(codes), iters)
    np.testing.assert_allclose(table[codes, :9], softmax(p_ref.reshape(-1, 9), axis=1), atol=1e-5)
    np.testing.assert_allclose(table[codes, 9], v_ref.reshape(-1), atol=1e-5)
    eng.play(base_seed=5)
    r = eng.export(trace=True)
    o = cref.play_games(table, cfg, [5 + g for g in range(n_games)])
    for k in ("lengths", "outcomes", "actions", "visits", "tree_size", "child_prior", "child_value_sum"):
        assert np.array_equal(r[k], o[k]), k
    eng.close()


def test_error_paths():
    """Bad arguments and out-of-order calls come back as status codes with a message."""
    from nuzero_amd import _lib
    from nuzero_amd._lib import NzError
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    cfg = legacy_ttt_search_config(10)
    bad = legacy_ttt_search_config(10)
    bad["Simulation"]["keep_subtree"] = False
    with pytest.raises(NzError, match="keep_subtree"):
        SelfPlayEngine(bad, 4)
    eng = SelfPlayEngine(cfg, 4)
    with pytest.raises(NzError) as ei:
        eng.play(0)                      # no network yet
    assert ei.value.code == _lib.NZ_ERR_STATE
    w = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True)
    with pytest.raises(NzError):
        eng.set_weights(dict(list(w.items())[:5]))        # wrong tensor count
    with pytest.raises(NzError):
        eng.set_weights(w, width=128)                     # unsupported width
    eng.set_weights(w)
    eng.play(0)
    eng2 = SelfPlayEngine(cfg, 32, n_slots=16)
    eng2.set_weights(w)
    with pytest.raises(NzError, match="n_slots == n_games"):
        eng2.play_lockstep(0)
    eng.close()
    eng2.close()


@pytest.mark.parametrize("sims1,tab1,sims2,tab2", [(20, "A", 50, "B"), (60, "B", 15, "A"), (33, "C", 33, "C")])
def test_evaluation_match_between_two_mcts_agents(net_kat, sims1, tab1, sims2, tab2):
    """Tester.Test_using_agents with two MctsAgents that keep their subtrees: two engines
    (training=False), both search every ply, both apply the mover's action."""
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import ttt as ottt, search as osearch
    from oracle.agents import MctsAgentRef, play_match
    cfg1, cfg2 = legacy_ttt_search_config(sims1), legacy_ttt_search_config(sims2)
    t1, t2 = full_table(net_kat, tab1), full_table(net_kat, tab2)
    game = ottt.TicTacToe()
    want = play_match(game, MctsAgentRef(cfg1, osearch.table_evaluator(t1)),
                      MctsAgentRef(cfg2, osearch.table_evaluator(t2)))
    e1, e2 = _engine(cfg1, 3, training=False), _engine(cfg2, 3, training=False)
    e1.set_table(t1)
    e2.set_table(t2)
    e1.reset()
    e2.reset()
    got = []
    ply = 0
    while e1.live_games() > 0:
        mover, other = (e1, e2) if ply % 2 == 0 else (e2, e1)
        mover.search()
        other.search()                       # update_subtree: the opponent searches the same position
        mover.apply()                        # choose_action
        a = mover.last_actions()
        other.apply(actions=a)
        got.append(int(a[0]))
        assert (a == a[0]).all()
        ply += 1
    assert got == want
    r1, r2 = e1.export(), e2.export()
    assert r1["outcomes"][0] == r2["outcomes"][0] == game.terminal_value
    assert np.array_equal(r1["actions"], r2["actions"])
    e1.close()
    e2.close()


@pytest.mark.parametrize("name", ["D", "E", "F"])
def test_feedforward_networks_match_reference(net_kat2, name):
    """ResNet and ConvNet (hex=False; 3x3 and 1x1 trunks, ELU) on the fused kernel vs the reference
    classes' outputs: priors and values within 1e-5; and a fused search against the C oracle."""
    from conftest import NETS2, nets2_weights
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import cref
    arch, seed, width, depth, k, gain = NETS2[name]
    w = nets2_weights(name)
    nm = Network_Manager(w)
    s = nm.spec()
    assert (s.arch, s.width, s.num_blocks, s.kernel_size) == (arch, width, depth, k) and not nm.is_recurrent()
    codes = net_kat2["codes"]
    p, v = nm.inference(_images(codes), False)
    want_l = net_kat2[f"{name}_logits"]
    np.testing.assert_allclose(p.cpu().numpy().reshape(-1, 9), want_l, atol=2e-6 * max(1.0, np.abs(want_l).max()))
    np.testing.assert_allclose(v.cpu().numpy().reshape(-1), net_kat2[f"{name}_value"], atol=1e-5)
    cfg = legacy_ttt_search_config(40)
    eng = _engine(cfg, 24)
    eng.set_weights(w, width=width, num_blocks=depth, arch=arch, kernel_size=k)
    _, _, probs = eng.net_forward(_images(codes))
    np.testing.assert_allclose(probs.cpu().numpy(), net_kat2[f"{name}_probs"], atol=1e-5)
    table = _gpu_table(eng)
    eng.play(base_seed=9)
    r = eng.export(trace=True)
    o = cref.play_games(table, cfg, [9 + g for g in range(24)])
    for key in ("lengths", "outcomes", "actions", "visits", "child_prior", "child_value_sum"):
        assert np.array_equal(r[key], o[key]), key
    eng.close()


def test_position_cache_is_results_neutral():
    """cache_choice != 'disabled': every leaf is read from the all-positions table; games must be
    identical to the uncached route (the reference's caches are results-neutral)."""
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights

    class tic_tac_toe:
        pass

    nm = Network_Manager(synthetic_recurrent_net_weights(1, 2, 1, 64, 2, True, 3.0))
    cfg = legacy_ttt_search_config(60)
    out = {}
    for choice in ("disabled", "dict"):
        g = Gamer(None, nm, tic_tac_toe, [], 0, cfg, 2, choice, num_games=48, base_seed=4)
        stats, cache = g.play_game()
        assert cache.get_hit_ratio() == (0.0 if choice == "disabled" else 1.0)
        out[choice] = g.engine.export(trace=True)
        g.engine.close()
    for k in ("lengths", "outcomes", "actions", "visits", "child_prior", "child_value_sum", "states"):
        assert np.array_equal(out["disabled"][k], out["dict"][k]), k


@pytest.mark.parametrize("gain,value_tol", [(0.05, 1e-5), (6.0, 1e-3)])
def test_network_keeps_float32_range(net_kat, gain, value_tol):
    """The split-bf16 arithmetic keeps float32's exponent range (an fp16 split would not): with tiny weights
    (logits around 1e-6) and with large ones (logits around 2e4) the logits stay within 1e-5 of the oracle
    relative to the largest logit.  Values: 1e-5 absolute for the small net (the kernel's tanh has an absolute
    error of 3e-7, so a value of 5e-10 comes out as 0); the large net saturates its value head's tanh layers, which
    turns the logit-level differences of ANY two float32 implementations (0.06 on 2e4 here) into 1e-4 on the value."""
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle.net import RecurrentNetRef
    w = synthetic_recurrent_net_weights(4, 2, 1, 64, 2, True, gain)
    eng = _engine(legacy_ttt_search_config(), 16)
    eng.set_weights(w, recurrent_iterations=2)
    x = _images(net_kat["codes"][::37])
    logits, value, _ = eng.net_forward(x)
    p, v = RecurrentNetRef(w, 2, 1, 64, 2).inference(np.asarray(x), 2)
    p = p.reshape(len(x), -1)
    scale = float(np.abs(p).max())
    assert scale > 0 and np.isfinite(scale)
    assert np.max(np.abs(logits.cpu().numpy() - p)) <= 1e-5 * scale
    np.testing.assert_allclose(value.cpu().numpy(), v.reshape(-1), rtol=0, atol=value_tol)
    eng.close()


def test_gamer_and_inference_follow_in_place_weight_updates():
    """Play a round, train the model in place (same module, same Network_Manager: what the reference's trainer hands
    over after every step, AlphaZero.py:152,293,462), play again: the second round and Network_Manager.inference must use
    the new weights -- identical to a fresh Gamer / Network_Manager built from them."""
    import torch
    from conftest import named_weights_module
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights

    class tic_tac_toe:
        pass

    w0 = synthetic_recurrent_net_weights(1, 2, 1, 64, 2, True, 3.0)
    w1 = synthetic_recurrent_net_weights(5, 2, 1, 64, 2, True, 3.0)
    model = named_weights_module(w0)
    nm = Network_Manager(model)
    cfg = legacy_ttt_search_config(40)
    keys = ("lengths", "outcomes", "actions", "visits", "child_prior", "child_value_sum")
    x = _images([0, 1, 5, 14, 70, 7, 16, 86, 3, 9, 27, 81, 243, 4, 10, 28])

    def round_of(gamer):
        gamer.base_seed = 4
        gamer.play_games()
        return gamer.engine.export(trace=True)

    g = Gamer(None, nm, tic_tac_toe, [], 0, cfg, 2, "disabled", num_games=32, base_seed=4)
    r0 = round_of(g)
    p0, v0 = nm.inference(x, False)
    with torch.no_grad():
        for p, v in zip(model.parameters(), w1.values()):
            p.copy_(torch.from_numpy(v))
    r1 = round_of(g)
    p1, v1 = nm.inference(x, False)
    fresh_nm = Network_Manager(w1)
    fresh = Gamer(None, fresh_nm, tic_tac_toe, [], 0, cfg, 2, "disabled", num_games=32, base_seed=4)
    r2 = round_of(fresh)
    p2, v2 = fresh_nm.inference(x, False)
    for k in keys:
        assert np.array_equal(r1[k], r2[k]), k
    assert any(not np.array_equal(r0[k], r1[k]) for k in keys)
    assert torch.equal(p1, p2) and torch.equal(v1, v2) and not torch.equal(p0, p1)
    r1b = round_of(g)                                   # unchanged weights: no re-upload, same games
    for k in keys:
        assert np.array_equal(r1[k], r1b[k]), k
    g.engine.close(); fresh.engine.close()


def test_end_to_end_agreement_with_the_real_network():
    """SURVEY.md section 8c KAT 6: 64 self-play games the GENUINE reference played with the real RecurrentNet(2,1,64,2)
    inside its Explorer (tests/golden/e2e_kat.json.gz: 25 and 100 simulations, the bench network A and the sharper B)
    against the device playing the same seeds with its own fused network.  The tree arithmetic is exact and the
    network agrees to 1e-5, but a last-bit difference in a prior can flip a near-tied PUCT argmax, after which the two
    games are different games.  Reported and bounded here: the fraction of moves whose visit vectors are identical,
    and -- on every move up to and including a game's first divergence -- priors within 1e-5 (north-star tolerance)
    and value sums within 1e-5 per visit."""
    import gzip
    import json
    import os
    from conftest import GOLDEN
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    with gzip.open(os.path.join(GOLDEN, "e2e_kat.json.gz"), "rt") as f:
        kat = json.load(f)
    nets = {"A": (0, 64, 1.0), "B": (1, 64, 3.0)}
    report = {}
    for name, case in kat.items():
        games = case["games"]
        seed, width, gain = nets[case["net"]]
        eng = _engine(case["config"], len(games), training=True)
        eng.set_weights(synthetic_recurrent_net_weights(seed, 2, 1, width, 2, True, gain), width=width, recurrent_iterations=2)
        eng.play_with_numpy_rng([g["seed"] for g in games])
        r = eng.export(trace=True)
        same_moves = total_moves = same_games = 0
        worst_prior = worst_value = 0.0
        for g, ref in enumerate(games):
            diverged = False
            for m, mv in enumerate(ref["moves"]):
                total_moves += 1
                if diverged:
                    continue
                acts = mv["child_actions"]
                # up to the first divergence the two searches saw the same positions: floats must agree to tolerance
                if r["n_children"][g, m] == len(acts):
                    worst_prior = max(worst_prior, float(np.abs(r["child_prior"][g, m][acts] - np.array(mv["child_priors"])).max()))
                    n = max(1, mv["root_visits"])
                    worst_value = max(worst_value, abs(float(r["root_value_sum"][g, m]) - mv["root_value_sum"]) / n)
                want = np.zeros(9, np.int64)
                want[acts] = mv["child_visits"]
                if np.array_equal(r["visits"][g, m], want) and r["actions"][g, m] == mv["action"]:
                    same_moves += 1
                else:
                    diverged = True
            if not diverged and r["lengths"][g] == ref["length"] and r["outcomes"][g] == ref["terminal_value"]:
                same_games += 1
        report[name] = {"identical_move_fraction": same_moves / total_moves, "identical_games": same_games,
                        "games": len(games), "max_prior_diff": worst_prior, "max_value_diff_per_visit": worst_value}
        eng.close()
    print("end-to-end agreement with the reference's real-network games:", json.dumps(report))
    for name, rep in report.items():
        assert rep["max_prior_diff"] <= 1e-5 and rep["max_value_diff_per_visit"] <= 1e-5, (name, rep)
        assert rep["identical_move_fraction"] >= 0.9, (name, rep)
    out_dir = os.environ.get("NZ_REPORT_DIR")
    if out_dir:
        with open(os.path.join(out_dir, "e2e_agreement.json"), "w") as f:
            json.dump(report, f, indent=1)


def test_baseline_config_3_share_of_one_gpu_equals_c_oracle():
    """BASELINE configs[2] (400 simulations/move, 8192 games over 8 GPUs) as ONE rank plays it: its shard of 1024 games
    (seeds of rank 5: nuzero_amd.dist.shard_seeds), 400 simulations/move, network fused into the search; the C oracle
    replays every game from the GPU network's own outputs -- all arrays identical.  (The 8-GPU split itself is the gather
    of tests/test_host_logic.py::test_round_on_two_ranks_gloo; there is no multi-GPU hardware in this pool.)"""
    from nuzero_amd import dist as nzdist
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from oracle import cref
    cfg = legacy_ttt_search_config(400)
    n_games, rank = 1024, 5
    base = nzdist.shard_seeds(90000, n_games, rank)
    assert base == 90000 + 5 * 1024
    eng = _engine(cfg, n_games)
    eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
    table = _gpu_table(eng)
    eng.play(base_seed=base)
    r = eng.export(trace=True)
    c = eng.counters()
    assert eng.live_games() == 0
    o = cref.play_games(table, cfg, [base + g for g in range(n_games)])
    for k in ("lengths", "outcomes", "actions", "visits", "tree_size", "n_children", "bias", "child_prior",
              "child_value_sum", "root_value_sum"):
        assert np.array_equal(r[k], o[k]), k
    assert c["simulations"] == o["simulations"] == int(r["lengths"].sum()) * 400 and c["expansions"] == o["expansions"]
    assert (r["tree_size"][:, 0] == 400).all()
    eng.close()


def test_next_round_randomness_drawn_ahead_changes_nothing(search_kat, net_kat):
    """nz_engine_play_next draws the next round's random numbers while the kernel runs: the round played from them equals
    the reference's games, whether the hint was right, wrong or absent."""
    case = search_kat["explore50_B"]
    games = case["games"]
    base = games[0]["seed"]
    eng = _engine(case["config"], len(games), training=True)
    eng.set_table(full_table(net_kat, case["table"]))
    eng.play(base_seed=base + 1000, next_base_seed=base)          # some other round first; hint = the golden round
    eng.play(base_seed=base, next_base_seed=base + 5)
    _compare_with_reference_games(eng.export(trace=True), games)
    eng.play(base_seed=base)                                     # prepared data for base + 5 is ignored
    _compare_with_reference_games(eng.export(trace=True), games)
    eng.close()


@pytest.mark.gpu
def test_round_pipeline_plays_the_same_rounds():
    """RoundPipeline (two engines, two streams, rounds in flight together): every round's games are the ones a single
    engine plays from the same base seed -- outcomes, lengths, visit records and counters."""
    from nuzero_amd.engine import SelfPlayEngine, RoundPipeline
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    w = synthetic_recurrent_net_weights(3, 2, 1, 64, 2, True)
    cfg = legacy_ttt_search_config(25)
    N, SLOTS, ROUNDS = 96, 32, 5

    def make():
        e = SelfPlayEngine(cfg, N, training=True, device=0, n_slots=SLOTS)
        e.set_weights(w, recurrent_iterations=2)
        return e

    def snapshot(e):
        import torch
        r = e.export()
        return {k: v.cpu().numpy().copy() for k, v in r.items() if torch.is_tensor(v)}, dict(e.counters())

    single = make()
    want = []
    for i in range(ROUNDS):
        single.play(base_seed=1000 * i)
        want.append(snapshot(single))
    single.close()
    pipe = RoundPipeline(make, depth=2)
    got = {}
    for i in range(ROUNDS):
        if len(pipe.pending) == 2:
            j, e = pipe.collect()
            got[j] = snapshot(e)
        pipe.submit(1000 * i, next_base_seed=1000 * (i + 2))
    while pipe.pending:
        j, e = pipe.collect()
        got[j] = snapshot(e)
    pipe.close()
    for i in range(ROUNDS):
        assert got[i][1] == want[i][1], i
        for k in want[i][0]:
            assert np.array_equal(got[i][0][k], want[i][0][k]), (i, k)


@pytest.mark.gpu
def test_gamer_play_forever_keeps_two_rounds_in_flight():
    """Gamer.play_forever (the trainer's asynchronous mode) with two rounds in flight: the rounds that reach the replay
    buffer are, in order, the rounds play_games() plays one after the other -- same records, same statistics -- and the
    weights of a round are those in shared storage when the round starts."""
    import threading
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights

    class tic_tac_toe:
        pass

    w = synthetic_recurrent_net_weights(5, 2, 1, 64, 2, True)
    cfg = legacy_ttt_search_config(25)
    N, SLOTS, ROUNDS = 48, 16, 5

    def gamer():
        return Gamer(ReplayBuffer(10 ** 6, 64), Network_Manager(w), tic_tac_toe, [], 0, cfg, 2, "disabled", num_games=N,
                     concurrent_games=SLOTS, base_seed=300)

    seq = gamer()
    want = [seq.play_games() for _ in range(ROUNDS)]
    free = gamer()
    got = []

    def on_round(records, stats):
        got.append((records, stats))
        if len(got) >= ROUNDS:
            free.stop()

    t = threading.Thread(target=free.play_forever, kwargs={"rounds_in_flight": 2, "on_round": on_round})
    t.start()
    t.join(timeout=120)
    assert not t.is_alive() and len(got) >= ROUNDS
    for (ra, sa), (rb, sb) in zip(want, got):
        assert sa == sb
        assert len(ra) == len(rb) == N
        for a, b in zip(ra, rb):
            assert a.length == b.length and a.terminal_value == b.terminal_value and a.child_policy == b.child_policy
    assert free.buffer.len() == sum(r.length for recs, _ in got for r in recs)
    assert free.base_seed == 300 + len(got) * N
