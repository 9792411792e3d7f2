"""SURVEY.md section 8(f) ranks 2 and 3 on the device: the replay buffer in HBM with batch assembly by one gather
kernel (nz_replay_*), and the batched loss with gradients (nz_loss_forward_backward), through the C ABI.

Replay buffer: bit-exact on indices and contents against (1) tests/golden/replay_kat.json, traces and contents of the
GENUINE ReplayBuffer class (tests/golden/make_golden_replay.py), and (2) the host ReplayBuffer of this repository
(the reference's list semantics, tests/test_host_logic.py) fed with the same games.
Loss: against tests/golden/loss_kat.npz (the reference's loss functions and its calculate_loss loop with torch autograd
gradients).  Tolerance: float32 sums in a different order than torch's -- 2e-6 relative on the losses, 2e-6 of the
largest gradient entry on the gradients (stated per assert).  Needs a GPU."""
import json
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, "golden")


class _Game:
    """What ReplayBuffer.save_game reads; same construction as tests/golden/make_golden_replay.py FakeGame."""

    def __init__(self, gid, length, num_actions, rs):
        import torch
        self.gid = gid
        self.state_history = [torch.tensor([[float(gid), float(m)]]) for m in range(length)]
        self.value = int(rs.randint(-1, 2))
        self.policies = []
        for m in range(length):
            v = rs.randint(0, 50, size=num_actions) * (rs.random_sample(num_actions) < 0.6)
            if v.sum() == 0:
                v[rs.randint(num_actions)] = 7
            total = int(v.sum())
            self.policies.append([int(x) / total for x in v])

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        return (self.value, self.policies[i])


def test_device_buffer_contents_equal_the_genuine_reference():
    """Five games through a window of three (evictions), position by position: identity, value, game index and the
    float32 policy the trainer makes of the target (torch.tensor(list), AlphaZero.py:901) -- as the genuine class
    returned them."""
    from nuzero_amd.replay_device import DeviceReplayBuffer
    with open(os.path.join(GOLDEN, "replay_kat.json")) as f:
        case = json.load(f)["content"]
    rs = np.random.RandomState(case["seed"])
    rb = DeviceReplayBuffer(case["window"], 4, (1, 1, 2), 9, max_game_length=8)
    games = [_Game(g, int(rs.randint(2, 6)), 9, rs) for g in range(5)]
    assert [len(g.state_history) for g in games] == case["lengths"]
    for g in games:
        rb.save_game(g, g.gid % 2)
    got = rb.get_buffer()
    rb.check()
    assert len(got) == len(case["content"]) == rb.len() and rb.played_games() == 3
    for (state, (value, policy), gi), want in zip(got, case["content"]):
        assert [int(state.reshape(-1)[0]), int(state.reshape(-1)[1]), gi] == want["id"]
        assert value == want["value"]
        assert np.array_equal(np.asarray(policy, np.float32), np.asarray(want["policy_f32"], np.float32))
    rb.close()


def _host_and_device_buffers(window, G, sims, seed):
    """A Tic-Tac-Toe round saved (a) game by game into the host ReplayBuffer from GameRecords and (b) from the engine's
    export buffers into the device buffer."""
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.gamer import GameRecord
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.replay_device import DeviceReplayBuffer
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    eng = SelfPlayEngine(legacy_ttt_search_config(sims), G, training=True)
    eng.set_weights(synthetic_recurrent_net_weights(3, 2, 1, 64, 2, True, 3.0))
    host, dev = ReplayBuffer(window, 32), DeviceReplayBuffer(window, 32, (2, 3, 3), 9, max_game_length=9)
    for rnd in range(3):
        eng.play(base_seed=seed + 1000 * rnd)
        r = eng.export()
        for g in range(G):
            host.save_game(GameRecord(r["states"][g], r["visits"][g], r["actions"][g], r["lengths"][g], r["outcomes"][g]),
                           rnd % 2)
        dev.save_games_from_engine(eng, rnd % 2)
    dev.check()
    eng.close()
    return host, dev


def _same_entries(got, want):
    import torch
    assert len(got) == len(want)
    for (s1, (v1, p1), g1), (s2, (v2, p2), g2) in zip(got, want):
        assert torch.equal(s1, s2) and v1 == v2 and g1 == g2
        assert np.array_equal(np.asarray(p1, np.float32), torch.tensor(p2).numpy())      # AlphaZero.py:901


@pytest.mark.parametrize("window", [40, 100000])
def test_device_buffer_equals_the_host_list_through_a_training_schedule(window):
    """Rounds of real self-play games, window smaller / larger than what is played: buffer order and contents, shuffles,
    slices, samples (uniform, without replacement, late_heavy) and batches grouped by game index -- device buffer ==
    host list (the reference's semantics) under the same `random` / `np.random` seeds."""
    import torch
    from nuzero_amd.replay_device import late_heavy_probs
    host, dev = _host_and_device_buffers(window, 24, 20, seed=11)
    assert dev.len() == host.len() and dev.played_games() == host.played_games()
    _same_entries(dev.get_buffer(), host.get_buffer())
    for seed in (1, 2):
        random.seed(seed); host.shuffle()
        random.seed(seed); dev.shuffle()
        _same_entries(dev.get_slice(3, 35).as_list(), host.get_slice(3, 35))
    n = host.len()
    for replace, probs in ((True, []), (False, []), (True, late_heavy_probs(n))):
        np.random.seed(5); want = host.get_sample(32, replace, probs)
        np.random.seed(5); got = dev.get_sample(32, replace, probs)
        _same_entries(got.as_list(), want)
        assert got.states.shape == (32, 2, 3, 3) and got.policies.shape == (32, 9) and got.states.is_cuda
    # grouped by game index, keys ascending, order kept inside a group (AlphaZero.py:846-852)
    np.random.seed(9); want = host.get_sample(32, True, [])
    np.random.seed(9); got = dev.get_sample(32, True, [], group_by_game=True)
    keys = sorted(set(e[2] for e in want))
    assert got.keys == keys
    for (k, states, policies, values), key in zip(got.by_game(), keys):
        group = [e for e in want if e[2] == key]
        assert k == key and torch.equal(states.cpu(), torch.cat([e[0] for e in group], 0))
        assert torch.equal(policies.cpu(), torch.tensor([e[1][1] for e in group]))
        assert values.cpu().tolist() == [float(e[1][0]) for e in group]
    dev.close()


def test_device_buffer_checkpoint_round_trip(tmp_path):
    """save_to_file writes the reference's layout; the host class and a fresh device buffer both load it."""
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.replay_device import DeviceReplayBuffer
    host, dev = _host_and_device_buffers(100000, 8, 10, seed=3)
    dev.save_to_file(tmp_path / "rb.pt", step=7)
    other = ReplayBuffer(100000, 32)
    other.load_from_file(tmp_path / "rb.pt", 7)
    _same_entries(dev.get_buffer(), other.get_buffer())
    again = DeviceReplayBuffer(100000, 32, (2, 3, 3), 9, max_game_length=9)
    again.load_from_file(tmp_path / "rb.pt", 7)
    _same_entries(again.get_buffer(), host.get_buffer())
    assert again.played_games() == host.played_games()
    dev.close(); again.close()


def test_scs_games_into_the_device_buffer():
    """SCS round -> device buffer (state images regenerated by the device rules, sparse policy targets) == the host
    records of the same round (scs_game_records, tests/test_gpu_scs.py) saved into the host list."""
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.replay_device import DeviceReplayBuffer
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig, scs_game_records
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    path = os.path.join(GOLDEN, "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=6)
    net.set_weights(synthetic_weights(4, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 2), 2.0))
    search = {"Simulation": {"mcts_simulations": 12, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.2, "root_dist_beta": 1}}
    sp = ScsSelfPlay(cfg, search, 6)
    r = sp.play_native(net, range(20, 26))
    host = ReplayBuffer(4, 8)
    for rec in scs_game_records(sp, r):
        host.save_game(rec, 1)
    dev = DeviceReplayBuffer(4, 8, (cfg.channels, cfg.rows, cfg.cols), cfg.num_actions, max_game_length=sp.MAX_MOVES)
    dev.save_scs_games(sp, sp.export_device(), 1)
    dev.check()
    assert dev.len() == host.len() and dev.played_games() == 4
    _same_entries(dev.get_buffer(), host.get_buffer())
    dev.close(); sp.close(); net.close()


def test_gamer_fills_the_device_buffer_without_host_records():
    """Gamer(records=False) + DeviceReplayBuffer: the round's positions go from the engine into the buffer on the
    device; the statistics still come back per game (Gamer.py:42-50)."""
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_device import DeviceReplayBuffer
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights

    class tic_tac_toe:
        pass

    nm = Network_Manager(synthetic_recurrent_net_weights(1, 2, 1, 64, 2, True, 3.0))
    dev = DeviceReplayBuffer(1000, 64, (2, 3, 3), 9, max_game_length=9)
    g = Gamer(dev, nm, tic_tac_toe, [], 2, legacy_ttt_search_config(25), 2, "disabled", num_games=64, base_seed=4,
              records=False)
    records, stats = g.play_games()
    assert records == [] and len(stats) == 64
    ref = Gamer(None, nm, tic_tac_toe, [], 2, legacy_ttt_search_config(25), 2, "disabled", num_games=64, base_seed=4)
    recs, ref_stats = ref.play_games()
    assert stats == ref_stats
    assert dev.len() == sum(r.length for r in recs) and dev.played_games() == 64
    batch = dev.get_slice(0, recs[0].length)
    assert [e[1][1] for e in batch.as_list()] == [np.asarray(p, np.float32).tolist() for p in recs[0].child_policy]
    assert set(batch.game_index.tolist()) == {2}
    dev.close(); g.engine.close(); ref.engine.close()


@pytest.mark.parametrize("name", ["ttt", "scs", "one"])
def test_fused_loss_equals_the_reference(name):
    """calculate_loss for a whole batch in one launch vs the reference's per-sample loop: every policy loss (CEL with
    label smoothing 0.02, CEL normalised by log(batch), KLD, masked MSE) x value loss (SE, AE); losses and the gradients
    of the combined loss w.r.t. the policy logits and the values (torch autograd on the reference's functions)."""
    import torch
    from nuzero_amd.loss import calculate_loss
    kat = np.load(os.path.join(GOLDEN, "loss_kat.npz"))
    logits = torch.tensor(kat[f"{name}_logits"], device="cuda", requires_grad=True)
    values = torch.tensor(kat[f"{name}_values"], device="cuda", requires_grad=True)
    tp = torch.tensor(kat[f"{name}_target_policies"]).float().cuda()        # torch.tensor(list of floats): float32
    tv = torch.tensor(kat[f"{name}_target_values"]).float().cuda()
    for pkey, pname, norm in (("ce", "CEL", False), ("ce_norm", "CEL", True), ("kld", "KLD", False), ("mse", "MSE", False)):
        for vkey, vname in (("se", "SE"), ("ae", "AE")):
            key = f"{name}_{pkey}_{vkey}"
            if key + "_losses" not in kat:
                continue
            logits.grad = values.grad = None
            v_loss, p_loss, c_loss = calculate_loss((logits, values), tp, tv, pname, vname, norm)
            c_loss.backward()
            got = np.array([v_loss.item(), p_loss.item(), c_loss.item()])
            want = kat[key + "_losses"]
            assert np.all(np.abs(got - want) <= 2e-6 * np.abs(want) + 1e-7), (key, got, want)       # 2e-6 relative
            for g, w in ((logits.grad, kat[key + "_dlogits"]), (values.grad, kat[key + "_dvalues"])):
                scale = max(float(np.abs(w).max()), 1e-6)
                assert float(np.abs(g.cpu().numpy() - w).max()) <= 2e-6 * scale + 1e-9, key       # 2e-6 of the largest entry


def test_fused_loss_rejects_what_the_reference_cannot_compute():
    import torch
    from nuzero_amd._lib import NzError
    from nuzero_amd.loss import calculate_loss
    x, v = torch.zeros((1, 9), device="cuda"), torch.zeros((1, 1), device="cuda")
    with pytest.raises(NzError):          # log(1) = 0 in the normalisation (AlphaZero.py:912-915)
        calculate_loss((x, v), torch.full((1, 9), 1 / 9.0, device="cuda"), torch.zeros(1, device="cuda"), "CEL", "SE", True)
    with pytest.raises(KeyError):
        calculate_loss((x, v), x, v, "nope", "SE")


def test_masked_mse_with_an_all_zero_target_row():
    """loss_functions.py:7-26 divides by the count of non-zero target entries: the surface raises ZeroDivisionError as
    the reference does; at the C ABI the loss comes out NaN and the sample's gradient row is zeros, the other samples'
    gradients untouched."""
    import torch
    from ctypes import c_void_p
    from nuzero_amd import _lib
    from nuzero_amd._lib import lib
    from nuzero_amd.loss import calculate_loss
    B, A = 3, 9
    g = torch.Generator().manual_seed(5)
    x, v = torch.randn((B, A), generator=g).cuda(), torch.randn(B, generator=g).cuda()
    t = torch.softmax(torch.randn((B, A), generator=g), 1).cuda()
    tv = torch.tensor([1.0, 0.0, -1.0]).cuda()
    t_bad = t.clone()
    t_bad[1] = 0
    with pytest.raises(ZeroDivisionError):
        calculate_loss((x, v), t_bad, tv, "MSE", "SE")

    def raw(tp):
        losses, dl, dv = torch.empty(3, device="cuda"), torch.empty_like(x), torch.empty_like(v)
        work = torch.empty(2 * B, device="cuda")
        st = lib.nz_loss_forward_backward(c_void_p(x.data_ptr()), c_void_p(v.data_ptr()), c_void_p(tp.data_ptr()),
                                          c_void_p(tv.data_ptr()), B, A, _lib.NZ_LOSS_MSE, _lib.NZ_LOSS_SE, 0,
                                          c_void_p(losses.data_ptr()), c_void_p(dl.data_ptr()), c_void_p(dv.data_ptr()),
                                          c_void_p(work.data_ptr()), c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == _lib.NZ_OK
        torch.cuda.synchronize()
        return losses, dl

    good_l, good_dl = raw(t)
    bad_l, bad_dl = raw(t_bad)
    assert torch.isnan(bad_l[1]) and torch.isnan(bad_l[2]) and not torch.isnan(bad_l[0])
    assert torch.equal(bad_dl[1], torch.zeros(A, device="cuda"))
    assert torch.equal(bad_dl[0], good_dl[0]) and torch.equal(bad_dl[2], good_dl[2]) and torch.isfinite(bad_dl).all()
