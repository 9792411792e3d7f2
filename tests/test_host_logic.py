"""Host-side pieces that need no GPU: replay buffer window, payload packing, the
world_size-2 gather over gloo, search-config conversion."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Rec:
    def __init__(self, n, tag):
        self.state_history = [torch.full((1, 2, 3, 3), float(tag * 100 + i)) for i in range(n)]
        self.n, self.tag = n, tag

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        return (self.tag, [i / 9.0] * 9)


def test_replay_buffer_window():
    """Training/ReplayBuffer.py:24-36: the window counts games, but once full one
    oldest POSITION is dropped per position added."""
    from nuzero_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(window_size=2, batch_size=4)
    rb.save_game(_Rec(5, 1), 0)
    rb.save_game(_Rec(7, 2), 0)
    assert rb.len() == 12 and rb.played_games() == 2
    rb.save_game(_Rec(3, 3), 1)          # full: 3 in, 3 oldest out
    assert rb.len() == 12 and rb.played_games() == 2
    firsts = [float(e[0].flatten()[0]) for e in rb.get_buffer()]
    assert firsts[0] == 103.0 and firsts[-1] == 302.0
    state, (value, policy), idx = rb.get_buffer()[-1]
    assert state.shape == (1, 2, 3, 3) and state.dtype == torch.float32 and value == 3 and len(policy) == 9 and idx == 1
    batch = rb.get_sample(4, False, [])
    assert len(batch) == 4
    assert len(rb.get_slice(2, 5)) == 3


def test_search_config_struct():
    from nuzero_amd.search_config import legacy_ttt_search_config, to_struct
    c = to_struct(legacy_ttt_search_config(25), True)
    assert (c.mcts_simulations, c.keep_subtree, c.training) == (25, 1, 1)
    assert (c.pb_c_base, c.pb_c_init, c.root_dist_alpha, c.root_exploration_fraction) == (5000.0, 1.15, 0.15, 0.2)
    bad = legacy_ttt_search_config()
    bad["Exploration"]["root_exploration_distribution"] = "dirichlet"
    with pytest.raises(ValueError):
        to_struct(bad, True)


def _payload(rank, g=6, t=9, a=9):
    rs = np.random.RandomState(rank)
    return {
        "states": torch.from_numpy(rs.rand(g, t, 2, 3, 3).astype(np.float32)),
        "visits": torch.from_numpy(rs.randint(0, 100, (g, t, a)).astype(np.int32)),
        "actions": torch.from_numpy(rs.randint(-1, 9, (g, t)).astype(np.int32)),
        "lengths": torch.from_numpy(rs.randint(5, 10, (g,)).astype(np.int32)),
        "outcomes": torch.from_numpy(rs.randint(-1, 2, (g,)).astype(np.int32)),
        "tree_size": torch.from_numpy(rs.randint(0, 900, (g, t)).astype(np.int32)),
        "n_children": torch.from_numpy(rs.randint(0, 9, (g, t)).astype(np.int32)),
        "bias": torch.from_numpy(rs.rand(g, t)),
    }


def test_pack_unpack_roundtrip():
    from nuzero_amd import dist as nzdist
    p = _payload(3)
    buf, layout = nzdist.pack(p)
    assert buf.dtype == torch.uint8 and buf.numel() % 16 == 0
    q = nzdist.unpack(buf, layout)
    for k in nzdist.FIELDS:
        assert q[k].dtype == p[k].dtype and torch.equal(q[k], p[k])


def _gather_worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from nuzero_amd import dist as nzdist
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = nzdist.gather_payload(_payload(rank), world, rank, dst=0)
        if rank == 0:
            ok = True
            for k in nzdist.FIELDS:
                want = torch.cat([_payload(r)[k] for r in range(world)], 0)
                ok = ok and torch.equal(got[k], want)
            ret["ok"] = ok
        else:
            ret[f"none{rank}"] = got is None
        assert nzdist.shard_seeds(1000, 512, rank) == 1000 + 512 * rank
    finally:
        td.destroy_process_group()


def test_gather_world_size_2_gloo():
    """The once-per-round collection of finished games, two ranks on CPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get("ok") is True and ret.get("none1") is True


def _scs_result(rank, n=5):
    rs = np.random.RandomState(100 + rank)
    return {"lengths": rs.randint(10, 60, n).astype(np.int32), "outcomes": rs.randint(-1, 2, n).astype(np.int32),
            "actions": rs.randint(-1, 525, (n, 256)).astype(np.int32), "n_children": rs.randint(0, 40, (n, 256)).astype(np.int32),
            "child_action": rs.randint(0, 525, (n, 256, 64)).astype(np.int32),
            "child_visit": rs.randint(0, 200, (n, 256, 64)).astype(np.int32),
            "bias": rs.random_sample((n, 256))}            # not part of the gathered fields


def _scs_gather_worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from nuzero_amd import dist as nzdist
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = nzdist.gather_payload(nzdist.scs_payload(_scs_result(rank)), world, rank, dst=0, fields=nzdist.SCS_FIELDS)
        if rank == 0:
            ok = set(got.keys()) == set(nzdist.SCS_FIELDS)
            for k in nzdist.SCS_FIELDS:
                want = np.concatenate([_scs_result(r)[k] for r in range(world)], 0)
                ok = ok and np.array_equal(got[k].numpy(), want)
            ret["ok"] = ok
        else:
            ret[f"none{rank}"] = got is None
    finally:
        td.destroy_process_group()


def test_scs_gather_world_size_2_gloo():
    """SCS rounds shard by game like Tic-Tac-Toe rounds: one gather of moves and policy targets per round."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_scs_gather_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get("ok") is True and ret.get("none1") is True


def test_game_record_contract():
    """What ReplayBuffer.save_game and AlphaZero.batch_update_weights touch."""
    from nuzero_amd.gamer import GameRecord, game_stats
    states = np.zeros((9, 2, 3, 3), np.float32)
    states[1, 0, 1, 1] = 1
    visits = np.zeros((9, 9), np.int32)
    visits[0] = [11, 11, 11, 11, 11, 11, 11, 11, 11]
    visits[1] = [0, 30, 40, 0, 0, 57, 0, 0, 0]
    rec = GameRecord(states, visits, np.array([4, 5, -1, -1, -1, -1, -1, -1, -1]), 2, -1)
    assert len(rec.state_history) == 2 and rec.get_state_from_history(1).shape == (1, 2, 3, 3)
    assert rec.get_state_from_history(1).dtype == torch.float32
    v, pol = rec.make_target(1)
    assert v == -1 and pol == [0, 30 / 127, 40 / 127, 0, 0, 57 / 127, 0, 0, 0]
    assert torch.cat([rec.get_state_from_history(i) for i in range(2)], 0).shape == (2, 2, 3, 3)
    r = {"lengths": np.array([2]), "tree_size": np.array([[99, 127] + [0] * 7]),
         "n_children": np.array([[9, 8] + [0] * 7]), "bias": np.array([[1.17, 1.18] + [0.0] * 7])}
    st = game_stats(r, 0)
    assert st == {"number_of_moves": 2, "average_children": 8.5, "average_tree_size": 113.0, "final_tree_size": 127,
                  "average_bias_value": (1.17 + 1.18) / 2, "final_bias_value": 1.18}


def test_network_manager_follows_in_place_weight_updates():
    """The reference's trainer trains one long-lived module in place and hands the same Network_Manager to the
    Gamers after every step (Training/AlphaZero.py:152,293,462): the weights the engine uses must follow.
    Host logic only (no GPU): the snapshot and its version."""
    import torch
    from conftest import named_weights_module
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    w0 = synthetic_recurrent_net_weights(1, 2, 1, 16, 2, True)
    w1 = synthetic_recurrent_net_weights(2, 2, 1, 16, 2, True)
    model = named_weights_module(w0)
    nm = Network_Manager(model)
    v0 = nm.sync()
    assert nm.sync() == v0 and nm.is_recurrent()                  # nothing written: no new snapshot
    assert all(np.array_equal(nm.state_dict()[k], w0[k]) for k in w0)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    opt.step()                                                     # an optimizer step writes in place
    assert nm.sync() == v0 + 1
    assert all(np.array_equal(nm.state_dict()[k], w0[k] - 0.5) for k in w0)
    with torch.no_grad():                                          # load_state_dict-style copy
        for p, v in zip(model.parameters(), w1.values()):
            p.copy_(torch.from_numpy(v))
    assert nm.sync() == v0 + 2 and nm.sync() == v0 + 2
    assert all(np.array_equal(nm.state_dict()[k], w1[k]) for k in w1)
    d = Network_Manager(dict(w0))                                  # plain dicts are data: explicit refresh()
    assert d.sync() == d.sync()
    d.model["projection.0.weight"] = w1["projection.0.weight"]
    assert d.refresh() == d.version and np.array_equal(d.state_dict()["projection.0.weight"], w1["projection.0.weight"])
    nm.model_to_cpu(); nm.model_to_device(); nm.check_devices()   # surface of Network_Manager.py:32-44


def test_replay_buffer_checkpoint_in_the_reference_layout(tmp_path):
    """save_to_file / load_from_file use the reference's layout {'buffer', 'map', 'partial_loading'} and its loading
    rule by training step (ReplayBuffer.py:64-107); a file laid out the way the reference's save_to_file writes it
    loads (with torch.load(weights_only=True))."""
    from nuzero_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(window_size=3, batch_size=4)
    rb.save_game(_Rec(4, 1), 0)
    rb.save_to_file(tmp_path / "a.pt", step=10)
    rb.save_game(_Rec(2, 2), 1)
    rb.save_to_file(tmp_path / "a.pt", step=20)
    raw = torch.load(tmp_path / "a.pt", weights_only=True)
    assert set(raw) == {"buffer", "map", "partial_loading"} and raw["map"] == {10: (4, 1), 20: (6, 2)} and raw["partial_loading"]
    other = ReplayBuffer(3, 4)
    other.load_from_file(tmp_path / "a.pt", step=10)
    assert other.len() == 5 and other.played_games() == 1          # buffer[:buffer_len + 1], as the reference slices
    other.load_from_file(tmp_path / "a.pt", step=20)
    assert other.len() == 6 and other.played_games() == 2
    with pytest.raises(Exception):
        other.load_from_file(tmp_path / "a.pt", step=15)
    # a reference-written file: plain containers, tensors and numbers
    entry = lambda g, m: (torch.full((1, 2, 3, 3), float(10 * g + m)), (1, [m / 9.0] * 9), g % 2)
    torch.save({"buffer": [entry(g, m) for g in range(4) for m in range(3)], "map": {5: (6, 2), 9: (12, 3)},
                "partial_loading": False}, tmp_path / "ref.pt")
    other.load_from_file(tmp_path / "ref.pt", step=5)              # partial loading off: the whole latest buffer
    assert other.len() == 12 and other.played_games() == 3 and other.full
    state, (value, policy), idx = other.get_buffer()[-1]
    assert float(state.flatten()[0]) == 32.0 and value == 1 and len(policy) == 9 and idx == 1
    # window full -> partial loading is switched off for good
    rb.save_game(_Rec(2, 3), 0)
    rb.save_game(_Rec(2, 4), 0)
    rb.save_to_file(tmp_path / "b.pt", step=30)
    assert torch.load(tmp_path / "b.pt", weights_only=True)["partial_loading"] is False


def _replay_kat():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay_kat.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["small_window", "never_full", "window_1", "long"])
def test_replay_index_equals_the_reference(name):
    """The host side of the device replay buffer (order, window in games with per-position eviction, random.shuffle,
    slices, np.random.choice samples incl. the late_heavy weights, grouping by game index) against traces of the
    GENUINE ReplayBuffer class (tests/golden/make_golden_replay.py): same (game, move) identities in the same order."""
    import random
    from nuzero_amd.replay_device import ReplayIndex, late_heavy_probs
    case = _replay_kat()[name]
    idx = ReplayIndex(case["window"])
    ident = {}                       # physical slot -> (game, move)
    for op in case["ops"]:
        if op["op"] == "save":
            g = op["game"]
            dst = idx.save_games([case["lengths"][g]], case["game_types"][g])
            for m, slot in enumerate(dst[0]):
                ident[int(slot)] = [g, m]
            assert len(idx) == op["len"] and idx.n_games == op["played"] and idx.full == op["full"]
            assert [ident[int(s)] for s in idx.seq] == op["order"]
        elif op["op"] == "shuffle":
            random.seed(op["seed"])
            idx.shuffle()
            assert [ident[int(s)] for s in idx.seq] == op["order"]
        elif op["op"] == "slice":
            got = idx.get_slice(op["start"], op["stop"])
            assert [ident[int(s)] + [int(idx.slot_game_index[s])] for s in got] == op["got"]
        elif op["op"] == "sample":
            probs = late_heavy_probs(len(idx)) if op["late_heavy"] else []
            np.random.seed(op["seed"])
            got = idx.get_sample(op["batch_size"], op["replace"], probs)
            assert [ident[int(s)] + [int(idx.slot_game_index[s])] for s in got] == op["got"]
        elif op["op"] == "bucket":
            np.random.seed(op["seed"])
            slots, keys, counts = idx.bucket(idx.get_sample(op["batch_size"], True, []))
            assert keys == op["keys"] and counts == [len(g) for g in op["groups"]]
            flat = [e for g in op["groups"] for e in g]
            assert [ident[int(s)] + [int(idx.slot_game_index[s])] for s in slots] == flat


def test_replay_index_batches_of_games_equal_one_by_one():
    """save_games(many lengths) == save_games([length]) game after game, also when the window fills inside the batch and
    when a batch is larger than the whole buffer."""
    from nuzero_amd.replay_device import ReplayIndex
    rs = np.random.RandomState(0)
    for window in (1, 3, 8, 40):
        a, b = ReplayIndex(window), ReplayIndex(window)
        ida, idb = {}, {}
        g0 = 0
        for _ in range(6):
            lengths = rs.randint(1, 8, size=int(rs.randint(1, 12)))
            da = a.save_games(lengths, 0)
            stored = da[da >= 0]
            assert len(np.unique(stored)) == len(stored)          # one writer per slot: the batch's writes may run in any order
            for g, row in enumerate(da):
                for m, s in enumerate(row):
                    if s >= 0:
                        ida[int(s)] = (g0 + g, m)
            for g, n in enumerate(lengths):
                for m, s in enumerate(b.save_games([n], 0)[0]):
                    idb[int(s)] = (g0 + g, m)
            g0 += len(lengths)
            assert [ida[int(s)] for s in a.seq] == [idb[int(s)] for s in b.seq]
            assert (a.n_games, a.full) == (b.n_games, b.full)


def _round_worker(rank, world, port, ret):
    """One self-play round on two ranks as bench.py --gpus 2 runs it, with CPU stand-ins for the engines: weights handed
    over by one broadcast, games sharded by rank (disjoint seeds), rank-DEPENDENT game lengths, one gather, replay buffer
    order on rank 0."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from nuzero_amd import dist as nzdist
    from nuzero_amd.replay_device import ReplayIndex
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = synthetic_recurrent_net_weights(7, 2, 1, 16, 2, True) if rank == 0 else None
        got = nzdist.broadcast_weights(w, src=0)
        want = synthetic_recurrent_net_weights(7, 2, 1, 16, 2, True)
        ok = list(got) == list(want) and all(np.array_equal(got[k].numpy(), want[k]) for k in want)
        G, T = 6, 9
        seed0 = nzdist.shard_seeds(1000, G, rank)
        seeds = [seed0 + g for g in range(G)]
        rs = np.random.RandomState(seed0)
        lengths = rs.randint(5 + 2 * rank, 8 + 2 * rank, size=G).astype(np.int32)      # rank 1 plays longer games
        p = _payload(rank, G, T)
        p["lengths"] = torch.from_numpy(lengths)
        p["states"][:, :, 0, 0, 0] = torch.tensor(seeds, dtype=torch.float32)[:, None]   # tag every position with its seed
        out = nzdist.gather_payload(p, world, rank, dst=0)
        all_seeds = [None] * world
        td.all_gather_object(all_seeds, seeds)
        flat = [s for part in all_seeds for s in part]
        ok = ok and sorted(flat) == list(range(1000, 1000 + world * G)) and len(set(flat)) == world * G
        if rank == 0:
            tags = out["states"][:, 0, 0, 0, 0].tolist()
            ok = ok and tags == [float(s) for s in flat]                                # rank-major order
            idx = ReplayIndex(window_size=100)
            dst = idx.save_games(out["lengths"].numpy(), 0)
            ok = ok and len(idx) == int(out["lengths"].sum()) and (dst >= 0).sum(1).tolist() == out["lengths"].tolist()
            ok = ok and out["lengths"][G:].float().mean() > out["lengths"][:G].float().mean()
            ret["ok"] = bool(ok)
        else:
            ret["ok1"] = bool(ok) and out is None
    finally:
        td.destroy_process_group()


def test_round_on_two_ranks_gloo():
    """Weights broadcast + sharded seeds + ragged gather + replay order, world size 2 on CPU (the N > 1 path of bench.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_round_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get("ok") is True and ret.get("ok1") is True


def test_broadcast_weights_single_process():
    from nuzero_amd import dist as nzdist
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    w = synthetic_recurrent_net_weights(3, 2, 1, 16, 2, True)
    got = nzdist.broadcast_weights(w)
    assert list(got) == list(w) and all(np.array_equal(got[k].numpy(), w[k]) for k in w)


class _FakeEngine:
    def __init__(self, payload):
        self.payload = payload

    def export_device(self):
        return self.payload


class _IndexBuffer:
    """Host stand-in for DeviceReplayBuffer.save_games_from_engine (the order logic is the real ReplayIndex)."""

    def __init__(self):
        from nuzero_amd.replay_device import ReplayIndex
        self.index, self.tags = ReplayIndex(window_size=100), []

    def save_games_from_engine(self, engine, game_index, export=None):
        self.index.save_games(export["lengths"].numpy(), game_index)
        self.tags += export["states"][:, 0, 0, 0, 0].tolist()


def _replay_gather_worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from nuzero_amd import dist as nzdist
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G, T = 5, 9
        buf = _IndexBuffer() if rank == 0 else None
        total = 0
        for rnd in range(2):
            p = _payload(rank, G, T)
            p["lengths"] = torch.full((G,), 5 + rank + rnd, dtype=torch.int32)
            p["states"][:, :, 0, 0, 0] = float(100 * rnd + 10 * rank)
            if rnd == 0:
                rg = nzdist.ReplayGather(_FakeEngine(p), world, rank, buffer=buf, game_index=3)
            rg.engine = _FakeEngine(p)
            out = rg.gather()
            total += G * (5 + rnd) + G * (6 + rnd)
            if rank == 0:
                assert rg.ranks_seen == world and rg.games_saved == (rnd + 1) * world * G
                assert len(buf.index) == total
            else:
                assert out is None and rg.ranks_seen == 0
        if rank == 0:
            ret["ok"] = buf.tags == [0.0] * G + [10.0] * G + [100.0] * G + [110.0] * G
        else:
            ret["ok1"] = True
    finally:
        td.destroy_process_group()


def test_replay_gather_fills_the_buffer_on_two_ranks_gloo():
    """ReplayGather with a buffer: every round's gathered games reach rank 0's replay buffer, rank-major, and the gather
    reports how many ranks it saw (the N > 1 product path of bench.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_replay_gather_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get("ok") is True and ret.get("ok1") is True


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no WORLD_SIZE): the parent starts the two ranks itself before touching a GPU;
    on a node with fewer GPUs each rank says so and the command exits non-zero -- not with a 'use a launcher' message."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs: the command would run the benchmark")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--no-extras",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "2 GPUs requested" in r.stderr and "launch with torch.distributed.run" not in r.stderr


def test_gamer_without_records_refuses_a_host_replay_buffer():
    """records=False with a buffer that is filled from record objects would drop every game silently."""
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.replay_buffer import ReplayBuffer
    from nuzero_amd.search_config import legacy_ttt_search_config

    class tic_tac_toe:
        pass

    class SCS_Game:
        pass

    for game, args in ((tic_tac_toe, []), (SCS_Game, ["x.yml"])):
        with pytest.raises(ValueError, match="records=False"):
            Gamer(ReplayBuffer(10, 4), None, game, args, 0, legacy_ttt_search_config(25), 2, num_games=4, records=False)


def test_action_index_division_by_multiply_high():
    """scs_step_wave splits an action index into (plane, tile) with one multiply-high by ceil(2^32 / tiles)
    (scs_dev.hpp): exact for every board of <= 100 tiles and every action index the engine admits (21 planes x tiles)."""
    for tiles in range(1, 101):
        magic = 0xFFFFFFFF // tiles + 1
        a = np.arange(0, 21 * 100 + 1, dtype=np.uint64)
        assert np.array_equal((a * np.uint64(magic)) >> np.uint64(32), a // np.uint64(tiles)), tiles
