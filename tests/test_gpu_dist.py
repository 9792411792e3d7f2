"""The RCCL path of the multi-GPU round on the one GPU a test box has: a process group of ONE rank on the `nccl`
backend runs the same collectives `bench.py --gpus N` runs (weights broadcast, one gather of the round's export
buffers, append of the gathered games to rank 0's DeviceReplayBuffer), forced past their single-process shortcuts.
World size 2 is covered on gloo by tests/test_host_logic.py.  Needs a GPU."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_group_of_one():
    import torch
    import torch.distributed as td
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield td
    finally:
        td.destroy_process_group()


def test_rccl_broadcast_gather_and_append_on_one_rank(nccl_group_of_one):
    import torch
    from nuzero_amd import dist as nzdist
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.replay_device import DeviceReplayBuffer
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    td = nccl_group_of_one
    assert td.get_backend() == "nccl" and td.get_world_size() == 1
    dev = torch.device("cuda", 0)
    w = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True)
    got = nzdist.broadcast_weights(w, src=0, device=dev, force_collective=True)          # td.broadcast on RCCL
    assert list(got) == list(w) and all(v.is_cuda for v in got.values())
    assert all(np.array_equal(got[k].cpu().numpy(), w[k]) for k in w)

    G = 96
    eng = SelfPlayEngine(legacy_ttt_search_config(25), G, training=True, device=0)
    eng.set_weights(got, recurrent_iterations=2)
    eng.play(base_seed=31000)
    shape = dict(window_size=4 * G, batch_size=64, state_shape=(2, 3, 3), num_actions=9, max_game_length=9, device=0)
    shared, direct = DeviceReplayBuffer(**shape), DeviceReplayBuffer(**shape)
    rg = nzdist.ReplayGather(eng, 1, 0, buffer=shared, game_index=2, force_collective=True)   # td.gather on RCCL
    out = rg.gather()
    assert rg.ranks_seen == 1 and rg.games_saved == G
    own = eng.export_device()
    for k in nzdist.FIELDS:
        assert torch.equal(out[k], own[k]), k
    # the gathered games in the shared buffer == the same round saved without any collective
    direct.save_games_from_engine(eng, 2, export=own)
    n = direct.len()
    assert shared.len() == n == int(own["lengths"].sum()) and shared.played_games() == G
    a, b = shared.get_slice(0, n), direct.get_slice(0, n)
    for x, y in ((a.states, b.states), (a.policies, b.policies), (a.values, b.values), (a.game_index, b.game_index)):
        assert torch.equal(x, y)
    assert int(a.game_index[0]) == 2
    shared.close(); direct.close(); eng.close()
