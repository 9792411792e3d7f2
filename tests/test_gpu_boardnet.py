"""The board-sized policy/value network kernels (nz_boardnet_*, nuzero_amd/csrc/boardnet.hip)
against the reference's own outputs (tests/golden/net_kat3.npz) and against the oracle nets
(oracle/net.py, pinned bit-exactly to the reference by tests/test_oracle_golden.py).

Tolerance: 1e-5 absolute on softmax probabilities and values (BASELINE.json north_star);
raw logits are compared relative to the largest logit of the position.  Needs a GPU."""
import os

import numpy as np
import pytest

from conftest import NETS3, nets3_inputs, nets3_oracle, nets3_weights

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _boardnet(name, max_batch):
    from nuzero_amd.boardnet import BoardNet
    arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain = NETS3[name]
    net = BoardNet(arch, cin, planes, rows, cols, width=width, num_blocks=depth, recall=recall,
                   value_activation=vact, kernel_size=3, max_batch=max_batch)
    net.set_weights(nets3_weights(name), iters)
    return net, iters


def _check(probs, value, logits, want_probs, want_value, want_logits):
    scale = np.abs(want_logits).max(axis=1, keepdims=True) + 1.0
    assert np.max(np.abs(logits - want_logits) / scale) < TOL
    assert np.max(np.abs(probs - want_probs)) < TOL
    assert np.max(np.abs(value - want_value)) < TOL
    assert np.allclose(probs.sum(axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("name", ["G", "H", "I", "J", "K", "L"])
def test_boardnet_equals_reference_outputs(net_kat3, name):
    import torch
    net, _ = _boardnet(name, 16)
    x = torch.from_numpy(nets3_inputs(name)).cuda()
    probs, value, logits = net.forward(x, want_logits=True)
    _check(probs.cpu().numpy(), value.cpu().numpy(), logits.cpu().numpy(),
           net_kat3[f"{name}_probs"], net_kat3[f"{name}_value"], net_kat3[f"{name}_logits"])
    net.close()


@pytest.mark.parametrize("name,n", [("G", 333), ("H", 70), ("I", 41), ("J", 1500)])
def test_boardnet_batches_equal_oracle(name, n):
    """Ragged batch sizes (rows not a multiple of the 16/64-row tiles) and both row-tile shapes."""
    import torch
    from scipy.special import softmax
    net, iters = _boardnet(name, n)
    x = nets3_inputs(name, n, offset=7)
    probs, value, logits = net.forward(torch.from_numpy(x).cuda(), want_logits=True)
    p, v = nets3_oracle(name).inference(x, iters)
    p = p.reshape(n, -1)
    _check(probs.cpu().numpy(), value.cpu().numpy(), logits.cpu().numpy(), softmax(p, axis=1), v.reshape(-1), p)
    assert net.flops_per_position > 0
    net.close()


def test_boardnet_device_side_batch_count():
    """The live batch size can stay on the device (the leaf count of a simulation wave)."""
    import torch
    net, _ = _boardnet("G", 64)
    x = torch.from_numpy(nets3_inputs("G", 64, offset=3)).cuda()
    full_p, full_v = net.forward(x)
    n_dev = torch.tensor([37], dtype=torch.int32, device="cuda")
    from ctypes import c_void_p
    from nuzero_amd._lib import lib
    probs = torch.full((64, net.num_actions), -1.0, dtype=torch.float32, device="cuda")
    value = torch.full((64,), -7.0, dtype=torch.float32, device="cuda")
    st = lib.nz_boardnet_forward(net._h, c_void_p(x.data_ptr()), 64, c_void_p(n_dev.data_ptr()), None,
                                 c_void_p(probs.data_ptr()), c_void_p(value.data_ptr()),
                                 c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    assert torch.equal(probs[:37], full_p[:37]) and torch.equal(value[:37], full_v[:37])
    assert (probs[37:] == -1.0).all() and (value[37:] == -7.0).all()
    net.close()


def test_boardnet_rejects_bad_arguments():
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd._lib import NzError
    import torch
    with pytest.raises(NzError):
        BoardNet("convnet", 86, 21, 5, 5, width=32, num_blocks=2, kernel_size=5)
    net = BoardNet("resnet", 86, 21, 5, 5, width=32, num_blocks=2, max_batch=4)
    with pytest.raises(NzError):          # forward before weights
        net.forward(torch.zeros((1, 86, 5, 5), device="cuda"))
    with pytest.raises(NzError):          # wrong tensor count
        net.set_weights({"a": np.zeros((32, 86, 3, 3), np.float32)})
    net.close()


def test_scs_selfplay_with_the_native_network():
    """SCS self-play with tree, rules AND network on the device (no PyTorch model in the loop):
    the first root's priors are the oracle net's softmax over the legal moves of the opening
    position, and every game replays legally through the oracle rules."""
    import torch
    from scipy.special import softmax
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from nuzero_amd.weights import synthetic_weights, resnet_param_shapes
    from oracle.net import FeedForwardRef
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scs_configs", "mirrored_5x5.yml")
    cfg = ScsGameConfig(path)
    w = synthetic_weights(5, resnet_param_shapes(cfg.channels, cfg.planes, 32, 2), 2.0)
    net = BoardNet("resnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=8)
    net.set_weights(w)
    search = {"Simulation": {"mcts_simulations": 16, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.0,
                              "root_dist_alpha": 0.2, "root_dist_beta": 1}}
    sp = ScsSelfPlay(cfg, search, 8)
    r = sp.play(net.evaluator(), seeds=range(40, 48))
    ocfg = ScsConfig(path)
    og = ScsGame(ocfg)
    p, _ = FeedForwardRef(w, "resnet", 2).inference(og.state_image(), None)
    mask = og.possible_actions().reshape(-1)
    pri = softmax(p).reshape(-1) * mask
    pri = pri[mask > 0] / pri.sum()
    k = r["n_children"][0, 0]
    assert k == int(mask.sum())
    assert np.max(np.abs(r["child_prior"][0, 0, :k] - pri)) < TOL      # fraction 0: noise leaves the priors alone
    for g in range(8):
        og = ScsGame(ocfg)
        for m in range(r["lengths"][g]):
            legal = np.nonzero(og.possible_actions().reshape(-1))[0]
            kk = r["n_children"][g, m]
            assert r["child_action"][g, m, :kk].tolist() == legal.tolist()
            og.step_index(int(r["actions"][g, m]))
        assert og.terminal and og.terminal_value == r["outcomes"][g]
    assert r["expansions"] == sp.evaluations
    sp.close()
    net.close()


def test_native_move_loop_plays_the_same_games_as_the_lockstep_api():
    """nz_scs_search_play (move loop in the library, leaf count on the device, terminal-simulation
    budget per wave) against ScsSelfPlay.play driving the same kernels wave by wave from Python:
    identical roots, visit counts, priors and value sums for every move of every game."""
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsSelfPlay, ScsGameConfig
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scs_configs", "late_reinforcements_5x5.yml")
    cfg = ScsGameConfig(path)
    w = synthetic_weights(9, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 2), 2.0)
    net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=2, max_batch=24)
    net.set_weights(w)
    search = {"Simulation": {"mcts_simulations": 40, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 6, "epsilon_softmax_exploration": 0.1,
                              "epsilon_random_exploration": 0.05, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.25,
                              "root_dist_alpha": 0.3, "root_dist_beta": 1}}
    seeds = list(range(500, 524))
    a = ScsSelfPlay(cfg, search, 24)
    ra = a.play(net.evaluator(), seeds)
    b = ScsSelfPlay(cfg, search, 24)
    rb = b.play_native(net, seeds)
    assert (ra["lengths"] > 10).all()
    for k in ("lengths", "outcomes", "actions", "tree_size", "n_children"):
        assert np.array_equal(ra[k], rb[k]), k
    for g in range(24):                       # records past a game's end / a root's children are not defined
        n = ra["lengths"][g]
        for k in ("bias", "root_value_sum"):
            assert np.array_equal(ra[k][g, :n], rb[k][g, :n]), (k, g)
        for m in range(n):
            c = ra["n_children"][g, m]
            for k in ("child_action", "child_visit", "child_prior", "child_value_sum"):
                assert np.array_equal(ra[k][g, m, :c], rb[k][g, m, :c]), (k, g, m)
    assert ra["expansions"] == rb["expansions"] and ra["simulations"] == rb["simulations"]
    assert rb["waves"] > 0
    a.close(); b.close(); net.close()


@pytest.mark.parametrize("arch,cin,planes,rows,cols,width,depth,recall,vact,iters,n", [
    ("recurrent", 86, 21, 5, 5, 32, 2, True, "relu", 2, 37),
    ("resnet", 105, 30, 6, 5, 48, 2, False, "tanh", 1, 50),
    ("convnet", 86, 21, 10, 10, 32, 3, False, "tanh", 1, 19)])
def test_hexagonal_nets_equal_the_oracle(arch, cin, planes, rows, cols, width, depth, recall, vact, iters, n):
    """hex=True nets (every conv a hexagdly.Conv2d(kernel_size=1): two weight tensors per conv, 7 taps with
    column-parity offsets) against oracle/net.py HexNetRef.  PARITY UNPINNED: hexagdly is not installed, the
    oracle restates its documented addressing (tests/test_hex_oracle.py ties it to the game's adjacency)."""
    import torch
    from scipy.special import softmax
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, recurrent_net_param_shapes, resnet_param_shapes,
                                    convnet_param_shapes)
    from oracle.net import HexNetRef
    if arch == "recurrent":
        shapes = recurrent_net_param_shapes(cin, planes, width, depth, recall)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cin, planes, width, depth)
    else:
        shapes = convnet_param_shapes(cin, planes, 3, width, depth)
    w = synthetic_weights(21, hex_param_shapes(shapes), 2.0)
    rs = np.random.RandomState(5)
    x = (rs.random_sample((n, cin, rows, cols)) < 0.15).astype(np.float32)
    x[:, -3:] = rs.random_sample((n, 3, rows, cols)).astype(np.float32)
    net = BoardNet(arch, cin, planes, rows, cols, width=width, num_blocks=depth, recall=recall, value_activation=vact,
                   max_batch=n, hex=True)
    net.set_weights(w, iters)
    probs, value, logits = net.forward(torch.from_numpy(x).cuda(), want_logits=True)
    p, v = HexNetRef(w, arch, depth, recall, vact).inference(x, iters)
    p = p.reshape(n, -1)
    _check(probs.cpu().numpy(), value.cpu().numpy(), logits.cpu().numpy(), softmax(p, axis=1), v.reshape(-1), p)
    net.close()


@pytest.mark.parametrize("arch,cin,planes,rows,cols,width,depth,recall,vact,iters,n,hexnet", [
    ("recurrent", 86, 21, 5, 5, 128, 1, True, "relu", 2, 1560, False),    # two K sources (recall), residuals
    ("resnet", 86, 21, 5, 5, 256, 1, False, "tanh", 1, 790, False),       # two tiles of 128 channels, ragged tile of positions
    ("convnet", 86, 21, 10, 10, 128, 2, False, "tanh", 1, 260, True)])    # hexagonal taps (unpinned)
def test_wide_layers_equal_the_oracle(arch, cin, planes, rows, cols, width, depth, recall, vact, iters, n, hexnet):
    """Layers whose width is a multiple of 128 run on conv_wide_kernel (float32 products from six bf16 MFMA terms,
    LDS-staged 256-position x 128-channel tiles) when there are at least 160 such tiles -- the batch sizes here are
    chosen so (ragged last tile included): same 1e-5 tolerance."""
    import torch
    from scipy.special import softmax
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, recurrent_net_param_shapes, resnet_param_shapes,
                                    convnet_param_shapes)
    from oracle.net import FeedForwardRef, HexNetRef, RecurrentNetRef
    if arch == "recurrent":
        shapes = recurrent_net_param_shapes(cin, planes, width, depth, recall)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cin, planes, width, depth)
    else:
        shapes = convnet_param_shapes(cin, planes, 3, width, depth)
    w = synthetic_weights(31, hex_param_shapes(shapes) if hexnet else shapes, 2.0)
    rs = np.random.RandomState(6)
    x = (rs.random_sample((n, cin, rows, cols)) < 0.15).astype(np.float32)
    x[:, -3:] = rs.random_sample((n, 3, rows, cols)).astype(np.float32)
    net = BoardNet(arch, cin, planes, rows, cols, width=width, num_blocks=depth, recall=recall, value_activation=vact,
                   max_batch=n, hex=hexnet)
    net.set_weights(w, iters)
    probs, value, logits = net.forward(torch.from_numpy(x).cuda(), want_logits=True)
    if hexnet:
        ref = HexNetRef(w, arch, depth, recall, vact)
    elif arch == "recurrent":
        ref = RecurrentNetRef(w, cin, planes, width, depth, recall, vact)
    else:
        ref = FeedForwardRef(w, arch, depth, vact)
    p, v = ref.inference(x, iters)
    p = p.reshape(n, -1)
    _check(probs.cpu().numpy(), value.cpu().numpy(), logits.cpu().numpy(), softmax(p, axis=1), v.reshape(-1), p)
    net.close()


@pytest.mark.parametrize("arch,hexnet,rows,cols,width,depth,n", [
    ("convnet", False, 5, 5, 32, 8, 1024), ("convnet", True, 5, 5, 32, 8, 333), ("resnet", False, 5, 5, 32, 2, 37),
    ("recurrent", False, 5, 5, 32, 2, 640), ("convnet", False, 10, 10, 32, 3, 70), ("resnet", False, 6, 5, 48, 2, 100),
    # more positions than fit in LDS at an even share per CU: the BF16 one-launch grid is larger than the CU count
    ("convnet", False, 5, 5, 32, 8, 2500), ("resnet", False, 5, 5, 32, 6, 1500)])
def test_one_launch_network_equals_the_per_layer_kernels(arch, hexnet, rows, cols, width, depth, n):
    """The one-launch network (all layers + softmax + value in one launch, activations in LDS, rows = (position, cell))
    against the per-layer kernels: fused_net_kernel gives the SAME floats (same MFMA, same K order; an off-board tap adds
    an exact zero), fused16_net_kernel (ConvNets, ResNets) the same within 5e-6; at ragged batch sizes, with the batch size in
    device memory, for every architecture; and against the oracle within 1e-5."""
    import torch
    from scipy.special import softmax
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, recurrent_net_param_shapes, resnet_param_shapes,
                                    convnet_param_shapes)
    from oracle.net import FeedForwardRef, HexNetRef, RecurrentNetRef
    cin, planes, iters = 86, 21, 2
    if arch == "recurrent":
        shapes = recurrent_net_param_shapes(cin, planes, width, depth, True)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cin, planes, width, depth)
    else:
        shapes = convnet_param_shapes(cin, planes, 3, width, depth)
    w = synthetic_weights(23, hex_param_shapes(shapes) if hexnet else shapes, 2.0)
    rs = np.random.RandomState(8)
    x = (rs.random_sample((n, cin, rows, cols)) < 0.15).astype(np.float32)
    x[:, -3:] = rs.random_sample((n, 3, rows, cols)).astype(np.float32)
    net = BoardNet(arch, cin, planes, rows, cols, width=width, num_blocks=depth, recall=True, max_batch=n, hex=hexnet)
    net.set_weights(w, iters)
    assert net.fused()                                   # the one-launch form exists for these shapes
    xd = torch.from_numpy(x).cuda()
    pf, vf, lf = net.forward(xd, want_logits=True)
    net.fused(False)
    pu, vu, lu = net.forward(xd, want_logits=True)
    if arch in ("convnet", "resnet"):
        # ConvNets and ResNets run the one-launch form on the BF16 matrix cores (three-way split, six of nine piece products): the
        # arithmetic of the Tic-Tac-Toe network, as accurate as float32 but not the per-layer kernels' rounding
        scale = lu.abs().amax(dim=1, keepdim=True) + 1.0
        assert float(((lf - lu).abs() / scale).max()) < 5e-6 and float((pf - pu).abs().max()) < 5e-6
        assert float((vf - vu).abs().max()) < 5e-6
    else:
        assert torch.equal(lf, lu) and torch.equal(pf, pu) and torch.equal(vf, vu)
    net.fused(True)
    m = max(1, n // 3)                                   # live batch size on the device, smaller than the launch
    n_dev = torch.tensor([m], dtype=torch.int32, device="cuda")
    p2, v2 = net.forward(xd, n_dev=n_dev)
    assert torch.equal(p2[:m], pf[:m]) and torch.equal(v2[:m], vf[:m])
    k = min(n, 24)
    if hexnet:
        ref = HexNetRef(w, arch, depth, True, "tanh")
    elif arch == "recurrent":
        ref = RecurrentNetRef(w, cin, planes, width, depth, True, "tanh")
    else:
        ref = FeedForwardRef(w, arch, depth, "tanh")
    p, v = ref.inference(x[:k], iters)
    _check(pf[:k].cpu().numpy(), vf[:k].cpu().numpy(), lf[:k].cpu().numpy(), softmax(p.reshape(k, -1), axis=1), v.reshape(-1),
           p.reshape(k, -1))
    net.close()
