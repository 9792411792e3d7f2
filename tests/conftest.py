import gzip
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def search_kat():
    with gzip.open(os.path.join(GOLDEN, "search_kat.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def unit_kat():
    with open(os.path.join(GOLDEN, "unit_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def net_kat():
    return dict(np.load(os.path.join(GOLDEN, "net_kat.npz")))


@pytest.fixture(scope="session")
def rng_kat():
    return dict(np.load(os.path.join(GOLDEN, "rng_kat.npz")))


@pytest.fixture(scope="session")
def rules_kat():
    return dict(np.load(os.path.join(GOLDEN, "ttt_rules.npz")))


def full_table(net_kat, name):
    """[19683,10] float32 table (9 probs + value) from the reference outputs of
    golden net `name`; rows of unreachable/terminal positions stay zero."""
    t = np.zeros((3 ** 9, 10), np.float32)
    codes = net_kat["codes"]
    t[codes, :9] = net_kat[f"{name}_i2_probs"]
    t[codes, 9] = net_kat[f"{name}_i2_value"]
    return t


@pytest.fixture(scope="session")
def net_kat2():
    return dict(np.load(os.path.join(GOLDEN, "net_kat2.npz")))


@pytest.fixture(scope="session")
def net_kat3():
    return dict(np.load(os.path.join(GOLDEN, "net_kat3.npz")))


NETS2 = {"D": ("resnet", 3, 32, 3, 3, 2.0), "E": ("convnet", 4, 32, 3, 3, 2.0), "F": ("convnet", 5, 48, 2, 1, 2.0)}


def nets2_weights(name):
    from nuzero_amd.weights import synthetic_weights, resnet_param_shapes, convnet_param_shapes
    arch, seed, width, depth, k, gain = NETS2[name]
    shapes = resnet_param_shapes(2, 1, width, depth) if arch == "resnet" else convnet_param_shapes(2, 1, k, width, depth)
    return synthetic_weights(seed, shapes, gain)


# board-sized nets (SCS shapes), as tests/golden/make_golden.py NETS3:
# name -> (arch, seed, in, planes, rows, cols, width, depth, recall, value_activation, iters, positions, gain)
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


def nets3_weights(name):
    from nuzero_amd.weights import (synthetic_weights, resnet_param_shapes, convnet_param_shapes,
                                    recurrent_net_param_shapes)
    arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain = NETS3[name]
    if arch == "recurrent":
        shapes = recurrent_net_param_shapes(cin, planes, width, depth, recall)
    elif arch == "resnet":
        shapes = resnet_param_shapes(cin, planes, width, depth)
    else:
        shapes = convnet_param_shapes(cin, planes, 3, width, depth)
    return synthetic_weights(seed, shapes, gain)


def nets3_inputs(name, n=None, offset=0):
    """The generator's inputs (offset 0, n = positions of the case) or more of the same kind."""
    import numpy as np
    _, seed, cin, _, rows, cols, *_rest = NETS3[name]
    n = NETS3[name][11] if n is None else n
    rs = np.random.RandomState(1000 + seed + offset)
    x = (rs.random_sample((n, cin, rows, cols)) < 0.15).astype(np.float32)
    x[:, -3:] = rs.random_sample((n, 3, rows, cols)).astype(np.float32)
    return x


def nets3_oracle(name):
    from oracle.net import RecurrentNetRef, FeedForwardRef
    arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain = NETS3[name]
    w = nets3_weights(name)
    if arch == "recurrent":
        return RecurrentNetRef(w, cin, planes, width, depth, recall, vact)
    return FeedForwardRef(w, arch, depth, vact)


def named_weights_module(weights, recurrent=True):
    """A torch module whose state_dict() has the reference's parameter names (what Network_Manager reads)."""
    import collections
    import torch

    class NamedWeights(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.recurrent = recurrent
            self._names = list(weights)
            self.params = torch.nn.ParameterList(
                [torch.nn.Parameter(torch.from_numpy(np.array(v, np.float32))) for v in weights.values()])

        def state_dict(self, *args, keep_vars=False, **kwargs):
            return collections.OrderedDict((n, p if keep_vars else p.detach()) for n, p in zip(self._names, self.params))

    return NamedWeights()
