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


NETS2 = {"D": ("resnet", 3, 32, 3, 3, 2.0), "E": ("convnet", 4, 32, 3, 3, 2.0), "F": ("convnet", 5, 48, 2, 1, 2.0)}


def nets2_weights(name):
    from nuzero_amd.weights import synthetic_weights, resnet_param_shapes, convnet_param_shapes
    arch, seed, width, depth, k, gain = NETS2[name]
    shapes = resnet_param_shapes(2, 1, width, depth) if arch == "resnet" else convnet_param_shapes(2, 1, k, width, depth)
    return synthetic_weights(seed, shapes, gain)
