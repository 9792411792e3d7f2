#!/usr/bin/env python3
"""Golden vectors for the hexagonal nets -- NOT RUN YET: needs the real `hexagdly` package next to the reference
(/root/reference), which this build container does not have (hex parity is unpinned).

Where hexagdly is installed, this script imports the reference's RecurrentNet / ResNet / ConvNet with hex=True,
loads the synthetic weights of tests/conftest.py NETS3 in hexagdly's own parameter layout (kernel0 / kernel1) and
writes tests/golden/net_kat_hex.npz; tests/test_oracle_golden.py::test_hex_nets_against_reference (skipped while
the file is missing) then pins oracle/net.py HexNetRef -- and through tests/test_gpu_boardnet.py the kernels --
to the reference.

    python tests/golden/make_golden_hex.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

_tc = types.ModuleType("termcolor")
_tc.colored = lambda s, *a, **k: s
sys.modules.setdefault("termcolor", _tc)

import hexagdly  # noqa: E402,F401  (the real package: no stand-in here)
import torch  # noqa: E402
from scipy.special import softmax  # noqa: E402

from Neural_Networks.Network_Manager import Network_Manager  # noqa: E402
from Neural_Networks.Architectures.RecurrentNet import RecurrentNet  # noqa: E402
from Neural_Networks.Architectures.ResNet import ResNet  # noqa: E402
from Neural_Networks.Architectures.ConvNet import ConvNet  # noqa: E402

from conftest import NETS3, nets3_inputs  # noqa: E402
from nuzero_amd.weights import (synthetic_weights, hex_param_shapes, recurrent_net_param_shapes,  # noqa: E402
                                resnet_param_shapes, convnet_param_shapes)

torch.set_num_threads(1)


def main():
    out = {}
    for name, (arch, seed, cin, planes, rows, cols, width, depth, recall, vact, iters, n, gain) in NETS3.items():
        if arch == "recurrent":
            shapes = recurrent_net_param_shapes(cin, planes, width, depth, recall)
            net = RecurrentNet(cin, planes, width, depth, recall=recall, value_activation=vact, hex=True)
        elif arch == "resnet":
            shapes = resnet_param_shapes(cin, planes, width, depth)
            net = ResNet(cin, planes, num_filters=width, num_blocks=depth, value_activation=vact, hex=True)
        else:
            shapes = convnet_param_shapes(cin, planes, 3, width, depth)
            net = ConvNet(cin, planes, kernel_size=1, num_filters=width, num_layers=depth, hex=True)
        w = synthetic_weights(100 + seed, hex_param_shapes(shapes), gain)
        sd = net.state_dict()
        assert list(sd.keys()) == list(w.keys()), (list(sd.keys()), list(w.keys()))
        assert all(tuple(sd[k].shape) == w[k].shape for k in w), "hexagdly's parameter shapes differ from the restatement"
        net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        nm = Network_Manager(net)
        x = nets3_inputs(name)
        logits = np.zeros((n, planes * rows * cols), np.float32)
        vals = np.zeros((n,), np.float32)
        for i in range(n):
            state = torch.from_numpy(x[i:i + 1])
            p, v = nm.inference(state, False, iters) if arch == "recurrent" else nm.inference(state, False)
            logits[i], vals[i] = p.numpy().reshape(-1), v.item()
        out[f"{name}_logits"], out[f"{name}_probs"], out[f"{name}_value"] = logits, softmax(logits, axis=1), vals
    np.savez_compressed(os.path.join(HERE, "net_kat_hex.npz"), **out)
    print("wrote net_kat_hex.npz")


if __name__ == "__main__":
    main()
