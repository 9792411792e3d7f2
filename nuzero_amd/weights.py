"""Synthetic weights for the square-conv RecurrentNet of the reference
(Neural_Networks/Architectures/RecurrentNet.py:18-99, blocks.py).

There is no network access for checkpoints, so benchmarks and tests use
random-init weights of the reference architecture.  They are drawn with numpy
(not torch) so that the same bytes can be regenerated anywhere: uniform in
(-1/sqrt(fan_in), 1/sqrt(fan_in)), the range PyTorch's default Conv2d
initialisation uses.

Parameter names are the reference's ``state_dict`` keys, so a dict from here and
a reference checkpoint are interchangeable.
"""
import math

import numpy as np


def head_channels(width, out_channels, n_layers):
    """Reduce_PolicyHead / Reduce_ValueHead channel schedule: a float step,
    truncated with int() per layer (blocks.py:56-66,144-153)."""
    step = (out_channels - width) / n_layers
    chans, prev = [int(width)], float(width)
    for _ in range(n_layers):
        prev += step
        chans.append(int(prev))
    return chans


def recurrent_net_param_shapes(in_channels, policy_channels, width=64, num_blocks=2, recall=True):
    shapes = [("projection.0.weight", (width, in_channels, 3, 3))]
    first_block = 0
    if recall:
        shapes.append(("recur_module.0.weight", (width, width + in_channels, 3, 3)))
        first_block = 1
    for b in range(num_blocks):
        pre = f"recur_module.{first_block + b}.before_shortcut."
        shapes.append((pre + "0.weight", (width, width, 3, 3)))
        shapes.append((pre + "2.weight", (width, width, 3, 3)))
    pc = head_channels(width, policy_channels, 2)
    for i in range(2):
        shapes.append((f"policy_head.layers.{2 * i}.weight", (pc[i + 1], pc[i], 3, 3)))
    vc = head_channels(width, 1, 4)
    for i in range(4):
        shapes.append((f"value_head.layers.{2 * i}.weight", (vc[i + 1], vc[i], 3, 3)))
    return shapes


def resnet_param_shapes(in_channels, policy_channels, width=64, num_blocks=4):
    """ResNet(..., hex=False, batch_norm=False) (Neural_Networks/Architectures/ResNet.py:13-70)."""
    shapes = [("input_block.0.weight", (width, in_channels, 3, 3))]
    for b in range(num_blocks):
        shapes.append((f"residual_blocks.{b}.before_shortcut.0.weight", (width, width, 3, 3)))
        shapes.append((f"residual_blocks.{b}.before_shortcut.2.weight", (width, width, 3, 3)))
    return shapes + _head_shapes(width, policy_channels)


def convnet_param_shapes(in_channels, policy_channels, kernel_size=3, width=64, num_layers=6):
    """ConvNet(..., hex=False) (Neural_Networks/Architectures/ConvNet.py:12-57)."""
    k = kernel_size
    shapes = [("general_module.0.weight", (width, in_channels, k, k))]
    for i in range(num_layers):
        shapes.append((f"general_module.{2 * (i + 1)}.weight", (width, width, k, k)))
    return shapes + _head_shapes(width, policy_channels)


def _head_shapes(width, policy_channels):
    shapes = []
    pc = head_channels(width, policy_channels, 2)
    for i in range(2):
        shapes.append((f"policy_head.layers.{2 * i}.weight", (pc[i + 1], pc[i], 3, 3)))
    vc = head_channels(width, 1, 4)
    for i in range(4):
        shapes.append((f"value_head.layers.{2 * i}.weight", (vc[i + 1], vc[i], 3, 3)))
    return shapes


def synthetic_weights(seed, shapes, gain=1.0):
    """name -> float32 array for any (name, shape) list, same rule as below."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes:
        bound = gain / math.sqrt(shape[1] * shape[2] * shape[3])
        out[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def synthetic_recurrent_net_weights(seed, in_channels, policy_channels, width=64,
                                    num_blocks=2, recall=True, gain=1.0):
    """name -> float32 array, drawn from ``np.random.RandomState(seed)`` in
    parameter order.  ``gain`` scales every tensor (tests use gain > 1 to get
    sharper, less uniform policies)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in recurrent_net_param_shapes(in_channels, policy_channels, width,
                                                  num_blocks, recall):
        fan_in = shape[1] * shape[2] * shape[3]
        bound = gain / math.sqrt(fan_in)
        out[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def hex_param_shapes(shapes):
    """The same network with hex=True: every conv is hexagdly.Conv2d(kernel_size=1), whose parameters are
    kernel0 [out, in, 3, 1] (the cell's own column) and kernel1 [out, in, 2, 2] (the two adjacent columns)."""
    out = []
    for name, (o, i, _kh, _kw) in shapes:
        base = name[:-len("weight")]
        out.append((base + "kernel0", (o, i, 3, 1)))
        out.append((base + "kernel1", (o, i, 2, 2)))
    return out
