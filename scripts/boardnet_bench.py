#!/usr/bin/env python3
"""Time the board-sized network kernels alone (nz_boardnet_forward) and the same net in PyTorch/MIOpen.

    python scripts/boardnet_bench.py [--arch resnet] [--width 64] [--blocks 4] [--rows 5] [--cols 5] [--batch 1024]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet")
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--rows", type=int, default=5)
    ap.add_argument("--cols", type=int, default=5)
    ap.add_argument("--channels", type=int, default=86)
    ap.add_argument("--planes", type=int, default=21)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--torch", action="store_true", help="also time the oracle net's forward on the GPU (MIOpen)")
    a = ap.parse_args()
    import torch
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.weights import (synthetic_weights, resnet_param_shapes, convnet_param_shapes,
                                    recurrent_net_param_shapes)
    if a.arch == "recurrent":
        shapes = recurrent_net_param_shapes(a.channels, a.planes, a.width, a.blocks, True)
    elif a.arch == "resnet":
        shapes = resnet_param_shapes(a.channels, a.planes, a.width, a.blocks)
    else:
        shapes = convnet_param_shapes(a.channels, a.planes, 3, a.width, a.blocks)
    w = synthetic_weights(1, shapes)
    net = BoardNet(a.arch, a.channels, a.planes, a.rows, a.cols, width=a.width, num_blocks=a.blocks, max_batch=a.batch)
    net.set_weights(w, a.iters)
    x = (torch.rand((a.batch, a.channels, a.rows, a.cols), device="cuda") < 0.15).float()
    for _ in range(3):
        net.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        net.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    fl = net.flops_per_position * a.batch
    print(f"boardnet {a.arch} w{a.width} b{a.blocks} {a.rows}x{a.cols} batch {a.batch}: {dt * 1e6:.1f} us/forward, "
          f"{fl / dt / 1e12:.2f} TFLOP/s algorithmic ({net.flops_per_position / 1e6:.2f} MFLOP/position)")
    if a.torch:
        import torch.nn.functional as F
        torch.backends.cudnn.benchmark = True
        wd = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
        conv = lambda t, name: F.conv2d(t, wd[name], None, 1, "same")

        def fwd(t):
            if a.arch == "resnet":
                t = F.relu(conv(t, "input_block.0.weight"))
                for b in range(a.blocks):
                    pre = f"residual_blocks.{b}.before_shortcut."
                    t = F.relu(conv(F.relu(conv(t, pre + "0.weight")), pre + "2.weight") + t)
            elif a.arch == "convnet":
                t = F.elu(conv(t, "general_module.0.weight"))
                for i in range(a.blocks):
                    t = F.elu(conv(t, f"general_module.{2 * (i + 1)}.weight"))
            else:
                x0 = t
                t = F.relu(conv(t, "projection.0.weight"))
                for _ in range(a.iters):
                    t = conv(torch.cat([t, x0], 1), "recur_module.0.weight")
                    for b in range(a.blocks):
                        pre = f"recur_module.{1 + b}.before_shortcut."
                        t = F.relu(conv(F.relu(conv(t, pre + "0.weight")), pre + "2.weight") + t)
            p = conv(F.relu(conv(t, "policy_head.layers.0.weight")), "policy_head.layers.2.weight")
            v = t
            for i in range(4):
                v = conv(v, f"value_head.layers.{2 * i}.weight")
                if i != 3:
                    v = torch.tanh(v)
            return torch.softmax(p.reshape(p.shape[0], -1), 1), torch.tanh(v.mean(dim=(1, 2, 3)))
        with torch.no_grad():
            for _ in range(3):
                fwd(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fwd(x)
            torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0) / a.reps
        print(f"torch/MIOpen same net: {dtt * 1e6:.1f} us/forward ({fl / dtt / 1e12:.2f} TFLOP/s)")


if __name__ == "__main__":
    main()
