"""The one-launch board network alone (configs[3] shape: 5x5, ConvNet 32 x 8) at several batch sizes: time per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nuzero_amd.boardnet import BoardNet
from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
net = BoardNet("convnet", 86, 21, 5, 5, width=32, num_blocks=8, max_batch=1024)
net.set_weights(synthetic_weights(0, convnet_param_shapes(86, 21, 3, 32, 8)))
x = (torch.rand((1024, 86, 5, 5), device="cuda") < 0.15).float()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for fused in ((True,) if os.environ.get('FUSED_ONLY') else (True, False)):
    net.fused(fused)
    for n in (64, 256, 512, 640, 768, 1024):
        n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
        net.forward(x, n_dev=n_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net.forward(x, n_dev=n_dev)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("fused" if fused else "layers", n, "positions: %.1f us per forward (incl. input conversion), %.1f TFLOP/s" % (dt * 1e6, n * net.flops_per_position / dt / 1e12))
