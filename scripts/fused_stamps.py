import os, sys
sys.path.insert(0, ".")
import torch
from nuzero_amd.boardnet import BoardNet
from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
net = BoardNet("convnet", 86, 21, 5, 5, width=32, num_blocks=8, max_batch=1024)
net.set_weights(synthetic_weights(0, convnet_param_shapes(86, 21, 3, 32, 8)))
x = (torch.rand((1024, 86, 5, 5), device="cuda") < 0.15).float()
for n in (1024,):
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    net.forward(x, n_dev=n_dev); torch.cuda.synchronize()
    print("---- second launch", n, flush=True)
    net.forward(x, n_dev=n_dev); torch.cuda.synchronize()
