"""Run the stand-alone network kernel a few times (profiling target)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuzero_amd.engine import SelfPlayEngine
from nuzero_amd.weights import synthetic_recurrent_net_weights
from nuzero_amd.search_config import legacy_ttt_search_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = SelfPlayEngine(legacy_ttt_search_config(100), 16)
eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
x = (torch.rand((B, 2, 3, 3), device="cuda") > 0.6).float()
for _ in range(5):
    eng.net_forward(x, want_probs=False)
torch.cuda.synchronize()
