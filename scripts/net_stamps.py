"""Print the network kernel's per-section ticks (diagnostic build) and its launch time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuzero_amd.engine import SelfPlayEngine
from nuzero_amd.weights import synthetic_recurrent_net_weights
from nuzero_amd.search_config import legacy_ttt_search_config
eng = SelfPlayEngine(legacy_ttt_search_config(100), 16)
eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
x = (torch.rand((4096, 2, 3, 3), device="cuda") > 0.6).float()
eng.net_forward(x, want_probs=False); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): eng.net_forward(x, want_probs=False)
b.record(); torch.cuda.synchronize()
print(os.environ.get("NZ_LIB_PATH", "default"), "us/launch %.1f" % (a.elapsed_time(b) * 20), eng.net_forward_stamps(x))
