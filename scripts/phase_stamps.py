import sys, time; import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuzero_amd.engine import SelfPlayEngine
from nuzero_amd.weights import synthetic_recurrent_net_weights
from nuzero_amd.search_config import legacy_ttt_search_config
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else G
eng = SelfPlayEngine(legacy_ttt_search_config(100), G, n_slots=S)
eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True))
eng.play(0)
for rep in range(2):
    t0 = time.perf_counter(); eng.play(rep * G); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = eng.counters()
    print("G", G, "slots", S, "round ms", dt * 1e3, "games/s", G / dt, "exp/s", c["expansions"] / dt, "exp/game", c["expansions"] / G)
eng.phase_stamps(True)
t0 = time.perf_counter(); eng.play(5 * G); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = eng.phase_stamps(False, read=True)
print(st)
print("stamped round ms", dt * 1e3, "=> shader clock GHz ~", st["max_workgroup_ticks"] / dt / 1e9)
# stand-alone network kernel timing
import numpy as np
for B in (16, 4096, 8192):
    x = torch.zeros((B, 2, 3, 3), device="cuda"); x[:, 0, 0, 0] = 1
    eng.net_forward(x, want_probs=False); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): eng.net_forward(x, want_probs=False)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    print("net_forward B", B, "us/launch", us, "TFLOP/s", B * eng.net_flops_per_position() / us / 1e6)
r = eng.export(states=False)
import numpy as np
ex = None
print("net stamps (ticks/workgroup)", eng.net_forward_stamps(torch.zeros((4096, 2, 3, 3))))
print("lengths", np.bincount(r["lengths"]))
