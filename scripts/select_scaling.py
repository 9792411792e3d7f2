"""Tree kernel alone (table evaluator, no network) at growing numbers of concurrent trees:
algorithmic select/backup bytes per second of advance_kernel (SURVEY.md 8d byte counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nuzero_amd.engine import SelfPlayEngine
from nuzero_amd.search_config import legacy_ttt_search_config
rs = np.random.RandomState(0)
table = np.zeros((3 ** 9, 10), np.float32)
p = rs.dirichlet(np.ones(9), 3 ** 9).astype(np.float32)
table[:, :9] = p
table[:, 9] = rs.uniform(-0.5, 0.5, 3 ** 9).astype(np.float32)
cfg = legacy_ttt_search_config(100)
for G in (4096, 16384, 65536, 262144):
    eng = SelfPlayEngine(cfg, G)
    eng.set_table(table)
    eng.play_lockstep(0)
    eng.profile(True)
    eng.play_lockstep(G)
    pr = eng.profile_read()["search"]
    eng.profile(False)
    c = eng.counters()
    b = 11 * c["select_nodes"] + 20 * c["select_children"] + 24 * (c["select_nodes"] + c["simulations"])
    print("G %7d: advance %8.2f ms over %d launches, %6.1f M sims/s, %7.1f GB/s algorithmic (%.2f %% of 8 TB/s)" % (
        G, pr["ms"], pr["launches"], c["simulations"] / pr["ms"] / 1e3, b / pr["ms"] / 1e6, b / pr["ms"] / 1e6 / 80))
    eng.close()
