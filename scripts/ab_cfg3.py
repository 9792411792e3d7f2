#!/usr/bin/env python3
"""bench.py's ttt_config3_share alone (configs[2]'s per-GPU share: 1024 games, 400 simulations) -- for same-box A/B of
NZ_SLOTS_PER_WG / NZ_SIMS_PER_CYCLE."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nuzero_amd.search_config import legacy_ttt_search_config  # noqa: E402
from nuzero_amd.weights import synthetic_recurrent_net_weights  # noqa: E402
r = bench.ttt_config3_share(legacy_ttt_search_config(100), synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True), 2, 0)
print(json.dumps({k: r[k] for k in ("value", "expansions_per_s")}))
