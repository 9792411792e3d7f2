#!/usr/bin/env python3
"""bench.py's scs_config5 alone (for rocprofv3 --kernel-trace --stats): prints its JSON."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
print(json.dumps(bench.scs_config5(0, games=int(os.environ.get("NZ_CFG5_GAMES", "256")))))
