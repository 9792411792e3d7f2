"""Diagnostic: does a second RoundPipeline in the same process overlap as well as the first?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuzero_amd.engine import SelfPlayEngine, RoundPipeline
from nuzero_amd.weights import synthetic_recurrent_net_weights
from nuzero_amd.search_config import legacy_ttt_search_config
ROUND, SLOTS, ROUNDS = 16384, 4096, 6
w = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True)
def make():
    e = SelfPlayEngine(legacy_ttt_search_config(100), ROUND, training=True, device=0, n_slots=SLOTS)
    e.set_weights(w, recurrent_iterations=2)
    return e
PRIO = int(os.environ.get('NORMAL_PRIORITY', '0'))
for trial in range(4):
    pipe = RoundPipeline(make, depth=2)
    if PRIO:   # 0 = normal priority, what RoundPipeline used before
        pipe.streams = [torch.cuda.Stream() for _ in pipe.engines]
    for i in range(2):
        pipe.submit(i * ROUND, next_base_seed=(i + 2) * ROUND)
    while pipe.pending: pipe.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 2 + ROUNDS):
        if len(pipe.pending) == 2: pipe.collect()
        pipe.submit(i * ROUND, next_base_seed=(i + 2) * ROUND)
    while pipe.pending: pipe.collect()
    torch.cuda.synchronize()
    print("trial", trial, round(ROUNDS * ROUND / (time.perf_counter() - t0)), "games/s; streams", [s.cuda_stream for s in pipe.streams], flush=True)
    pipe.close()
