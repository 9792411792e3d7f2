"""Experiment: rounds of two engines in flight at once (two host threads, two HIP streams), so that the workgroups of one
round's tail share the GPU with the next round.  Prints games/s for 1 and 2 engines over the same number of rounds."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuzero_amd.engine import SelfPlayEngine
from nuzero_amd.weights import synthetic_recurrent_net_weights
from nuzero_amd.search_config import legacy_ttt_search_config

ROUND, SLOTS, ROUNDS = 16384, 4096, 8
w = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True)
for n_eng in (1, 2):
    engs = []
    for i in range(n_eng):
        e = SelfPlayEngine(legacy_ttt_search_config(100), ROUND, training=True, device=0, n_slots=SLOTS)
        e.set_weights(w, recurrent_iterations=2)
        engs.append((e, torch.cuda.Stream()))

    def run(i, rounds):
        e, st = engs[i]
        with torch.cuda.stream(st):
            for r in rounds:
                e.play(base_seed=r * ROUND, next_base_seed=(r + n_eng) * ROUND)

    for i in range(n_eng):
        run(i, [i])                                     # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, list(range(n_eng + i, n_eng + ROUNDS, n_eng)))) for i in range(n_eng)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(n_eng, "engine(s):", round(ROUNDS * ROUND / dt), "games/s", flush=True)
    for e, _ in engs: e.close()
