#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--sims S]

A "step" is one self-play round: G concurrent Tic-Tac-Toe games per GPU played
to the end with S MCTS simulations per move, network fused into the search
(BASELINE.json configs[1]: 100 sims/move, 4096 concurrent games, 1 x MI355X),
followed -- when N > 1 -- by the RCCL gather of the finished games to rank 0's
replay buffer.  Weights are synthetic (random-init RecurrentNet(2,1,64,2), seed
0); games need no dataset.  Prints ONE JSON line on rank 0.

Extra keys: `expansions_per_s`, `simulations_per_s` (the metric's second half),
`roofline` for the dominant kernel (the fused network kernel, FP32 MFMA),
`roofline_select` for the tree kernel (HBM), `cpu_baseline` = the CPU oracle
(oracle/search.py + oracle/net.py, the restatement of the reference's
Explorer/Gamer path) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak (dense)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak


def cpu_baseline(sims, seconds=15.0):
    """The CPU oracle on a bounded sample of the same workload: whole games at
    `sims` simulations/move, one process, one torch thread."""
    import torch
    from oracle import ttt as ottt, search as osearch
    from oracle.net import RecurrentNetRef
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    torch.set_num_threads(1)
    net = RecurrentNetRef(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True), 2, 1, 64, 2)
    ev = osearch.net_evaluator(net, 2)
    cfg = legacy_ttt_search_config(sims)
    t0 = time.perf_counter()
    games = expansions = simulations = 0
    while time.perf_counter() - t0 < seconds:
        g = ottt.TicTacToe()
        _, cnt = osearch.play_game(g, ev, cfg, np.random.RandomState(games))
        games += 1
        expansions += cnt.expansions
        simulations += cnt.simulations
    dt = time.perf_counter() - t0
    return {"value": games / dt, "unit": "games/s", "cores": 1, "kind": "port",
            "expansions_per_s": expansions / dt, "simulations_per_s": simulations / dt,
            "sample": f"{games} whole games, {sims} sims/move, seeds 0..{games - 1}, "
                      f"{dt:.1f} s on 1 host core (python oracle + torch fp32 net, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--iters", type=int, default=2, help="recurrent iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd import dist as nzdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = legacy_ttt_search_config(args.sims)
    eng = SelfPlayEngine(cfg, args.games, training=True, device=local_rank)
    eng.set_weights(synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True), recurrent_iterations=args.iters)
    gather = nzdist.ReplayGather(eng, world, rank) if world > 1 else None

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def step(i):
        # game g of rank r in round i gets its own stream seed
        eng.play(base_seed=(i * world + rank) * args.games)
        if gather is not None:
            gather.gather()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    sims_total = exp_total = games_total = 0
    for i in range(args.steps):
        step(args.warmup + i)
        c = eng.counters()
        sims_total += c["simulations"]
        exp_total += c["expansions"]
        games_total += args.games
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, sims_total, exp_total, games_total], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        dt = float(tmax[0])
        sims_total, exp_total, games_total = float(t[1]), float(t[2]), float(t[3])

    # ---- per-kernel timing, HIP events on the engine's stream (one more round, not in `value`)
    eng.profile(True)
    eng.play(base_seed=10 ** 6 + rank * args.games)
    prof = eng.profile_read()
    eng.profile(False)
    pc = eng.counters()
    flops_pos = eng.net_flops_per_position()
    net_ms, net_n = prof["network"]["ms"], prof["network"]["launches"]
    tree_ms, tree_n = prof["tree_advance"]["ms"], prof["tree_advance"]["launches"]
    achieved_tf = pc["expansions"] * flops_pos / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
    # select/backup algorithmic bytes: 11 + 20 k bytes per internal node whose k
    # children are scored, 24 bytes per path node backed up (path = levels + 1 per simulation)
    sel_bytes = (11 * pc["select_nodes"] + 20 * pc["select_children"]
                 + 24 * (pc["select_nodes"] + pc["simulations"]))
    sel_gbs = sel_bytes / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "self-play games/sec (+ MCTS node-expansions/sec), Tic-Tac-Toe %d sims/move" % args.sims,
            "value": games_total / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Tic_Tac_Toe, %d sims/move, %d concurrent self-play games per GPU, "
                                   "RecurrentNet(2,1,64,2) %d recurrent iterations, legacy TTT search config"
                                   % (args.sims, args.games, args.iters),
                       "games_per_gpu": args.games, "sims_per_move": args.sims,
                       "parallelism": "games sharded by rank, 1 RCCL gather/round" if world > 1 else "1 GPU"},
            "expansions_per_s": exp_total / dt, "simulations_per_s": sims_total / dt,
            "roofline": {"bound": "mfma", "kernel": "net_kernel (fused RecurrentNet forward, FP32 MFMA)",
                         "achieved": achieved_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                         "flops_per_position": flops_pos, "positions_per_launch": pc["expansions"] / max(net_n, 1),
                         "avg_launch_us": net_ms * 1e3 / max(net_n, 1), "launches": net_n},
            "roofline_select": {"bound": "hbm", "kernel": "advance_kernel (select + expand + backup)",
                                "achieved": sel_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": sel_gbs / HBM_PEAK_GBS, "traffic": None,
                                "avg_launch_us": tree_ms * 1e3 / max(tree_n, 1), "launches": tree_n},
            "kernel_ms_per_round": {k: v["ms"] for k, v in prof.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
