#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--round R] [--sims S]

A "step" is one self-play round on one GPU: R Tic-Tac-Toe games played to the
end on G concurrent game trees (slots that finish a game take the next one, as
the reference's ActorPool does), S MCTS simulations per move, network fused
into the search (BASELINE.json configs[1]: 100 sims/move, 4096 concurrent games,
1 x MI355X; default R = 4 G), followed -- when N > 1 -- by the RCCL gather of the
finished games to rank 0's replay buffer.  Weights are synthetic (random-init
RecurrentNet(2,1,64,2), seed 0); games need no dataset.  Prints ONE JSON line on
rank 0.

Extra keys
  expansions_per_s, simulations_per_s   the metric's second half
  roofline          the dominant kernel: the persistent self-play kernel, priced
                    against the FP32 MFMA peak with the network's algorithmic
                    FLOPs (in-bounds taps only) over the kernel's HIP-event time
  roofline_net      the fused network kernel alone (4096 positions per launch)
  roofline_select   the lock-step tree kernel (select + expand + backup) against
                    HBM, SURVEY.md 8(d) byte counts
  phases            in-kernel shader-clock shares of the stamped diagnostic build
  cpu_baseline      the CPU oracle (oracle/search.py + oracle/net.py, the
                    restatement of the reference's Explorer/Gamer path) timed on
                    this box's host cores on a bounded sample
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
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16 matrix peak (dense)
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games (slots) per GPU")
    ap.add_argument("--round", type=int, default=0, help="games per self-play round per GPU (default 4 x --games)")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--iters", type=int, default=2, help="recurrent iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel measurements after the timed steps")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    n_round = args.round if args.round > 0 else 4 * args.games

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
    weights = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True)
    eng = SelfPlayEngine(cfg, n_round, training=True, device=local_rank, n_slots=args.games)
    eng.set_weights(weights, recurrent_iterations=args.iters)
    gather = nzdist.ReplayGather(eng, world, rank) if world > 1 else None

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def step(i):
        # every game of every rank and round has its own stream seed
        eng.play(base_seed=(i * world + rank) * n_round)
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
        games_total += n_round
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, sims_total, exp_total, games_total], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        dt = float(tmax[0])
        sims_total, exp_total, games_total = float(t[1]), float(t[2]), float(t[3])

    out = {
        "metric": "self-play games/sec (+ MCTS node-expansions/sec), Tic-Tac-Toe %d sims/move" % args.sims,
        "value": games_total / dt, "unit": "games/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "arithmetic": "network: float32 values, every product formed from six bf16 MFMA terms on exact three-way "
                      "splits, float32 accumulation (as accurate as a float32 convolution, DESIGN.md section 4); "
                      "tree statistics float64",
        "config": {"workload": "Tic_Tac_Toe, %d sims/move, %d concurrent self-play games per GPU (%d games per "
                               "round), RecurrentNet(2,1,64,2) %d recurrent iterations f32, tree statistics f64, "
                               "legacy TTT search config" % (args.sims, args.games, n_round, args.iters),
                   "concurrent_games_per_gpu": args.games, "games_per_round_per_gpu": n_round,
                   "sims_per_move": args.sims,
                   "parallelism": "games sharded by rank, 1 RCCL gather per round" if world > 1 else "1 GPU"},
        "expansions_per_s": exp_total / dt, "simulations_per_s": sims_total / dt,
    }

    if not args.no_extras:
        flops_pos = eng.net_flops_per_position()
        bf16_pos, f32_pos = eng.net_matrix_flops_per_position()

        def executed(tflops_algorithmic):
            """The same time priced by the MFMA instructions actually issued."""
            scale = tflops_algorithmic / flops_pos
            return {"bf16_mfma_tflops": scale * bf16_pos, "bf16_mfma_peak": MFMA_BF16_PEAK_TFLOPS,
                    "bf16_mfma_frac": scale * bf16_pos / MFMA_BF16_PEAK_TFLOPS,
                    "f32_mfma_tflops": scale * f32_pos,
                    "matrix_flops_per_position": {"bf16": bf16_pos, "f32": f32_pos}}
        # ---- dominant kernel: HIP events around the persistent kernel, on its own stream, one more round
        eng.profile(True)
        eng.play(base_seed=10 ** 6 + rank * n_round)
        prof = eng.profile_read()
        eng.profile(False)
        pc = eng.counters()
        k_ms, k_n = prof["search"]["ms"], prof["search"]["launches"]
        achieved = pc["expansions"] * flops_pos / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch of this kernel from PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs,
        # profiles/r01_pmc_traffic.json); only quoted for the configuration they were collected on
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and (args.games, n_round, args.sims, args.iters) == (4096, 16384, 100, 2):
            with open(tpath) as f:
                traffic = json.load(f)["hbm_bytes_per_launch_raw"]
        out["roofline"] = {
            "bound": "mfma", "kernel": "selfplay_kernel (persistent: tree phases + fused RecurrentNet forward)",
            "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
            "peak_note": "float32 dense MFMA peak: what a kernel doing this float32 arithmetic on the FP32 matrix "
                         "cores could reach; the kernel issues bf16 MFMAs instead, see `executed`",
            "flops_per_position": flops_pos, "positions_per_launch": pc["expansions"] / max(k_n, 1),
            "avg_launch_us": k_ms * 1e3 / max(k_n, 1), "launches": k_n, "executed": executed(achieved)}
    if not args.no_extras and world == 1:
        # ---- in-kernel phase shares (stamped diagnostic build; its run time is not quoted)
        eng.phase_stamps(True)
        eng.play(base_seed=2 * 10 ** 6 + rank * n_round)
        out["phases"] = eng.phase_stamps(False, read=True)
        # ---- the network kernel alone
        x = (torch.rand((4096, 2, 3, 3), device="cuda") > 0.6).float()
        eng.net_forward(x, want_probs=False)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(20):
            eng.net_forward(x, want_probs=False)
        pn = eng.profile_read()["network"]
        eng.profile(False)
        net_tf = 20 * 4096 * flops_pos / (pn["ms"] * 1e-3) / 1e12
        out["roofline_net"] = {"bound": "mfma", "kernel": "net_kernel (fused RecurrentNet forward, 4096 positions)",
                               "achieved": net_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": net_tf / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_us": pn["ms"] * 1e3 / 20, "executed": executed(net_tf)}
        # ---- the tree kernel alone against HBM (SURVEY.md 8d bytes: 11 + 20 k per scored node, 24 per
        #      path node backed up): lock-step route with a table evaluator, so advance_kernel does a whole
        #      move's select/expand/backup per launch; at the workload's 4096 trees and at 65536 trees
        rs = np.random.RandomState(0)
        table = np.zeros((3 ** 9, 10), np.float32)
        table[:, :9] = rs.dirichlet(np.ones(9), 3 ** 9)
        table[:, 9] = rs.uniform(-0.5, 0.5, 3 ** 9)
        sel = {}
        for n_trees in (args.games, 65536):
            ls = SelfPlayEngine(cfg, n_trees, training=True, device=local_rank)
            ls.set_table(table)
            ls.play_lockstep(base_seed=0)
            ls.profile(True)
            ls.play_lockstep(base_seed=n_trees)
            pl = ls.profile_read()["search"]
            ls.profile(False)
            lc = ls.counters()
            sel_bytes = (11 * lc["select_nodes"] + 20 * lc["select_children"]
                         + 24 * (lc["select_nodes"] + lc["simulations"]))
            sel[n_trees] = {"achieved": sel_bytes / (pl["ms"] * 1e-3) / 1e9,
                            "bytes_per_launch": sel_bytes / max(pl["launches"], 1),
                            "avg_launch_us": pl["ms"] * 1e3 / max(pl["launches"], 1), "launches": pl["launches"],
                            "simulations_per_s": lc["simulations"] / (pl["ms"] * 1e-3)}
            ls.close()
        big = sel[65536]
        out["roofline_select"] = {"bound": "hbm", "kernel": "advance_kernel (select + expand + backup, table evaluator, "
                                  "65536 concurrent trees)", "achieved": big["achieved"], "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": big["achieved"] / HBM_PEAK_GBS, "traffic": None,
                                  "bytes_per_launch": big["bytes_per_launch"], "avg_launch_us": big["avg_launch_us"],
                                  "launches": big["launches"], "simulations_per_s": big["simulations_per_s"],
                                  "at_workload_trees": dict(sel[args.games], trees=args.games)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
