#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--round R] [--sims S]
    python bench.py --round 16384 --no-extras      # the round size `value` was measured on until round 2

A "step" is one self-play round on one GPU: R Tic-Tac-Toe games played to the
end on G concurrent game trees (slots that finish a game take the next one, as
the reference's ActorPool does), S MCTS simulations per move, network fused
into the search (BASELINE.json configs[1]: 100 sims/move, 4096 concurrent games,
1 x MI355X; default R = 16 G: one launch of the persistent kernel, whose start and end-of-round
tail are paid once per round), followed -- when N > 1 -- by the RCCL gather of the
finished games to rank 0's replay buffer (and preceded, once, by the broadcast of
the weights from rank 0).  Weights are synthetic (random-init RecurrentNet(2,1,64,2),
seed 0); games need no dataset.  Prints ONE JSON line on rank 0.

Extra keys (N = 1)
  expansions_per_s, simulations_per_s   the metric's second half
  roofline          the dominant kernel, selfplay_kernel, against the dense BF16 MFMA peak: achieved = the bf16 MFMA
                    FLOPs it executes (six split terms per float32 product) over its HIP-event time; the
                    algorithmic float32 figure (in-bounds taps only) is the sub-key `algorithmic_f32`;
                    `traffic` = HBM bytes per launch from rocprofv3 PMC passes, quoted (with the commit they were
                    measured at) only for the configuration they were collected on
  roofline_tree_phase  the tree phases of the same kernel against HBM: algorithmic select / backup / expand bytes
                     from the run's own counters over the stamped tree-phase time
  roofline_net      the fused network kernel alone (4096 positions per launch), BF16 MFMA peak
  roofline_select   the lock-step tree kernel advance_kernel (select + expand + backup, table evaluator) against HBM --
                    NOT on the product route (the persistent kernel does this work in its tree phases)
  phases            in-kernel shader-clock shares of the stamped diagnostic build
  scs_config4       BASELINE.json configs[3]: SCS 5x5, ConvNet(32 filters x 8 layers, square convs), 200 sims/move,
                    1024 concurrent games, one whole round in the library (nz_scs_search_play)
  gamer_surface     Gamer.play_games with the replay buffer on the device: the whole reference-shaped round (search,
                    save_game for every game, statistics), games/s
  round_4x, gamer_surface_round_4x   (only with --rounds-in-flight N) rounds of 4 G games, the size `value` was measured
                    on until round 2: the engine alone, and the Gamer surface with the sub-key play_forever_in_flight
                    (the same worker in the trainer's asynchronous mode, Gamer.play_forever)
  rounds_in_flight_N  (only with --rounds-in-flight N) rounds of 4 G games with N engines in flight (nuzero_amd.engine.RoundPipeline: two HIP streams, the
                    next round's workgroups take the compute units the current round's tail leaves idle), games/s --
                    the reference's asynchronous mode (Gamers that play_forever); NOT `value`, whose rounds run one
                    after the other so that the kernel's duration in `roofline` and in rocprofv3's stats is that of an
                    undisturbed launch
  scs_config4_round4  scs_config4 with four games per concurrent tree in one round (nz_scs_search_play_round: a tree
                    whose game has ended starts the round's next game)
  scs_config5       BASELINE.json configs[4] on one GPU, bounded: SCS 10x10, RecurrentNet(256 x 2, recall) x 16 iterations,
                    400 sims/move, 1024 games x their first decision; expansions/s and the network's rate
  ttt_config3_share BASELINE.json configs[2]'s share of one GPU: 1024 concurrent Tic-Tac-Toe games, 400 sims/move
  cpu_baseline      the CPU oracle (oracle/search.py + oracle/net.py, the restatement of the reference's
                    Explorer/Gamer path) on this box's host cores, one process per core, on a bounded sample
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak (dense)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16 matrix peak (dense)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak


def _cpu_worker(args):
    """One process, one core: whole games at `sims` simulations/move until `seconds` are over."""
    worker, sims, seconds = args
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
        _, cnt = osearch.play_game(g, ev, cfg, np.random.RandomState(worker * 100000 + games))
        games += 1
        expansions += cnt.expansions
        simulations += cnt.simulations
    return games, expansions, simulations, time.perf_counter() - t0


def cpu_baseline(sims, seconds=15.0, max_procs=64):
    """The CPU oracle on a bounded sample of the same workload: one process per host core this job may run on
    (torch: 1 thread each), whole games at `sims` simulations/move for `seconds`."""
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), max_procs)
    with mp.get_context("spawn").Pool(cores) as pool:
        parts = pool.map(_cpu_worker, [(w, sims, seconds) for w in range(cores)])
    games = sum(p[0] for p in parts)
    dt = max(p[3] for p in parts)
    return {"value": games / dt, "unit": "games/s", "cores": cores, "kind": "port",
            "expansions_per_s": sum(p[1] for p in parts) / dt, "simulations_per_s": sum(p[2] for p in parts) / dt,
            "per_core_games_per_s": games / dt / cores,
            "sample": f"{games} whole games, {sims} sims/move, {dt:.1f} s on {cores} host cores, one process per core "
                      f"(python oracle + torch fp32 net, 1 thread each)"}


def launch_ranks(n):
    """One process per GPU through torch.distributed.run (rendezvous on 127.0.0.1, a free port), started from a process
    that has made no GPU call; stdout (rank 0's JSON line) passes through.  Returns the launcher's exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def live_traffic(args, n_round):
    """HBM bytes per launch of the persistent kernel, measured NOW: two child processes under rocprofv3 (--pmc FETCH_SIZE
    and --pmc WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; the program itself right after `--`),
    started before this process makes any GPU call.  None when rocprofv3 is not on PATH or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    vals = {}
    cmd_tail = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                "--games", str(args.games), "--round", str(n_round), "--sims", str(args.sims), "--iters", str(args.iters)]
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="nz_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + cmd_tail,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            got = []
            with open(files[0]) as f:
                rd = csv.reader(f)
                head = next(rd)
                kn, cn, cv = head.index("Kernel_Name"), head.index("Counter_Name"), head.index("Counter_Value")
                for row in rd:
                    if "selfplay_kernel<false>" in row[kn] and row[cn] == counter:
                        got.append(float(row[cv]))
            if not got:
                return None
            vals[counter] = sum(got) / len(got)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # KB units; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads (all of this kernel's reads): 2 F + W
    return {"hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
            "how": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child processes, one launch each), "
                   "2 x FETCH_SIZE + WRITE_SIZE"}


def git_commit():
    try:
        return subprocess.check_output(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL,
                                       text=True).strip()
    except Exception:
        return os.environ.get("NZ_COMMIT")      # a snapshot without .git (the GPU box): the caller may say


def scs_config4(device, games_per_tree=1):
    """BASELINE.json configs[3] (square-conv form, the pinned one): one whole self-play round, in-library move loop.
    games_per_tree > 1: the round has that many games per concurrent tree (nz_scs_search_play_round)."""
    import torch
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsGameConfig, ScsSelfPlay
    from nuzero_amd.weights import synthetic_weights, convnet_param_shapes
    cfg = ScsGameConfig(os.path.join(REPO, "tests", "golden", "scs_configs", "mirrored_5x5.yml"))
    search = {"Simulation": {"mcts_simulations": 200, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}      # Configs/Search/a1_search_config.yaml
    G = 1024
    N = G * games_per_tree
    net = BoardNet("convnet", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=32, num_blocks=8, max_batch=G, device=device)
    net.set_weights(synthetic_weights(0, convnet_param_shapes(cfg.channels, cfg.planes, 3, 32, 8)))
    sp = ScsSelfPlay(cfg, search, G, device=device)
    sp.play_native(net, range(G), max_moves=2)            # warm-up: first launches, allocations
    torch.cuda.synchronize()
    sp.persist_profile(1)                                 # HIP events around the persistent kernel, on its stream
    t0 = time.perf_counter()
    r = sp.play_round(net, range(10 ** 6, 10 ** 6 + N))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": N / dt, "unit": "games/s", "expansions_per_s": r["expansions"] / dt,
           "simulations_per_s": r["simulations"] / dt, "seconds": dt, "waves": r["waves"], "games_per_round": N,
           "net_tflops_algorithmic": r["expansions"] * net.flops_per_position / dt / 1e12,
           "route": "persistent kernel (one launch per move: every game's whole search, network included)"
                    if sp.persistent() else "wave by wave (wave_kernel + one network launch per simulation wave)",
           "workload": "SCS mirrored 5x5 map (stack 2, 86 planes, 525 actions), ConvNet(32 filters, 8 layers, 3x3 square "
                       "convs), 200 sims/move, 1024 concurrent self-play games, a1 search config, 1 GPU"}
    pp = sp.persist_profile()
    if sp.persistent() and pp["launches"]:
        # the dominant kernel of this workload against the matrix pipes it issues on: bf16 MFMA FLOPs executed (16x16x32:
        # 16,384 each; six split terms per float32 product; 25 cells in two 16-row tiles) over its HIP-event time
        executed = r["expansions"] * pp["mfma_per_position"] * 16384 / (pp["ms"] * 1e-3) / 1e12
        out["roofline_scs"] = {"bound": "mfma", "kernel": "persist_kernel (SCS tree search + rules + ConvNet forward per wavefront pair)",
                               "achieved": executed, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": executed / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_us": pp["ms"] * 1e3 / pp["launches"], "launches": pp["launches"],
                               "kernel_seconds": pp["ms"] * 1e-3,
                               "positions_per_launch": r["expansions"] / pp["launches"],
                               "mfma_per_position": pp["mfma_per_position"],
                               "algorithmic_f32": {"tflops": r["expansions"] * pp["flops_per_position"] / (pp["ms"] * 1e-3) / 1e12,
                                                   "flops_per_position": pp["flops_per_position"]},
                               "note": "latency-bound, not throughput-bound: 1024 games are 4 per CU, one simulation in "
                                       "flight per tree (Explorer.py:49-61), so a game's network pass runs on two wavefronts"}
    sp.close()
    net.close()
    return out


def scs_config5(device, games=1024, moves=1):
    """BASELINE.json configs[4] on one GPU, bounded: SCS 10x10 map, RecurrentNet(86 -> 21, 256 filters, 2 blocks, recall,
    relu value head; Run.py:148) with 16 recurrent iterations, 400 simulations per move -- `games` games, the first
    `moves` decisions of each (a whole game is ~120 decisions x 400 evaluations x 9 GFLOP).  Wave-by-wave route: the
    per-layer board-net kernels (a 256-wide net on 100 cells has no one-launch form).  configs[4] does not fix the number
    of concurrent games: conv_wide_kernel's tiles are (cell, 128 channels, 256 positions), so 256 games are 200 workgroups
    -- 56 of the 256 CUs idle and the corner / edge cells' workgroups (4 / 6 taps) waiting for the interior's (9) --
    while 1024 games are 800 workgroups that the dispatcher evens out over the chip (same box: 16.3 k expansions/s at 256
    games, 17.5 k at 512, 19.2 k at 768, 22.0 k at 1024, 21.4 k at 1280; scripts/ab_cfg5_games.sh)."""
    import torch
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsGameConfig, ScsSelfPlay
    from nuzero_amd.weights import synthetic_weights, recurrent_net_param_shapes
    cfg = ScsGameConfig(os.path.join(REPO, "tests", "golden", "scs_configs", "ten_by_ten.yml"))
    search = {"Simulation": {"mcts_simulations": 400, "keep_subtree": True}, "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}
    net = BoardNet("recurrent", cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=256, num_blocks=2, recall=True,
                   value_activation="relu", max_batch=games, device=device)
    net.set_weights(synthetic_weights(0, recurrent_net_param_shapes(cfg.channels, cfg.planes, 256, 2, True)), 16)
    sp = ScsSelfPlay(cfg, search, games, device=device)
    warm = ScsSelfPlay(cfg, dict(search, Simulation={"mcts_simulations": 8, "keep_subtree": True}), games, device=device)
    warm.play_native(net, range(games), max_moves=1)       # untimed: every kernel of the route has run once
    warm.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = sp.play_native(net, range(10 ** 5, 10 ** 5 + games), max_moves=moves)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tflops = r["expansions"] * net.flops_per_position / dt / 1e12
    out = {"value": r["expansions"] / dt, "unit": "expansions/s", "simulations_per_s": r["simulations"] / dt,
           "decisions_per_s": games * moves / dt, "seconds": dt, "waves": r["waves"],
           "sample": "%d games x first %d decisions x 400 simulations" % (games, moves),
           "net_flops_per_position": net.flops_per_position, "net_tflops_algorithmic_f32": tflops,
           "net_frac_of_bf16_peak_as_issued": 6 * tflops / MFMA_BF16_PEAK_TFLOPS,
           "note": "end to end (rules, tree, network); the 256-filter layers issue six bf16 MFMAs per float32 product, so the "
                   "fraction of the BF16 peak as issued is 6 x the algorithmic rate / 2500",
           "workload": "SCS 10x10 map (stack 2, 86 planes, 2100 actions), RecurrentNet(256 filters, 2 blocks, recall, relu "
                       "value head) x 16 iterations, 400 sims/move, a1 search config, 1 GPU's share, bounded sample"}
    sp.close()
    net.close()
    return out


def ttt_config3_share(cfg_sims, weights, iters, device, games=1024, sims=400):
    """BASELINE.json configs[2]'s share of one GPU: 8192 Tic-Tac-Toe games over 8 GPUs = 1024 concurrent games per GPU,
    400 simulations per move, one round (one launch of the persistent kernel)."""
    import torch
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.search_config import legacy_ttt_search_config
    eng = SelfPlayEngine(legacy_ttt_search_config(sims), games, training=True, device=device, n_slots=games)
    eng.set_weights(weights, recurrent_iterations=iters)
    eng.play(base_seed=8 * 10 ** 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rounds, exp, sim = 3, 0, 0
    for i in range(rounds):
        eng.play(base_seed=8 * 10 ** 6 + (i + 1) * games)
        c = eng.counters()
        exp += c["expansions"]
        sim += c["simulations"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": rounds * games / dt, "unit": "games/s", "expansions_per_s": exp / dt,
           "simulations_per_s": sim / dt, "rounds": rounds, "games_per_round": games,
           "sims_per_move": sims,
           "note": "1024 games are spread four to a workgroup over 256 workgroups (TreeParams.slots_per_wg: packed 16 to a "
                   "workgroup they would keep a quarter of the CUs busy); a network pass still costs 16 columns of MFMAs",
           "workload": "Tic_Tac_Toe, 400 sims/move, 1024 concurrent games (one GPU's share of 8192 over 8 GPUs)"}
    eng.close()
    return out


def rounds_in_flight(cfg, weights, games, n_round, device, iters, rounds=6, depth=2):
    """`rounds` self-play rounds through a RoundPipeline of `depth` engines (see the module docstring)."""
    import torch
    from nuzero_amd.engine import SelfPlayEngine, RoundPipeline

    def make():
        e = SelfPlayEngine(cfg, n_round, training=True, device=device, n_slots=games)
        e.set_weights(weights, recurrent_iterations=iters)
        return e

    pipe = RoundPipeline(make, depth=depth)
    seed = lambda i: (7 * 10 ** 6 + i) * n_round
    for i in range(depth):                                  # warm-up: one round per engine
        pipe.submit(seed(i), next_base_seed=seed(i + depth))
    while pipe.pending:
        pipe.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    expansions = 0
    for i in range(depth, depth + rounds):
        if len(pipe.pending) == depth:
            expansions += pipe.collect()[1].counters()["expansions"]
        pipe.submit(seed(i), next_base_seed=seed(i + depth))
    while pipe.pending:
        expansions += pipe.collect()[1].counters()["expansions"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    return {"value": rounds * n_round / dt, "unit": "games/s", "expansions_per_s": expansions / dt, "rounds": rounds,
            "engines": depth, "games_per_round": n_round, "concurrent_games_per_engine": games}


def gamer_surface(cfg, weights, games, n_round, device, rounds=2, in_flight=0):
    """The reference-shaped worker round: Gamer.play_games = search + ReplayBuffer.save_game for every game + the six
    statistics per game, with the replay buffer on the device; then one training batch is drawn."""
    import torch
    from nuzero_amd.gamer import Gamer
    from nuzero_amd.network import Network_Manager
    from nuzero_amd.replay_device import DeviceReplayBuffer

    class tic_tac_toe:
        pass

    rb = DeviceReplayBuffer(window_size=4 * n_round, batch_size=2048, state_shape=(2, 3, 3), num_actions=9,
                            max_game_length=9, device=device)
    g = Gamer(rb, Network_Manager(weights), tic_tac_toe, [], 0, cfg, 2, "disabled", num_games=n_round,
              concurrent_games=games, device=device, base_seed=5 * 10 ** 6, records=False)
    g.play_games()                                         # warm-up round
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        _, stats = g.play_games()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    batch = rb.get_sample(2048, True, [], group_by_game=True)
    torch.cuda.synchronize()
    out = {"value": rounds * n_round / dt, "unit": "games/s", "rounds": rounds, "games_per_round": n_round,
           "buffer_positions": rb.len(), "sample_2048_ms": (time.perf_counter() - t1) * 1e3,
           "includes": "nz_engine_play, export, nz_replay_append of every position, per-game statistics dicts"}
    assert len(stats) == n_round and len(batch) == 2048
    if in_flight < 2:
        rb.close()
        g.engine.close()
        return out
    # the same worker in the trainer's asynchronous mode: Gamer.play_forever with rounds in flight
    done = []

    def on_round(records, st):
        done.append(len(st))
        if len(done) >= 2 + 2 * rounds:                    # two rounds of warm-up (the second engine's first launches)
            g.stop()

    marks = []
    import threading
    t_async = threading.Thread(target=lambda: g.play_forever(rounds_in_flight=in_flight, on_round=lambda r, st: (on_round(r, st), marks.append(time.perf_counter()))))
    t_async.start()
    t_async.join()
    if len(marks) >= 2 + 2 * rounds:
        out["play_forever_in_flight"] = {"engines": in_flight, "value": 2 * rounds * n_round / (marks[1 + 2 * rounds] - marks[1]), "unit": "games/s",
                                           "rounds": 2 * rounds}
    rb.close()
    g.engine.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games (slots) per GPU")
    ap.add_argument("--round", type=int, default=0,
                    help="games per self-play round (= per launch of the persistent kernel) per GPU; default 16 x --games: "
                         "a launch starts with every workgroup's first network pass and ends with workgroups that have run "
                         "out of games, a fixed cost that rounds of 4 x --games (key `round_4x` under --rounds-in-flight, the default until round 2) "
                         "pay four times as often")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--iters", type=int, default=2, help="recurrent iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel measurements after the timed steps")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc); the "
                         "stored measurement of profiles/ is quoted instead")
    ap.add_argument("--rounds-in-flight", type=int, default=0,
                    help="also measure the rounds with this many engines in flight (key rounds_in_flight_N); not part of "
                         "the default run: overlapped launches would spoil the kernel's average duration in a "
                         "rocprofv3 trace of the same command")
    args = ap.parse_args()
    n_round = args.round if args.round > 0 else 16 * args.games
    n_small = 4 * args.games                  # the round size of `round_4x`, `gamer_surface` and `rounds_in_flight_N`

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process starts the N ranks itself (one process per GPU) BEFORE it
        # makes any GPU call, relays rank 0's JSON line and exits with the launcher's code; it never touches a GPU
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")

    # the CPU baseline runs first, in worker processes, before this process touches the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.sims, args.cpu_seconds)
    # ... and so do the two counter passes for roofline.traffic (child processes under rocprofv3)
    traffic_live = None
    if world == 1 and rank == 0 and not args.no_extras and not args.no_live_traffic:
        traffic_live = live_traffic(args, n_round)

    import torch
    from nuzero_amd.engine import SelfPlayEngine
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    from nuzero_amd.search_config import legacy_ttt_search_config
    from nuzero_amd import dist as nzdist

    visible = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    if visible < world or local_rank >= visible:
        raise SystemExit(f"{world} GPUs requested, {visible} visible on this node (rank {rank}, local rank {local_rank})")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = legacy_ttt_search_config(args.sims)
    # the rank that would train owns the weights; one broadcast hands them to every self-play rank (Gamer.py:40,61)
    weights = synthetic_recurrent_net_weights(0, 2, 1, 64, 2, True) if rank == 0 else None
    weights = nzdist.broadcast_weights(weights, src=0, device=torch.device("cuda", local_rank)) if world > 1 else weights
    eng = SelfPlayEngine(cfg, n_round, training=True, device=local_rank, n_slots=args.games)
    eng.set_weights(weights, recurrent_iterations=args.iters)
    # rank 0 owns the shared replay buffer (in its HBM): every round's games of every rank reach it through one RCCL
    # gather + one append (ReplayBuffer.save_game for each gathered game, rank-major)
    gather = shared = None
    if world > 1:
        if rank == 0:
            from nuzero_amd.replay_device import DeviceReplayBuffer
            shared = DeviceReplayBuffer(window_size=2 * world * n_round, batch_size=2048, state_shape=(2, 3, 3),
                                        num_actions=9, max_game_length=9, device=local_rank)
        gather = nzdist.ReplayGather(eng, world, rank, buffer=shared)

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def step(i):
        # every game of every rank and round has its own stream seed
        eng.play(base_seed=(i * world + rank) * n_round, next_base_seed=((i + 1) * world + rank) * n_round)
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
                      "splits, float32 accumulation (as accurate as a float32 convolution, DESIGN.md section 5); "
                      "tree statistics float64",
        "config": {"workload": "Tic_Tac_Toe, %d sims/move, %d concurrent self-play games per GPU (%d games per "
                               "round), RecurrentNet(2,1,64,2) %d recurrent iterations f32, tree statistics f64, "
                               "legacy TTT search config" % (args.sims, args.games, n_round, args.iters),
                   "concurrent_games_per_gpu": args.games, "games_per_round_per_gpu": n_round,
                   "sims_per_move": args.sims,
                   "parallelism": "games sharded by rank, weights broadcast once, 1 RCCL gather per round"
                                  if world > 1 else "1 GPU"},
        "expansions_per_s": exp_total / dt, "simulations_per_s": sims_total / dt,
        "commit": git_commit(),
    }
    if gather is not None and rank == 0:
        out["replay_gather"] = {"ranks_seen": gather.ranks_seen, "games_saved": gather.games_saved,
                                "buffer_positions": shared.len(), "buffer_games": shared.played_games()}

    if not args.no_extras and world == 1:
        flops_pos = eng.net_flops_per_position()
        bf16_pos, f32_pos = eng.net_matrix_flops_per_position()

        def mfma_roofline(kernel, positions, ms, launches, traffic=None, traffic_at=None):
            """A kernel that runs the fused network: priced by the bf16 MFMAs it issues against the dense BF16 peak."""
            executed = positions * bf16_pos / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": kernel, "achieved": executed, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": executed / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_measured_at": traffic_at,
                    "achieved_note": "bf16 MFMA FLOPs executed (v_mfma_f32_16x16x32_bf16, six split terms per float32 "
                                     "product, zero-padded channels of 32-channel K groups included)",
                    "avg_launch_us": ms * 1e3 / max(launches, 1), "launches": launches,
                    "positions_per_launch": positions / max(launches, 1),
                    "matrix_flops_per_position": {"bf16": bf16_pos, "f32": f32_pos},
                    "algorithmic_f32": {"tflops": positions * flops_pos / (ms * 1e-3) / 1e12,
                                        "flops_per_position": flops_pos,
                                        "note": "float32 conv FLOPs of the network, in-bounds taps only; for scale: "
                                                "the FP32 MFMA peak is %.1f TFLOP/s" % MFMA_F32_PEAK_TFLOPS}}

        # ---- dominant kernel: HIP events around the persistent kernel, on its own stream, one more round
        eng.profile(True)
        eng.play(base_seed=10 ** 6 + rank * n_round)
        prof = eng.profile_read()
        eng.profile(False)
        pc = eng.counters()
        k_ms, k_n = prof["search"]["ms"], prof["search"]["launches"]
        # HBM bytes per launch from PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs, corrected as the
        # guide prescribes): a stored measurement, quoted with its commit, only for the configuration it was made on
        traffic = traffic_at = None
        if traffic_live is not None:
            traffic, traffic_at = traffic_live["hbm_bytes_per_launch"], "this run"
        else:
            for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
                tpath = os.path.join(REPO, "profiles", name)
                if os.path.exists(tpath):
                    with open(tpath) as f:
                        tj = json.load(f)
                    if tj.get("config") == [args.games, n_round, args.sims, args.iters]:
                        traffic, traffic_at = tj["hbm_bytes_per_launch"], "stored: profiles/%s at commit %s" % (name, tj.get("commit"))
                        break
        out["roofline"] = mfma_roofline("selfplay_kernel (persistent: tree phases + fused RecurrentNet forward)",
                                        pc["expansions"], k_ms, k_n, traffic, traffic_at)
        if traffic_live is not None:
            out["roofline"]["traffic_detail"] = traffic_live

        # ---- in-kernel phase shares (stamped diagnostic build; its run time is only used for the tree-phase figure)
        eng.phase_stamps(True)
        eng.profile(True)
        eng.play(base_seed=2 * 10 ** 6 + rank * n_round)
        sp_ms = eng.profile_read()["search"]["ms"]
        eng.profile(False)
        ph = eng.phase_stamps(False, read=True)
        out["phases"] = ph
        tc = eng.counters()
        # SURVEY.md 8(d) algorithmic bytes: select 11 + 20 k per scored node with k children; backup 24 per path
        # node (levels + 1 per simulation); expand 5 A + 7 per expansion + 22 per child created (A = 9)
        sel_b = 11 * tc["select_nodes"] + 20 * tc["select_children"]
        bak_b = 24 * (tc["select_nodes"] + tc["simulations"])
        exp_b = (5 * 9 + 7) * tc["expansions"] + 22 * tc["new_nodes"]
        tree_s = sp_ms * 1e-3 * ph["tree_share"] * ph["mean_over_max_lifetime"]
        tree_gbs = (sel_b + bak_b + exp_b) / tree_s / 1e9
        out["roofline_tree_phase"] = {
            "bound": "hbm", "kernel": "selfplay_kernel, tree phases (select + backup + expand + move bookkeeping of 16 "
                                      "games per workgroup between two network passes)",
            "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tree_gbs / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes": {"select": sel_b, "backup": bak_b, "expand": exp_b},
            "tree_phase_seconds": tree_s, "simulations_per_s_in_phase": tc["simulations"] / tree_s,
            "note": "latency-bound, not bandwidth-bound: one simulation at a time per tree (Explorer.py:49-61), 16 trees "
                    "per CU, every level a dependent load; the phase is %.0f %% of the kernel" % (100 * ph["tree_share"])}

        # ---- the network kernel alone
        x = (torch.rand((4096, 2, 3, 3), device="cuda") > 0.6).float()
        eng.net_forward(x, want_probs=False)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(20):
            eng.net_forward(x, want_probs=False)
        pn = eng.profile_read()["network"]
        eng.profile(False)
        out["roofline_net"] = mfma_roofline("net_kernel (fused RecurrentNet forward, 4096 positions)", 20 * 4096,
                                            pn["ms"], 20)

        # ---- the lock-step tree kernel against HBM (select 11 + 20 k per scored node, backup 24 per path node):
        #      table evaluator, so advance_kernel does a whole move's select/expand/backup per launch
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
                         + 24 * (lc["select_nodes"] + lc["simulations"])
                         + (5 * 9 + 7) * lc["expansions"] + 22 * lc["new_nodes"])
            sel[n_trees] = {"achieved": sel_bytes / (pl["ms"] * 1e-3) / 1e9,
                            "bytes_per_launch": sel_bytes / max(pl["launches"], 1),
                            "avg_launch_us": pl["ms"] * 1e3 / max(pl["launches"], 1), "launches": pl["launches"],
                            "simulations_per_s": lc["simulations"] / (pl["ms"] * 1e-3)}
            ls.close()
        big = sel[65536]
        out["roofline_select"] = {"bound": "hbm", "kernel": "advance_kernel (lock-step route: select + expand + backup, "
                                  "table evaluator, 65536 concurrent trees; not on the product route)",
                                  "achieved": big["achieved"], "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": big["achieved"] / HBM_PEAK_GBS, "traffic": None,
                                  "bytes_per_launch": big["bytes_per_launch"], "avg_launch_us": big["avg_launch_us"],
                                  "launches": big["launches"], "simulations_per_s": big["simulations_per_s"],
                                  "at_workload_trees": dict(sel[args.games], trees=args.games)}
        eng.close()
        # ---- the reference-shaped surface and one SCS configuration, driver-timed
        out["gamer_surface"] = gamer_surface(cfg, weights, args.games, n_round, local_rank)
        out["gamer_surface"]["vs_kernel_rate"] = out["gamer_surface"]["value"] / out["value"]
        if args.rounds_in_flight > 1:
            # rounds of 4 x concurrent games (what `value` was measured on until round 2), one after the other and
            # overlapped; behind the flag so that the default command launches the persistent kernel at ONE size and
            # its rocprofv3 average stays comparable with roofline.avg_launch_us
            small = SelfPlayEngine(cfg, n_small, training=True, device=local_rank, n_slots=args.games)
            small.set_weights(weights, recurrent_iterations=args.iters)
            small.play(base_seed=3 * 10 ** 6, next_base_seed=3 * 10 ** 6 + n_small)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(1, 4):
                small.play(base_seed=3 * 10 ** 6 + i * n_small, next_base_seed=3 * 10 ** 6 + (i + 1) * n_small)
            torch.cuda.synchronize()
            out["round_4x"] = {"value": 3 * n_small / (time.perf_counter() - ts), "unit": "games/s", "rounds": 3,
                               "games_per_round": n_small, "concurrent_games": args.games}
            small.close()
            gs = gamer_surface(cfg, weights, args.games, n_small, local_rank, in_flight=args.rounds_in_flight)
            gs["vs_round_4x"] = gs["value"] / out["round_4x"]["value"]
            out["gamer_surface_round_4x"] = gs
            key = "rounds_in_flight_%d" % args.rounds_in_flight
            out[key] = rounds_in_flight(cfg, weights, args.games, n_small, local_rank, args.iters, depth=args.rounds_in_flight)
            out[key]["vs_round_4x"] = out[key]["value"] / out["round_4x"]["value"]
        out["scs_config4"] = scs_config4(local_rank)
        out["scs_config4_round4"] = scs_config4(local_rank, games_per_tree=4)
        out["scs_config5"] = scs_config5(local_rank)
        out["ttt_config3_share"] = ttt_config3_share(cfg, weights, args.iters, local_rank)

    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
