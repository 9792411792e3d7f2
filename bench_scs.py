#!/usr/bin/env python3
"""Measurement of the SCS self-play path (BASELINE.json configs[3] shape: SCS 5x5 map, 200
sims/move, 1024 games, 1 GPU).  Not the driver's bench (that is bench.py).  The SCS path is
lock-step: tree, rules and legal masks on the device (nz_scs_search_*), the leaf batch of every
simulation wave evaluated by the reference's ConvNet with square 3x3 convs (hexagdly is not
available, so the hex form cannot run or be pinned) -- either by the hand-written MFMA kernels
(nz_boardnet_*, --evaluator native) or by PyTorch/MIOpen (--evaluator torch).

    python bench_scs.py [--games 1024] [--sims 200] [--filters 32] [--layers 8] [--evaluator native|torch]

Several GPUs (weak scaling, --games per GPU; games are independent, so each rank plays its own shard with its own
engine and the finished games are gathered on rank 0 with one RCCL gather per round):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_scs.py --gpus N
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--arch", choices=["convnet", "resnet", "recurrent"], default="convnet")
    ap.add_argument("--filters", type=int, default=32)
    ap.add_argument("--layers", type=int, default=8, help="ConvNet layers / ResNet or RecurrentNet blocks")
    ap.add_argument("--iters", type=int, default=1, help="recurrent iterations (RecurrentNet)")
    ap.add_argument("--hex", action="store_true", help="hex=True nets (hexagdly.Conv2d(kernel_size=1) everywhere; parity "
                                                        "unpinned: hexagdly is not installed), native evaluator only")
    ap.add_argument("--evaluator", choices=["native", "torch"], default="native")
    ap.add_argument("--loop", choices=["library", "python"], default="library",
                    help="library: nz_scs_search_play (native evaluator only); python: one host round trip per wave")
    ap.add_argument("--nodes-per-sim", type=int, default=0,
                    help="tree arena per game = 1 + sims * this many nodes (32 B each; never freed within a game); "
                         "0 = the engine's board-scaled default")
    ap.add_argument("--cache", type=int, default=0,
                    help="entries of the device inference cache (the reference's cache_choice keyless / cache_max); 0: off")
    ap.add_argument("--round-games", type=int, default=0,
                    help="games per round (default: one per concurrent game); more: finished slots start the round's next "
                         "game (nz_scs_search_play_round)")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the concurrent games into this many independent sets, each with its own engine, network "
                         "buffers, host thread and HIP stream (native evaluator, library loop): their small kernels overlap")
    ap.add_argument("--cu-split", action="store_true",
                    help="with --streams S: every set's stream gets its own 1/S of the compute units (hipExtStreamCreateWithCUMask), "
                         "so that one set's network workgroups and another's rule waves do not share a CU; run with "
                         "NZ_BOARDNET_CUS=<CUs per set> so that the one-launch network sizes its grid for the share")
    ap.add_argument("--config", default=os.path.join(REPO, "tests", "golden", "scs_configs", "mirrored_5x5.yml"))
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    from nuzero_amd import dist as nzdist
    from nuzero_amd.boardnet import BoardNet
    from nuzero_amd.scs import ScsGameConfig, ScsSelfPlay
    from nuzero_amd.weights import (synthetic_weights, convnet_param_shapes, resnet_param_shapes,
                                    recurrent_net_param_shapes)
    cfg = ScsGameConfig(args.config)
    # the reference's nets with hex=False and random-init weights: ConvNet(in, policy, 3, filters, layers),
    # ResNet(in, policy, filters, blocks), RecurrentNet(in, policy, filters, blocks, recall=True)
    if args.arch == "convnet":
        shapes = convnet_param_shapes(cfg.channels, cfg.planes, 3, args.filters, args.layers)
    elif args.arch == "resnet":
        shapes = resnet_param_shapes(cfg.channels, cfg.planes, args.filters, args.layers)
    else:
        shapes = recurrent_net_param_shapes(cfg.channels, cfg.planes, args.filters, args.layers, True)
    if args.hex:
        from nuzero_amd.weights import hex_param_shapes
        assert args.evaluator == "native", "--hex needs the native evaluator"
        shapes = hex_param_shapes(shapes)
    w = synthetic_weights(0, shapes)
    net_name = {"convnet": "ConvNet(%d filters, %d layers", "resnet": "ResNet(%d filters, %d blocks",
                "recurrent": "RecurrentNet(%d filters, %d blocks"}[args.arch] % (args.filters, args.layers)
    if args.arch == "recurrent":
        net_name += ", %d iterations" % args.iters
    if args.evaluator == "native":
        net = BoardNet(args.arch, cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=args.filters,
                       num_blocks=args.layers, kernel_size=3, max_batch=args.games, device=local, hex=args.hex)
        net.set_weights(w, args.iters)
        ev = net.evaluator()
    else:
        assert args.arch == "convnet", "the PyTorch comparison path is written for ConvNet"
        torch.backends.cudnn.benchmark = True        # let MIOpen pick a solver for the (fixed) leaf-batch shape
        wd = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
        conv = lambda t, name: F.conv2d(t, wd[name], None, 1, "same")

        def ev(images):
            n = images.shape[0]
            if n < args.games:                       # one batch shape: MIOpen tunes once
                images = torch.cat([images, images.new_zeros((args.games - n,) + tuple(images.shape[1:]))], 0)
            with torch.no_grad():
                t = F.elu(conv(images, "general_module.0.weight"))
                for i in range(args.layers):
                    t = F.elu(conv(t, f"general_module.{2 * (i + 1)}.weight"))
                p = conv(F.relu(conv(t, "policy_head.layers.0.weight")), "policy_head.layers.2.weight")
                v = t
                for i in range(4):
                    v = conv(v, f"value_head.layers.{2 * i}.weight")
                    if i != 3:
                        v = torch.tanh(v)
                probs = torch.softmax(p.reshape(p.shape[0], -1), 1)
                return probs[:n].contiguous(), torch.tanh(v.mean(dim=(1, 2, 3)))[:n].contiguous()
    search = {"Simulation": {"mcts_simulations": args.sims, "keep_subtree": True},
              "UCT": {"pb_c_base": 10000, "pb_c_init": 1.15},
              "Exploration": {"number_of_softmax_moves": 0, "epsilon_softmax_exploration": 0.04,
                              "epsilon_random_exploration": 0.001, "value_factor": 1,
                              "root_exploration_distribution": "gamma", "root_exploration_fraction": 0.2,
                              "root_dist_alpha": 0.15, "root_dist_beta": 1}}      # Configs/Search/a1_search_config.yaml
    sp = ScsSelfPlay(cfg, search, args.games, nodes_per_game=(1 + args.sims * args.nodes_per_sim) if args.nodes_per_sim else None,
                     device=local)
    n_round = max(args.round_games, args.games)
    if args.cache > 0:
        sp.cache(args.cache)
    seeds = range(rank * n_round, (rank + 1) * n_round)             # game index = rank * games per round + g
    ev(torch.zeros((1, cfg.channels, cfg.rows, cfg.cols), device="cuda"))     # solver search outside the timed region
    torch.cuda.synchronize()
    if world > 1:
        td.barrier()
    t0 = time.perf_counter()
    if args.evaluator == "native" and args.loop == "library" and args.streams > 1:
        import threading
        S = args.streams
        assert args.games % S == 0
        per = args.games // S
        sets = []

        def masked_stream(i):
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            n_cu = torch.cuda.get_device_properties(local).multi_processor_count
            words = (n_cu + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for cu in range(i * n_cu // S, (i + 1) * n_cu // S):
                mask[cu // 32] |= 1 << (cu % 32)
            st = ctypes.c_void_p()
            err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
            assert err == 0, "hipExtStreamCreateWithCUMask: %d" % err
            return torch.cuda.ExternalStream(st.value, device=local)
        for i in range(S):
            n_i = BoardNet(args.arch, cfg.channels, cfg.planes, cfg.rows, cfg.cols, width=args.filters,
                           num_blocks=args.layers, kernel_size=3, max_batch=per, device=local, hex=args.hex)
            n_i.set_weights(w, args.iters)
            sets.append((ScsSelfPlay(cfg, search, per, device=local,
                                     nodes_per_game=(1 + args.sims * args.nodes_per_sim) if args.nodes_per_sim else None), n_i,
                         masked_stream(i) if args.cu_split else torch.cuda.Stream(device=local, priority=-1)))
        results = [None] * S

        def run(i):
            sp_i, n_i, st = sets[i]
            with torch.cuda.stream(st):
                results[i] = sp_i.play_native(n_i, seeds[i * per:(i + 1) * per])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=run, args=(i,)) for i in range(S)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert all(x is not None for x in results), "a stream's games failed"
        r = {k: np.concatenate([x[k] for x in results], 0) for k in results[0] if isinstance(results[0][k], np.ndarray)}
        for k in ("simulations", "expansions", "waves"):
            r[k] = sum(x[k] for x in results)
    elif args.evaluator == "native" and args.loop == "library":
        r = sp.play_round(net, seeds) if n_round > args.games else sp.play_native(net, seeds)
    else:
        r = sp.play(ev, seeds=seeds)
    if world > 1:                                  # the round's games to rank 0's replay buffer
        nzdist.gather_payload(nzdist.scs_payload(r, torch.device("cuda", local)), world, rank, 0, nzdist.SCS_FIELDS)
    torch.cuda.synchronize()
    if world > 1:
        td.barrier()
    dt = time.perf_counter() - t0
    totals = torch.tensor([dt, float(n_round), float(r["expansions"]), float(r["simulations"])], dtype=torch.float64,
                          device="cuda")
    if world > 1:
        tmax = totals.clone()
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        td.all_reduce(totals, op=td.ReduceOp.SUM)
        totals[0] = tmax[0]
    dt, n_games, n_exp, n_sim = [float(v) for v in totals.cpu()]
    if rank != 0:
        td.destroy_process_group()
        return
    out = {}
    native = args.evaluator == "native"
    if args.cache > 0:
        out["cache"] = sp.cache_stats()
    if os.environ.get("NZ_LIB_PATH") and args.streams == 1:      # a diagnostic build may carry the wave kernel's phase stamps
        out["wave_kernel_phase_ticks"] = sp.phase_ticks()
        out["persist_kernel_ticks"] = sp.persist_ticks()
    print(json.dumps({"workload": "SCS %dx%d stack %d, %d sims/move, %d concurrent games, %s, %s convs), %s evaluator, "
                                  "%s move loop" % (cfg.rows, cfg.cols, cfg.stacking, args.sims, args.games, net_name,
                                                    "hexagonal (unpinned)" if args.hex else "square",
                                                    args.evaluator, args.loop if native else "python"),
                      "net_flops_per_position": net.flops_per_position if native else None,
                      "net_tflops": n_exp * net.flops_per_position / dt / 1e12 if native else None,
                      "waves": r.get("waves"), "n_gpus": world, "scaling": "weak", "streams": args.streams,
                      "games_per_s": n_games / dt, "expansions_per_s": n_exp / dt,
                      "simulations_per_s": n_sim / dt, "seconds": dt,
                      "games_per_round": int(n_games), "mean_game_length": float(r["lengths"].mean()),
                      "game_length_p50_p90_max": [int(v) for v in np.percentile(r["lengths"], [50, 90, 100])],
                      "outcomes": {str(v): int((r["outcomes"] == v).sum()) for v in (-1, 0, 1)}, **out}))
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
