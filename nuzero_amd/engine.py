"""Python face of the C ABI: owns one nz_engine and the PyTorch-ROCm tensors it
writes into.  PyTorch is used for device memory and streams only."""
import ctypes
from ctypes import byref, c_double, c_int32, c_int64, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import lib, check
from .search_config import to_struct


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class SelfPlayEngine:
    """G concurrent Tic-Tac-Toe self-play games on one GPU.

    ``search_config`` is the reference's nested dict; ``training`` is the second
    argument of the reference's ``Explorer(search_config, training)``.
    """

    def __init__(self, search_config, n_games, training=True, device=0, negate_player=2, n_slots=None):
        if not torch.cuda.is_available():
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        self.search_config = search_config
        self.training = bool(training)
        cfg = to_struct(search_config, training)
        game = _lib.GameDesc(game=_lib.NZ_GAME_TIC_TAC_TOE, negate_player=negate_player)
        self._h = c_void_p(0)
        with torch.cuda.device(self.device):
            check(lib.nz_engine_create_ex(byref(self._h), byref(cfg), byref(game),
                                          int(n_slots if n_slots else n_games), int(n_games), int(device)))
        d = _lib.Dims()
        check(lib.nz_engine_dims(self._h, byref(d)), self._h)
        self.n_games, self.num_actions, self.max_moves = d.n_games, d.num_actions, d.max_moves
        self.n_slots = d.n_slots
        self.state_shape = (d.state_channels, d.rows, d.cols)
        self.node_capacity = d.node_capacity
        self.net_spec = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.nz_engine_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- network ---------------------------------------------------------------
    def set_weights(self, weights, width=64, num_blocks=2, recall=True, value_activation="tanh",
                    recurrent_iterations=2, in_channels=2, policy_channels=1, arch="recurrent", kernel_size=3):
        """``weights``: name -> array/tensor with the reference's state_dict keys (hex=False
        RecurrentNet, ResNet or ConvNet; for ConvNet num_blocks = num_layers), in state_dict order."""
        tensors = []
        for v in weights.values():
            t = v.detach() if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
            tensors.append(t.to(torch.float32).contiguous())
        ptrs = (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        nd = _lib.NetDesc(in_channels=in_channels, policy_channels=policy_channels, width=width,
                          num_blocks=num_blocks, recall=int(recall),
                          value_activation=_lib.NZ_ACT_RELU if value_activation == "relu" else _lib.NZ_ACT_TANH,
                          arch={"recurrent": _lib.NZ_ARCH_RECURRENT, "resnet": _lib.NZ_ARCH_RESNET,
                                "convnet": _lib.NZ_ARCH_CONVNET}[arch], kernel_size=kernel_size)
        for t in tensors:
            if t.is_cuda:
                torch.cuda.synchronize(t.device)
        check(lib.nz_engine_set_weights(self._h, byref(nd), ptrs, len(tensors), int(recurrent_iterations)), self._h)
        self.net_spec = dict(width=width, num_blocks=num_blocks, recall=recall, iters=recurrent_iterations)
        self.position_cache = False

    def set_table(self, table):
        """Test hook: [19683, 10] float32 table evaluator (9 probs + value)."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        check(lib.nz_engine_set_table(self._h, t.ctypes.data_as(c_void_p), int(t.shape[0])), self._h)

    def cache_all_positions(self):
        """Inference cache for Tic-Tac-Toe (the reference's optional Cache, Utils/Caches, used at
        Explorer.py:146-155): the game has only 3^9 position codes, so instead of memoising leaf by
        leaf the network is evaluated ONCE on every code and the search then reads (softmax probs,
        value) from that table -- a cache with a 100 % hit ratio and the same results as no cache.
        Call again after set_weights.  bench.py never uses it (it would skip the measured work)."""
        if self.net_spec is None:
            raise RuntimeError("set_weights first")
        codes = torch.arange(3 ** 9, device=self.device)
        pw = 3 ** torch.arange(9, device=self.device)
        cells = (codes[:, None] // pw[None, :]) % 3                       # [N, 9] in {0,1,2}
        x = torch.stack([(cells == 1), (cells == 2)], 1).to(torch.float32).reshape(-1, 2, 3, 3)
        _, value, probs = self.net_forward(x)
        table = torch.cat([probs, value[:, None]], 1).contiguous()
        torch.cuda.synchronize(self.device)
        check(lib.nz_engine_set_table(self._h, _ptr(table), int(table.shape[0])), self._h)
        self.position_cache = True

    def net_flops_per_position(self):
        v = c_double(0)
        check(lib.nz_engine_net_flops(self._h, byref(v)), self._h)
        return v.value

    def net_matrix_flops_per_position(self):
        """(bf16, f32) FLOPs the matrix cores execute per position (six split terms per float32 product)."""
        a, b = c_double(0), c_double(0)
        check(lib.nz_engine_net_matrix_flops(self._h, byref(a), byref(b)), self._h)
        return a.value, b.value

    def net_forward(self, states, want_probs=True):
        """Network_Manager.inference for a float32 [B, 2, 3, 3] batch on the GPU.
        Returns (logits [B, 9], value [B], probs [B, 9] or None)."""
        x = torch.as_tensor(states, dtype=torch.float32).to(self.device).contiguous()
        b = x.shape[0]
        logits = torch.empty((b, 9), dtype=torch.float32, device=self.device)
        value = torch.empty((b,), dtype=torch.float32, device=self.device)
        probs = torch.empty((b, 9), dtype=torch.float32, device=self.device) if want_probs else None
        with torch.cuda.device(self.device):
            check(lib.nz_net_forward(self._h, _ptr(x), b, _ptr(logits), _ptr(value), _ptr(probs), _stream()), self._h)
        return logits, value, probs

    def net_forward_stamps(self, states):
        """Diagnostic build: mean ticks wave 0 of a workgroup spends computing jobs / waiting at barriers."""
        x = torch.as_tensor(states, dtype=torch.float32).to(self.device).contiguous()
        b = x.shape[0]
        logits = torch.empty((b, 9), dtype=torch.float32, device=self.device)
        value = torch.empty((b,), dtype=torch.float32, device=self.device)
        out = (c_double * 4)()
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            check(lib.nz_net_forward_stamps(self._h, _ptr(x), b, _ptr(logits), _ptr(value), out), self._h)
        return dict(zip(("k_loops", "input_planes", "epilogues", "barriers"), list(out)))

    # ---- stepping --------------------------------------------------------------
    def reset(self):
        with torch.cuda.device(self.device):
            check(lib.nz_engine_reset(self._h, _stream()), self._h)

    def root_children(self):
        out = torch.empty((self.n_games,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.nz_engine_root_children(self._h, _ptr(out), _stream()), self._h)
        return out

    def move(self, noise=None, uniforms=None):
        """One move for every live game.  noise: float64 [G, A]; uniforms: float64 [G, 3]."""
        if noise is not None:
            noise = torch.as_tensor(noise, dtype=torch.float64).to(self.device).contiguous()
            uniforms = torch.as_tensor(uniforms, dtype=torch.float64).to(self.device).contiguous()
        with torch.cuda.device(self.device):
            check(lib.nz_engine_move(self._h, _ptr(noise), _ptr(uniforms), _stream()), self._h)

    def search(self, noise=None):
        """Root noise (training) + simulations for every live game; no action yet."""
        if noise is not None:
            noise = torch.as_tensor(noise, dtype=torch.float64).to(self.device).contiguous()
        with torch.cuda.device(self.device):
            check(lib.nz_engine_search(self._h, _ptr(noise), _stream()), self._h)

    def apply(self, actions=None, uniforms=None):
        """Take `actions` (int32 [G]; None = the search's own choice), step and re-root."""
        if actions is not None:
            actions = torch.as_tensor(actions, dtype=torch.int32).to(self.device).contiguous()
        if uniforms is not None:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float64).to(self.device).contiguous()
        with torch.cuda.device(self.device):
            check(lib.nz_engine_apply(self._h, _ptr(actions), _ptr(uniforms), _stream()), self._h)

    def last_actions(self):
        out = torch.empty((self.n_games,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.nz_engine_last_actions(self._h, _ptr(out), _stream()), self._h)
        return out

    def live_games(self):
        n = c_int32(0)
        with torch.cuda.device(self.device):
            check(lib.nz_engine_live_games(self._h, byref(n), _stream()), self._h)
        return n.value

    def play(self, base_seed=0, next_base_seed=None):
        """Reset and play every game to the end with the engine's own host
        random streams (game g <- RandomState(base_seed + g)).  `next_base_seed`: the seed of the round that will
        be played next -- its random numbers are drawn on the host while this round's kernel runs."""
        with torch.cuda.device(self.device):
            if next_base_seed is None:
                check(lib.nz_engine_play(self._h, int(base_seed), _stream()), self._h)
            else:
                check(lib.nz_engine_play_next(self._h, int(base_seed), 1, int(next_base_seed), _stream()), self._h)

    def play_lockstep(self, base_seed=0):
        """Same results as play() by the lock-step route (one launch sequence per move)."""
        with torch.cuda.device(self.device):
            check(lib.nz_engine_play_lockstep(self._h, int(base_seed), _stream()), self._h)

    def desync_count(self):
        n = c_int64(0)
        check(lib.nz_engine_desync_count(self._h, byref(n)), self._h)
        return n.value

    def play_with_numpy_rng(self, seeds):
        """Same as play(), but the randomness is drawn here from numpy
        RandomState objects in the reference's call order -- the slow,
        ground-truth path the parity tests use."""
        ex = self.search_config["Exploration"]
        rngs = [np.random.RandomState(int(s)) for s in seeds]
        assert len(rngs) == self.n_games
        self.reset()
        for move in range(self.max_moves):
            if self.live_games() == 0:
                break
            if not self.training:
                self.move()
                continue
            nchild = self.root_children().cpu().numpy()
            alive = self.alive().cpu().numpy()
            noise = np.zeros((self.n_games, self.num_actions), np.float64)
            uni = np.zeros((self.n_games, 3), np.float64)
            for g, rs in enumerate(rngs):
                if alive[g] == 0:
                    continue
                n = int(nchild[g])
                noise[g, :n] = rs.gamma(ex["root_dist_alpha"], ex["root_dist_beta"], n)
                if move < ex["number_of_softmax_moves"]:
                    uni[g, 2] = rs.random_sample()
                else:
                    u1, u2 = rs.random_sample(), rs.random_sample()
                    uni[g, 0], uni[g, 1] = u1, u2
                    if u1 < ex["epsilon_softmax_exploration"] or u2 < ex["epsilon_random_exploration"]:
                        uni[g, 2] = rs.random_sample()
            self.move(noise, uni)
        assert self.live_games() == 0

    def alive(self):
        out = torch.empty((self.n_games,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.nz_engine_alive(self._h, _ptr(out), _stream()), self._h)
        return out

    # ---- results ---------------------------------------------------------------
    def export(self, states=True, trace=False):
        G, T, A = self.n_games, self.max_moves, self.num_actions
        dev = self.device
        out = {
            "states": torch.empty((G, T) + self.state_shape, dtype=torch.float32, device=dev) if states else None,
            "visits": torch.empty((G, T, A), dtype=torch.int32, device=dev),
            "actions": torch.empty((G, T), dtype=torch.int32, device=dev),
            "lengths": torch.empty((G,), dtype=torch.int32, device=dev),
            "outcomes": torch.empty((G,), dtype=torch.int32, device=dev),
            "tree_size": torch.empty((G, T), dtype=torch.int32, device=dev),
            "n_children": torch.empty((G, T), dtype=torch.int32, device=dev),
            "bias": torch.empty((G, T), dtype=torch.float64, device=dev),
        }
        with torch.cuda.device(dev):
            check(lib.nz_engine_export(self._h, _ptr(out["states"]), _ptr(out["visits"]), _ptr(out["actions"]),
                                       _ptr(out["lengths"]), _ptr(out["outcomes"]), _ptr(out["tree_size"]),
                                       _ptr(out["n_children"]), _ptr(out["bias"]), _stream()), self._h)
            if trace:
                out["child_prior"] = torch.empty((G, T, A), dtype=torch.float64, device=dev)
                out["child_value_sum"] = torch.empty((G, T, A), dtype=torch.float64, device=dev)
                out["root_value_sum"] = torch.empty((G, T), dtype=torch.float64, device=dev)
                check(lib.nz_engine_export_trace(self._h, _ptr(out["child_prior"]), _ptr(out["child_value_sum"]),
                                                 _ptr(out["root_value_sum"]), _stream()), self._h)
        return {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}

    def export_device(self):
        """The replay payload as device tensors (for the multi-GPU gather)."""
        G, T, A = self.n_games, self.max_moves, self.num_actions
        dev = self.device
        out = {
            "states": torch.empty((G, T) + self.state_shape, dtype=torch.float32, device=dev),
            "visits": torch.empty((G, T, A), dtype=torch.int32, device=dev),
            "actions": torch.empty((G, T), dtype=torch.int32, device=dev),
            "lengths": torch.empty((G,), dtype=torch.int32, device=dev),
            "outcomes": torch.empty((G,), dtype=torch.int32, device=dev),
            "tree_size": torch.empty((G, T), dtype=torch.int32, device=dev),
            "n_children": torch.empty((G, T), dtype=torch.int32, device=dev),
            "bias": torch.empty((G, T), dtype=torch.float64, device=dev),
        }
        with torch.cuda.device(dev):
            check(lib.nz_engine_export(self._h, _ptr(out["states"]), _ptr(out["visits"]), _ptr(out["actions"]),
                                       _ptr(out["lengths"]), _ptr(out["outcomes"]), _ptr(out["tree_size"]),
                                       _ptr(out["n_children"]), _ptr(out["bias"]), _stream()), self._h)
        return out

    def counters(self):
        out = (c_int64 * 5)()
        with torch.cuda.device(self.device):
            check(lib.nz_engine_counters_n(self._h, out, 5, _stream()), self._h)
        return {"simulations": out[0], "expansions": out[1], "select_nodes": out[2], "select_children": out[3],
                "new_nodes": out[4]}

    # ---- kernel timing -----------------------------------------------------------
    def profile(self, enable=True):
        check(lib.nz_engine_profile(self._h, int(enable)), self._h)

    def phase_stamps(self, enable, read=False):
        """Select (or deselect) the stamped diagnostic build of the persistent kernel;
        with read=True return the last stamped run's phase shares first."""
        out = (c_double * 10)()
        check(lib.nz_engine_phase_stamps(self._h, int(enable), out if read else None), self._h)
        if read:
            return {"cycles_per_workgroup": out[0], "tree_share": out[1], "net_share": out[2],
                    "mean_over_max_lifetime": out[3], "net_ticks_per_cycle": out[4],
                    "tree_ticks_per_cycle": out[5], "max_workgroup_ticks": out[6], "workgroups": out[7],
                    "finish_move_ticks_per_cycle": out[8], "critical_sims_per_cycle": out[9]}
        return None

    def profile_read(self):
        ms = (c_double * 3)()
        n = (c_int64 * 3)()
        pos = c_int64(0)
        check(lib.nz_engine_profile_read(self._h, ms, n, byref(pos)), self._h)
        names = ("search", "network", "move_misc")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(names)}


class RoundPipeline:
    """Self-play rounds of several engines in flight at once (the reference's asynchronous mode: Gamers that
    `play_forever`, Training/Gamer.py:99-101, while the trainer consumes finished games).

    A round ends with workgroups that have run out of games while others still play; with `depth` engines, each on its
    own HIP stream and host thread, the next round's workgroups take the compute units the current round's tail leaves
    idle.  Rounds are submitted and handed back in order; round i runs on engine i % depth, so its games are exactly
    those SelfPlayEngine.play(base_seed) would play (every game has its own random stream)."""

    def __init__(self, make_engine, depth=2):
        from concurrent.futures import ThreadPoolExecutor
        self.engines = [make_engine() for _ in range(depth)]
        self.device = self.engines[0].device
        with torch.cuda.device(self.device):
            # high priority: those streams have hardware queues of their own.  Normal-priority streams share four
            # queues with the null stream, and the pair that lands on the null stream's queue loses the overlap
            # (measured: 107 k instead of 116 k games/s for every second pipeline of a process).
            self.streams = [torch.cuda.Stream(priority=-1) for _ in self.engines]
        self.pool = ThreadPoolExecutor(max_workers=depth)
        self.pending = []                 # (round index, engine, future), oldest first
        self.submitted = 0

    def _run(self, k, base_seed, next_base_seed):
        with torch.cuda.device(self.device), torch.cuda.stream(self.streams[k]):
            self.engines[k].play(base_seed=base_seed, next_base_seed=next_base_seed)
            self.streams[k].synchronize()

    def submit(self, base_seed, next_base_seed=None):
        """Start the next round on the engine whose turn it is (it must have been collected: at most `depth` rounds
        are in flight).  `next_base_seed`: the seed of the round this SAME engine will play next."""
        k = self.submitted % len(self.engines)
        assert all(e is not self.engines[k] for _, e, _ in self.pending), "collect() the engine's last round first"
        self.pending.append((self.submitted, self.engines[k], self.pool.submit(self._run, k, base_seed, next_base_seed)))
        self.submitted += 1

    def collect(self):
        """Wait for the oldest round in flight; returns (round index, its engine) -- read the engine's results before the
        next submit() that lands on it."""
        i, eng, fut = self.pending.pop(0)
        fut.result()
        return i, eng

    def close(self):
        while self.pending:
            self.collect()
        self.pool.shutdown()
        for e in self.engines:
            e.close()
