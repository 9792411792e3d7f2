"""Replay buffer on the device with the reference's surface (Training/ReplayBuffer.py:11-107).

The reference keeps one Python tuple per position -- (state tensor, (value, policy list), game_index) -- in a list,
evicts with list.pop(0), shuffles with random.shuffle and samples with np.random.choice; the trainer then builds every
batch with torch.cat / torch.tensor per sample (Training/AlphaZero.py:846-852,892-903).  Here the positions live in HBM
(C ABI nz_replay_*, nuzero_amd/csrc/replay.hip) and a batch is one gather kernel; what stays on the host is the
ORDER of the buffer (`ReplayIndex`): which physical slot holds the i-th position.  The order logic is the reference's,
driven by the reference's own generators (`random.shuffle` on the slot list, `np.random.choice` with the same
arguments), so the same seeds give the same batches -- pinned by tests/golden/replay_kat.json, which
tests/golden/make_golden_replay.py made with the genuine ReplayBuffer class.
"""
import random
from ctypes import byref, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import lib


class ReplayIndex:
    """Host side of the buffer: window in GAMES with per-POSITION eviction (ReplayBuffer.py:24-36), shuffle (:38-39),
    slice (:41-42), sample (:44-53).  `seq[i]` is the physical slot of the buffer's i-th position.  An evicted
    position's slot is reused by the position that evicts it, so the slots in use are always 0 .. len - 1."""

    def __init__(self, window_size, capacity=None):
        self.window_size = int(window_size)
        self.capacity = capacity
        self.seq = np.empty(0, np.int64)
        self.n_games = 0
        self.full = False
        self.slot_game_index = np.zeros(0, np.int32)      # host mirror: game_index of every physical slot

    def __len__(self):
        return len(self.seq)

    def save_games(self, lengths, game_index):
        """A batch of finished games, saved one after the other (save_game per game, in order).  Returns the physical
        slot of every position, int64 [G, max(lengths)], -1 where a game has no such move or where the position is
        evicted again before the batch ends (the writes of one batch may then run in any order)."""
        lengths = np.asarray(lengths, np.int64)
        G = len(lengths)
        T = int(lengths.max()) if G else 0
        dst = np.full((G, T), -1, np.int64)
        if G == 0:
            return dst
        # games that still fit under the window: no eviction, fresh slots (ReplayBuffer.py:25-29)
        k0 = int(min(G, max(0, self.window_size - self.n_games)))
        move = np.arange(T)[None, :]
        valid = move < lengths[:, None]
        if k0 > 0:
            n_new = int(lengths[:k0].sum())
            fresh = np.arange(len(self.seq), len(self.seq) + n_new, dtype=np.int64)
            dst[:k0][valid[:k0]] = fresh
            self.seq = np.concatenate([self.seq, fresh])
            self.n_games += k0
            self.full = False
        if k0 < G:
            # the window is full: every position evicts the buffer's first one and takes its slot (:34-36)
            self.full = True
            n = len(self.seq)
            P = int(lengths[k0:].sum())
            j = np.arange(P, dtype=np.int64)
            dst[k0:][valid[k0:]] = self.seq[j % n]
            self.seq = np.roll(self.seq, -(P % n))
            # a position evicted before the batch ends (it was saved earlier in this same batch) is never stored: of
            # the rows that name the same slot only the last one, in save order, keeps it
            flat = dst.reshape(-1)
            rows = np.nonzero(flat >= 0)[0]
            _, last_from_end = np.unique(flat[rows][::-1], return_index=True)
            keep = rows[len(rows) - 1 - last_from_end]
            gone = np.ones(len(flat), bool)
            gone[keep] = False
            flat[gone] = -1
        if self.capacity is not None and len(self.seq) > self.capacity:
            raise ValueError(f"replay buffer holds {len(self.seq)} positions, capacity is {self.capacity}")
        if len(self.slot_game_index) < len(self.seq):
            self.slot_game_index = np.concatenate(
                [self.slot_game_index, np.zeros(len(self.seq) - len(self.slot_game_index), np.int32)])
        self.slot_game_index[dst[dst >= 0]] = game_index
        return dst

    def shuffle(self):
        """random.shuffle(self.buffer) (ReplayBuffer.py:38-39): the same permutation for the same `random` state,
        because shuffle only looks at the length of what it permutes."""
        order = self.seq.tolist()
        random.shuffle(order)
        self.seq = np.asarray(order, np.int64).reshape(-1)

    def get_slice(self, start_index, last_index):
        return self.seq[start_index:last_index]

    def get_sample(self, batch_size, replace, probs):
        """np.random.choice with the reference's arguments (ReplayBuffer.py:44-50) on the global numpy stream."""
        if probs is None or len(probs) == 0:
            args = [len(self.seq), batch_size, replace]
        else:
            args = [len(self.seq), batch_size, replace, probs]
        return self.seq[np.random.choice(*args)]

    def bucket(self, slots):
        """Group a batch by game index, keys ascending, order kept within a group -- what batch_update_weights does
        with more_itertools.bucket + sorted (AlphaZero.py:846-848).  Returns (slots regrouped, keys, group sizes)."""
        slots = np.asarray(slots, np.int64)
        gi = self.slot_game_index[slots]
        order = np.argsort(gi, kind="stable")
        keys, counts = np.unique(gi, return_counts=True)
        return slots[order], keys.tolist(), counts.tolist()


def late_heavy_probs(num_positions, variation=0.5):
    """The sampling weights of train_with_samples(late_heavy=True) (AlphaZero.py:780-795), same float operations."""
    probs = []
    offset = (1 - variation) / 2
    fraction = variation / num_positions
    total = offset
    for _ in range(num_positions):
        total += fraction
        probs.append(total)
    total_sum = sum(probs)
    return [p / total_sum for p in probs]


class Batch:
    """A training batch on the device: states [B, C, H, W], policies [B, A], values [B] (float32), game_index [B]."""

    def __init__(self, states, policies, values, game_index, keys=None, counts=None):
        self.states, self.policies, self.values, self.game_index = states, policies, values, game_index
        self.keys, self.counts = keys, counts

    def __len__(self):
        return int(self.values.shape[0])

    def by_game(self):
        """(game_index, states, policies, values) per game type, keys ascending (AlphaZero.py:846-852); the batch must
        have been made with group_by_game=True."""
        assert self.keys is not None
        at = 0
        for k, c in zip(self.keys, self.counts):
            yield k, self.states[at:at + c], self.policies[at:at + c], self.values[at:at + c]
            at += c

    def as_list(self):
        """The reference's batch format: [(state [1, C, H, W] CPU tensor, (value, policy list), game_index)]."""
        s, p, v, g = (t.cpu() for t in (self.states, self.policies, self.values, self.game_index))
        return [(s[i:i + 1], (int(v[i]) if float(v[i]).is_integer() else float(v[i]), p[i].tolist()), int(g[i]))
                for i in range(len(self))]


class DeviceReplayBuffer:
    def __init__(self, window_size, batch_size, state_shape, num_actions, max_game_length, device=0):
        """`window_size` in games, `batch_size` as the reference's constructor (ReplayBuffer.py:14); `state_shape` =
        (C, H, W) of one position, `max_game_length` bounds the positions of one game (capacity = window x that)."""
        if not torch.cuda.is_available():
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        self.window_size, self.batch_size = int(window_size), int(batch_size)
        self.state_shape = tuple(int(x) for x in state_shape)
        self.state_floats = int(np.prod(self.state_shape))
        self.num_actions = int(num_actions)
        self.device = torch.device("cuda", device)
        self.capacity = self.window_size * int(max_game_length)
        self.index = ReplayIndex(window_size, self.capacity)
        self.step_to_size_map = {}
        self.allow_partial_loading = True
        self._h = c_void_p(0)
        st = lib.nz_replay_create(byref(self._h), self.capacity, self.state_floats, self.num_actions, int(device))
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_replay_last_error(None) or b"").decode())

    # ---- plumbing ------------------------------------------------------------------------------------------
    def _check(self, st):
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_replay_last_error(self._h) or b"").decode())

    def _stream(self):
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.nz_replay_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self):
        self._check(lib.nz_replay_check(self._h, self._stream()))

    def _append(self, states, dst, game_values, rows_per_game, game_index, visits=None, policies=None, children=None):
        dst_d = torch.from_numpy(np.ascontiguousarray(dst.reshape(-1))).to(self.device)
        n_rows = dst_d.numel()
        states = states.reshape(n_rows, self.state_floats)
        assert states.is_cuda and states.dtype == torch.float32 and states.is_contiguous()
        ptr = lambda t: c_void_p(t.data_ptr()) if t is not None else None
        ca = cv = nc = None
        max_children = 0
        if children is not None:
            ca, cv, nc = children
            max_children = int(ca.shape[-1])
        self._check(lib.nz_replay_append(self._h, ptr(states), ptr(visits), ptr(policies), ptr(ca), ptr(cv), ptr(nc),
                                         max_children, ptr(game_values), int(rows_per_game), ptr(dst_d), n_rows,
                                         int(game_index), self._stream()))

    # ---- filling ---------------------------------------------------------------------------------------------
    def save_games_from_engine(self, engine, game_index, export=None):
        """Every game of a finished Tic-Tac-Toe round of `engine` (nuzero_amd.engine.SelfPlayEngine), straight from
        the engine's export buffers in HBM: states, root visit counts and outcomes never visit the host; only the
        game lengths do (the order logic needs them).  Same buffer contents as save_game(record) game by game."""
        ex = export if export is not None else engine.export_device()
        lengths = ex["lengths"].cpu().numpy()
        dst = self.index.save_games(lengths, game_index)
        T = ex["states"].shape[1]
        full = np.full((len(lengths), T), -1, np.int64)
        full[:, :dst.shape[1]] = dst
        self._append(ex["states"].contiguous(), full, ex["outcomes"].contiguous(), T, game_index,
                     visits=ex["visits"].contiguous())
        return lengths

    def save_scs_games(self, selfplay, result_device, game_index, n_games=None):
        """The (first `n_games`) games of a finished SCS round (nuzero_amd.scs.ScsSelfPlay): the per-move state images
        are regenerated on the device by replaying the recorded actions through the device rules, one append per move;
        the policy targets come from the root's (action, visit) lists (SCS_Game.py:1517-1521)."""
        from .scs import ScsBatch
        r = result_device
        cfg, G = selfplay.cfg, int(r["lengths"].shape[0])     # rows of the round (>= selfplay.n_games after play_round)
        lengths = r["lengths"].cpu().numpy()
        n_keep = G if n_games is None else int(n_games)
        dst = np.full((G, int(lengths.max())), -1, np.int64)
        kept = self.index.save_games(lengths[:n_keep], game_index)
        dst[:n_keep, :kept.shape[1]] = kept
        batch = ScsBatch(cfg, G, device=self.device.index or 0)
        if cfg.per_game:                                      # every game is replayed on its own map
            batch.set_maps(selfplay.game_maps[0][:G], selfplay.game_maps[1][:G])
        outcomes = r["outcomes"].to(torch.int32).contiguous()
        moves = torch.arange(int(lengths.max()), device=self.device)
        actions = torch.where(moves[None, :] < r["lengths"][:, None].to(self.device), r["actions"][:, :len(moves)],
                              torch.full_like(r["actions"][:, :len(moves)], -1)).to(torch.int32)
        for m in range(int(lengths.max())):
            self._append(batch.state_image(), dst[:, m], outcomes, 1, game_index,
                         children=(r["child_action"][:, m].contiguous(), r["child_visit"][:, m].contiguous(),
                                   r["n_children"][:, m].contiguous()))
            batch.step(actions[:, m].contiguous())
        batch.close()
        return lengths[:n_keep]

    def save_game(self, game, game_index):
        """ReplayBuffer.save_game (ReplayBuffer.py:24-36) for one game object with the reference's attributes
        (state_history, get_state_from_history, make_target): uploads the game's positions."""
        n = len(game.state_history)
        dst = self.index.save_games([n], game_index)
        states = torch.cat([game.get_state_from_history(i).reshape(1, -1).float() for i in range(n)], 0)
        targets = [game.make_target(i) for i in range(n)]
        policies = torch.tensor([t[1] for t in targets])          # float32, as AlphaZero.py:901 makes them
        values = torch.tensor([int(targets[0][0])], dtype=torch.int32)
        for t in targets:
            assert t[0] == targets[0][0]
        self._append(states.to(self.device).contiguous(), dst, values.to(self.device), n, game_index,
                     policies=policies.to(self.device).contiguous())

    # ---- reading -----------------------------------------------------------------------------------------------
    def _gather(self, slots, group_by_game=False):
        keys = counts = None
        if group_by_game:
            slots, keys, counts = self.index.bucket(slots)
        slots_d = torch.from_numpy(np.ascontiguousarray(slots, dtype=np.int64)).to(self.device)
        B = int(slots_d.numel())
        states = torch.empty((B,) + self.state_shape, dtype=torch.float32, device=self.device)
        policies = torch.empty((B, self.num_actions), dtype=torch.float32, device=self.device)
        values = torch.empty((B,), dtype=torch.float32, device=self.device)
        gi = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._check(lib.nz_replay_gather(self._h, c_void_p(slots_d.data_ptr()), B, c_void_p(states.data_ptr()),
                                         c_void_p(policies.data_ptr()), c_void_p(values.data_ptr()),
                                         c_void_p(gi.data_ptr()), self._stream()))
        return Batch(states, policies, values, gi, keys, counts)

    def shuffle(self):
        self.index.shuffle()

    def get_slice(self, start_index, last_index, group_by_game=False):
        return self._gather(self.index.get_slice(start_index, last_index), group_by_game)

    def get_sample(self, batch_size, replace, probs, group_by_game=False):
        return self._gather(self.index.get_sample(batch_size, replace, probs), group_by_game)

    def get_buffer(self):
        """The whole buffer in the reference's list-of-tuples form (compatibility; copies everything to the host)."""
        return self._gather(self.index.seq).as_list()

    def len(self):
        return len(self.index)

    def played_games(self):
        return self.index.n_games

    # ---- checkpoints in the reference's format (ReplayBuffer.py:64-107) ---------------------------------------
    def save_to_file(self, file_path, step):
        self.step_to_size_map[step] = (self.len(), self.played_games())
        if self.index.full:
            self.allow_partial_loading = False
        torch.save({"buffer": self.get_buffer(), "map": self.step_to_size_map,
                    "partial_loading": self.allow_partial_loading}, file_path)

    def load_from_file(self, file_path, step):
        from .replay_buffer import load_reference_checkpoint
        buffer, n_games, self.step_to_size_map, self.allow_partial_loading = load_reference_checkpoint(file_path, step)
        self.index = ReplayIndex(self.window_size, self.capacity)
        n = len(buffer)
        if n == 0:
            self.index.n_games = n_games
            return
        states = torch.cat([e[0].reshape(1, -1).float() for e in buffer], 0).to(self.device).contiguous()
        policies = torch.tensor([e[1][1] for e in buffer]).to(self.device).contiguous()
        values = torch.tensor([int(e[1][0]) for e in buffer], dtype=torch.int32).to(self.device)
        gi = np.array([int(e[2]) for e in buffer], np.int32)
        self.index.seq = np.arange(n, dtype=np.int64)
        self.index.slot_game_index = gi.copy()
        self.index.n_games = n_games
        self.index.full = n_games >= self.window_size
        for g in np.unique(gi):                      # one append per game index (the index is a launch argument)
            dst = np.where(gi == g, np.arange(n, dtype=np.int64), -1)
            self._append(states, dst, values, 1, int(g), policies=policies)
