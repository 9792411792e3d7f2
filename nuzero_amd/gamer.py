"""Gamer with the reference's surface (Training/Gamer.py:17-104), without Ray.

One reference Gamer plays one game per `play_game()` call on one CPU process;
here one Gamer drives a GPU engine that plays `num_games` games per call on
`concurrent_games` trees.  What comes back is what the reference's loop hands to
the trainer and the replay buffer:

* per game a record with ``state_history`` / ``get_state_from_history`` /
  ``make_target`` / ``length`` -- the attributes ReplayBuffer.save_game reads
  (Training/ReplayBuffer.py:31-33) -- and the six statistics of Gamer.py:42-50;
* states are CPU float32 ``[1,C,H,W]`` tensors, policies flat lists of floats,
  values numbers, as Training/AlphaZero.py:851-852,892-903 requires.
"""
import numpy as np
import torch

from .engine import SelfPlayEngine
from .network import Network_Manager


class GameRecord:
    """The finished game as ReplayBuffer.save_game and the trainer see it."""

    def __init__(self, states, visits, actions, length, terminal_value):
        self.length = int(length)
        self.terminal_value = int(terminal_value)
        self.state_history = [torch.from_numpy(np.ascontiguousarray(states[m:m + 1])) for m in range(self.length)]
        # tic_tac_toe.py:177-182: visit / sum(visits) for the root's children, 0 elsewhere
        self.child_policy = []
        for m in range(self.length):
            v = [int(x) for x in visits[m]]
            total = sum(v)
            self.child_policy.append([x / total if x else 0 for x in v])
        self.action_history = [int(a) for a in actions[:self.length]]

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        return (self.terminal_value, self.child_policy[i])

    def get_length(self):
        return self.length

    def get_terminal_value(self):
        return self.terminal_value


class DeviceCacheStats:
    """What AlphaZero.run_selfplay reads of the cache a Gamer hands back (AlphaZero.py:553-568), for the device table of
    nz_scs_search_cache: hit ratio, entries, fill ratio.  The table itself stays on the device and is shared by all games
    of the round already, so `update` (merging another game's cache, KeylessCache.py:89-104) has nothing left to do."""

    def __init__(self, stats, update_threshold=0.8):
        self.stats, self.update_threshold = stats, update_threshold

    def get_hit_ratio(self):
        n = self.stats["hits"] + self.stats["misses"]
        return self.stats["hits"] / n if n else 0.0

    def length(self):
        return self.stats["entries"]

    def get_fill_ratio(self):
        return self.stats["entries"] / self.stats["size"] if self.stats["size"] else 0.0

    def get_update_threshold(self):
        return self.update_threshold

    def update(self, other):
        return None


class DisabledCache:
    """Cache-shaped object for callers that expect one back from play_game
    (AlphaZero.py:553-568).  hit_ratio 0: the engine evaluates every leaf; 1: every leaf is
    read from the all-positions table (SelfPlayEngine.cache_all_positions)."""

    def __init__(self, hit_ratio=0.0):
        self.hit_ratio = hit_ratio

    def get_hit_ratio(self):
        return self.hit_ratio

    def length(self):
        return 0

    def get_fill_ratio(self):
        return 0.0

    def get_update_threshold(self):
        return 1.0

    def update(self, other):
        return None


def round_stats(r):
    """game_stats for every game of an export at once (numpy): a list of the reference's statistics dicts."""
    lengths = np.asarray(r["lengths"], np.int64)
    G, T = r["tree_size"].shape
    live = np.arange(T)[None, :] < lengths[:, None]
    ts = np.where(live, r["tree_size"], 0).astype(np.int64)
    ch = np.where(live, r["n_children"], 0).astype(np.int64)
    bias = np.where(live, r["bias"], 0.0)
    acc = np.add.accumulate(bias, axis=1)                       # left to right: the reference's += in move order
    last = np.maximum(lengths - 1, 0)
    rows = np.arange(G)
    n = np.maximum(lengths, 1)
    cols = (lengths.tolist(), (ch.sum(1) / n).tolist(), (ts.sum(1) / n).tolist(), ts[rows, last].tolist(),
            (acc[rows, last] / n).tolist(), r["bias"][rows, last].tolist())
    keys = ("number_of_moves", "average_children", "average_tree_size", "final_tree_size", "average_bias_value",
            "final_bias_value")
    return [dict(zip(keys, row)) for row in zip(*cols)]


def game_stats(r, g):
    """The statistics dict of Gamer.py:42-50,81-92 for game g of an export."""
    n = int(r["lengths"][g])
    ts = r["tree_size"][g, :n].astype(np.int64)
    ch = r["n_children"][g, :n].astype(np.int64)
    bias = r["bias"][g, :n]
    acc = 0.0
    for b in bias:                       # the reference accumulates in move order
        acc += float(b)
    return {"number_of_moves": n,
            "average_children": int(ch.sum()) / n,
            "average_tree_size": int(ts.sum()) / n,
            "final_tree_size": int(ts[-1]),
            "average_bias_value": acc / n,
            "final_bias_value": float(bias[-1])}


class Gamer:
    def __init__(self, buffer, shared_storage, game_class, game_args, game_index, search_config,
                 recurrent_iterations, cache_choice="disabled", size_estimate=10000,
                 num_games=1, concurrent_games=None, device=0, base_seed=0, records=True):
        """Arguments up to `size_estimate` are the reference's (Gamer.py:20).
        `shared_storage` is anything with ``get()`` returning a Network_Manager (or
        the Network_Manager itself); `buffer` anything with ``save_game(game, i)``
        or None.  `game_class` is Tic_Tac_Toe or SCS_Game (recognised by name); for SCS `game_args[0]` is
        the game config file and all `num_games` games of a round run concurrently."""
        name = getattr(game_class, "__name__", str(game_class)).lower()
        self.is_scs = "scs" in name
        if not self.is_scs and "tic" not in name and "ttt" not in name:
            raise NotImplementedError(f"game {name!r}: Tic_Tac_Toe and SCS_Game run on the GPU engine")
        if cache_choice not in ("disabled", None, "dict", "keyless"):
            raise ValueError(f"bad cache_choice {cache_choice!r}")          # general_utils.py:14-24
        self.cache_choice = cache_choice if cache_choice else "disabled"
        self.buffer, self.shared_storage = buffer, shared_storage
        self.game_index = game_index
        self.search_config = search_config
        self.recurrent_iterations = recurrent_iterations
        self.num_games = num_games
        self.base_seed = base_seed
        self.time_to_stop = False
        self.device = device
        # records=False: play_games returns no per-game record objects (a round of 16 k games is 115 k Python lists);
        # with a device replay buffer (`buffer.save_games_from_engine`) the positions then never visit the host
        self.records = records
        if not records and buffer is not None and not hasattr(buffer, "save_scs_games" if self.is_scs else "save_games_from_engine"):
            # a host ReplayBuffer is filled from the record objects (buffer.save_game): without them every game of every
            # round would be dropped and training would run on an empty buffer
            raise ValueError("records=False needs a replay buffer that takes the games on the device "
                             "(nuzero_amd.replay_device.DeviceReplayBuffer) or buffer=None")
        self._loaded = None
        self._wrapped = None
        if self.is_scs:
            # game_args = [config path] as for SCS_Game(cfg) (Games/SCS/SCS_Game.py:45); every game of the round runs
            # concurrently
            from .scs import ScsGameConfig, ScsSelfPlay
            if not game_args:
                raise ValueError("SCS needs game_args = [path of the game config]")
            # A "Randomized" config gives every game its own map, as the reference's Gamer builds a game object -- and with
            # it a map -- per game (Gamer.py:52; SCS_Game.py:1678-1738): game i of a round is
            # `np.random.seed(base_seed + i); SCS_Game(config); play` -- one stream, the map's draws first.
            # [config path, map_seed] instead plays every game on the ONE map drawn after np.random.seed(map_seed).
            self.scs_config = ScsGameConfig(game_args[0], *game_args[1:2], per_game=len(game_args) < 2)
            # `concurrent_games` trees play the round's `num_games` games: a tree whose game has ended starts the round's
            # next one (the reference's ActorPool of num_actors Gamers over num_games_per_step games, AlphaZero.py:525-577;
            # games are independent and seeded by their index, so which tree plays a game does not change it)
            self.concurrent = min(int(concurrent_games or num_games), num_games)
            self.engine = ScsSelfPlay(self.scs_config, search_config, self.concurrent, training=True, device=device)
            if self.cache_choice != "disabled":
                # "keyless" and "dict" (general_utils.py:14-24) both become the device table: a 128-bit hash is stored
                # instead of the key (KeylessCache), sized like KeylessCache(size_estimate)
                self.engine.cache(size_estimate)
            self._board_net = None
            return
        self.engine = SelfPlayEngine(search_config, num_games, training=True, device=device,
                                     n_slots=concurrent_games or num_games)

    def _network(self):
        """shared_storage.get() (Gamer.py:40,61).  A raw module / state dict is wrapped once and the wrapper kept, so
        that its weights are only re-uploaded when they changed (Network_Manager.sync)."""
        nm = self.shared_storage.get() if hasattr(self.shared_storage, "get") else self.shared_storage
        if not isinstance(nm, Network_Manager):
            if self._wrapped is None or self._wrapped[0] is not nm:
                self._wrapped = (nm, Network_Manager(nm))
            nm = self._wrapped[1]
        return nm

    def play_games(self):
        """One self-play round: `num_games` games.  Returns (records, stats list)."""
        if self.is_scs:
            return self._play_scs_games()
        self._loaded = self._load_weights(self.engine, self._loaded)
        self.engine.play(base_seed=self.base_seed, next_base_seed=self.base_seed + self.num_games)
        self.base_seed += self.num_games
        return self._consume_round(self.engine)

    def _load_weights(self, engine, loaded):
        """The engine's weights follow the network in shared storage (Gamer.py:40,61); `loaded`: what it holds now."""
        nm = self._network() if self.shared_storage is not None else None
        if nm is None or loaded == (id(nm), nm.sync()):
            return loaded
        s = nm.spec()
        engine.set_weights(nm.state_dict(), width=s.width, num_blocks=s.num_blocks, recall=s.recall,
                           value_activation=s.value_activation, recurrent_iterations=self.recurrent_iterations,
                           arch=s.arch, kernel_size=s.kernel_size)
        if self.cache_choice != "disabled":
            engine.cache_all_positions()
        return (id(nm), nm.version)

    def _consume_round(self, engine):
        """A finished Tic-Tac-Toe round of `engine`: games to the replay buffer, records and statistics out."""
        on_device = hasattr(self.buffer, "save_games_from_engine")
        if on_device:
            # replay buffer in HBM: states, visit counts and outcomes go from the engine's export buffers into the
            # buffer's slots on the device (ReplayBuffer.save_game for every game of the round, in order)
            ex = engine.export_device()
            self.buffer.save_games_from_engine(engine, self.game_index, export=ex)
            if not self.records:
                r = {k: ex[k].cpu().numpy() for k in ("lengths", "tree_size", "n_children", "bias")}
                return [], round_stats(r)
            r = {k: (v.cpu().numpy() if v is not None else None) for k, v in ex.items()}
        else:
            r = engine.export(states=self.records)
        stats = round_stats(r)
        if not self.records:
            return [], stats
        records = [GameRecord(r["states"][g], r["visits"][g], r["actions"][g], r["lengths"][g], r["outcomes"][g])
                   for g in range(self.num_games)]
        if self.buffer is not None and not on_device:
            for rec in records:
                self.buffer.save_game(rec, self.game_index)
        return records, stats

    def _play_scs_games(self):
        from .scs import scs_game_records
        nm = self._network()
        if self._loaded != (id(nm), nm.sync()):
            if self._board_net is not None:
                self._board_net.close()
            c = self.scs_config
            self._board_net = nm.board_net(c.rows, c.cols, self.concurrent, self.recurrent_iterations, self.device)
            self._loaded = (id(nm), nm.version)
        if self.cache_choice != "disabled":
            self.engine.cache_clear()                  # a new round: new caches (AlphaZero.py:525-537)
        on_device = hasattr(self.buffer, "save_scs_games")
        records, stats = [], []
        # one round over the engine's `concurrent` slots: a slot whose game has ended starts the round's next game
        # (nz_scs_search_play_round), as the reference's actors play their games back to back (Gamer.py:45-98)
        seeds = [self.base_seed + i for i in range(self.num_games)]
        t = self.engine.play_round_device(self._board_net, seeds)
        r = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in t.items()}
        stats = round_stats({k: r[k] for k in ("lengths", "tree_size", "n_children", "bias")})
        if on_device:
            self.buffer.save_scs_games(self.engine, t, self.game_index)
        if self.records:
            records = scs_game_records(self.engine, r)
            if self.buffer is not None and not on_device:
                for rec in records:
                    self.buffer.save_game(rec, self.game_index)
        self.base_seed += self.num_games
        return records, stats

    def play_game(self, cache=None):
        """Reference signature (Gamer.py:39-97): (stats, cache).  With num_games > 1
        the stats are those of the round's first game; use play_games for all."""
        _, stats = self.play_games()
        if self.is_scs and self.cache_choice != "disabled":
            return stats[0], DeviceCacheStats(self.engine.cache_stats(), 0.8 if self.cache_choice == "keyless" else 0.7)
        return stats[0], DisabledCache(0.0 if self.cache_choice == "disabled" else 1.0)

    def play_forever(self, rounds_in_flight=2, on_round=None):
        """Gamer.play_forever (Gamer.py:99-101; the trainer's asynchronous mode, AlphaZero.py:404,496): rounds until
        stop().  Tic-Tac-Toe rounds run `rounds_in_flight` at a time (RoundPipeline: the next round's workgroups take the
        compute units the current round's tail leaves idle); every round is the one play_games() would play next, the
        weights are looked up before each round starts, and finished rounds reach the replay buffer in order.
        `on_round(records, stats)`: called after each round (tests, progress)."""
        if self.is_scs or rounds_in_flight < 2:
            while not self.time_to_stop:
                out = self.play_games()
                if on_round is not None:
                    on_round(*out)
            return
        from .engine import RoundPipeline
        first, made, loaded = self.engine, [], {}

        def make():
            if not made:
                made.append(first)
                return first
            e = SelfPlayEngine(self.search_config, self.num_games, training=True, device=self.device, n_slots=first.n_slots)
            made.append(e)
            return e

        pipe = RoundPipeline(make, depth=rounds_in_flight)
        loaded[id(first)] = self._loaded
        try:
            while not self.time_to_stop or pipe.pending:
                while len(pipe.pending) == rounds_in_flight or (self.time_to_stop and pipe.pending):
                    _, eng = pipe.collect()
                    out = self._consume_round(eng)
                    if on_round is not None:
                        on_round(*out)
                if self.time_to_stop:
                    break
                eng = pipe.engines[pipe.submitted % rounds_in_flight]
                loaded[id(eng)] = self._load_weights(eng, loaded.get(id(eng)))
                pipe.submit(self.base_seed, next_base_seed=self.base_seed + rounds_in_flight * self.num_games)
                self.base_seed += self.num_games
        finally:
            self._loaded = loaded.get(id(first))
            while pipe.pending:
                pipe.collect()
            pipe.pool.shutdown()
            for e in made[1:]:
                e.close()

    def stop(self):
        self.time_to_stop = True
