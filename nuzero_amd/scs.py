"""SCS rules on the GPU as batch operators (C ABI nz_scs_*), plus the game-config loader.

`ScsGameConfig` reads the reference's YAML game configs (Games/SCS/Game_configs/*.yml, parsed
by SCS_Game.load_game_from_config, SCS_Game.py:1570-1779) with the "Detailed" (or seeded "Randomized") map and
victory-point methods; `ScsBatch` runs G independent games on the device: step, legal-move
mask (possible_actions, :395-484), state image (generate_state, :1348-1505).
"""
import ctypes
from ctypes import byref, c_int32, c_void_p

import numpy as np
import torch
import yaml

from . import _lib
from ._lib import lib


def randomized_map(d, rows, cols, p1_last, p2_first, seed):
    """What SCS_Game.load_game_from_config draws for "Randomized" maps / victory points (SCS_Game.py:1678-1738) when
    numpy's global stream was seeded with `seed` just before: the sections in file order, a tile's terrain by
    np.random.choice(terrain_types, p=distribution) row by row, a victory point by np.random.choice(range(rows)) and
    np.random.choice(range(columns of the player's side)), redrawn while it repeats an earlier one.  Returns
    (terrain id map or None, (p1 points, p2 points) or None).  `seed` may be a RandomState: it is drawn from and left
    where the game's own draws go on."""
    rs = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    ids = [int(p["id"]) for p in d["Terrain"].values()]          # terrain_types: the Terrain section's order
    tmap = vps = None
    for section, values in d.items():
        if section == "Map" and values["creation_method"] == "Randomized":
            dist = values.get("distribution") or [1 / len(ids)] * len(ids)
            tmap = [[ids[int(rs.choice(len(ids), p=dist))] for _ in range(cols)] for _ in range(rows)]
        elif section == "Victory_points" and values["creation_method"] == "Randomized":
            vps = ([], [])
            col_ranges = (range(p1_last + 1), range(p2_first, cols))
            for side, key in enumerate(("p1", "p2")):
                for _ in range(int(values["number_vp"][key])):
                    pt = (int(rs.choice(range(rows))), int(rs.choice(col_ranges[side])))
                    while pt in vps[side]:
                        pt = (int(rs.choice(range(rows))), int(rs.choice(col_ranges[side])))
                    vps[side].append(pt)
    return tmap, vps


class ScsGameConfig:
    def __init__(self, path_or_dict, map_seed=None, per_game=False):
        """`map_seed`: for configs with "Randomized" maps or victory points -- the ONE map SCS_Game(config) builds when
        np.random.seed(map_seed) was called just before.  `per_game=True` (what the reference does: it draws from numpy's
        global stream whenever a game object is built, and Gamer builds one per game, Training/Gamer.py:52): every game
        gets its own map, drawn by `draw_game(rs)` from the game's own stream; this object then only holds a TEMPLATE
        map (the cheapest terrain everywhere: it bounds game lengths for the engine's limits)."""
        if isinstance(path_or_dict, dict):
            d = path_or_dict
        else:
            with open(path_or_dict) as f:
                d = yaml.safe_load(f)
        self.per_game = False
        self.rows, self.cols = int(d["Board_dimensions"]["rows"]), int(d["Board_dimensions"]["columns"])
        self.turns, self.stacking = int(d["Turns"]), int(d["Stacking_limit"])
        mid = self.cols // 2                    # the board's two sides (define_board_sides, :1140-1158)
        if self.cols % 2:
            side_p1_last, side_p2_first = mid - 1, mid + 1
        else:
            side_p1_last, side_p2_first = max(0, mid - 2), min(self.cols - 1, mid + 1)
        if d["Map"]["creation_method"] != "Detailed" or d["Victory_points"]["creation_method"] != "Detailed":
            if map_seed is None and not per_game:
                raise NotImplementedError("'Randomized' maps / victory points draw from numpy's global stream at load "
                                          "time (SCS_Game.py:1683-1738): pass map_seed (one map) or per_game=True")
            if per_game and map_seed is None:
                self.per_game = True
                self._random = (d, side_p1_last, side_p2_first)
                cheapest = min(d["Terrain"].values(), key=lambda t: t["cost"])["id"]
                rmap = [[cheapest] * self.cols for _ in range(self.rows)] if d["Map"]["creation_method"] != "Detailed" else None
                rvps = None
                if d["Victory_points"]["creation_method"] != "Detailed":
                    n1, n2 = (int(d["Victory_points"]["number_vp"][k]) for k in ("p1", "p2"))
                    rvps = ([(i % self.rows, 0) for i in range(n1)], [(i % self.rows, self.cols - 1) for i in range(n2)])
            else:
                rmap, rvps = randomized_map(d, self.rows, self.cols, side_p1_last, side_p2_first, map_seed)
            d = dict(d)
            if rmap is not None:
                d["Map"] = {"creation_method": "Detailed", "map_configuration": rmap}
            if rvps is not None:
                d["Victory_points"] = {"creation_method": "Detailed",
                                       "vp_locations": {"p1": [list(p) for p in rvps[0]], "p2": [list(p) for p in rvps[1]]}}
        units = {int(p["id"]): p for p in d["Units"].values()}
        terrain = {int(p["id"]): p for p in d["Terrain"].values()}
        self._terrain_by_id = terrain
        tmap = np.asarray(d["Map"]["map_configuration"])
        if tmap.shape != (self.rows, self.cols):
            raise ValueError("Wrong shape for map configuration")
        self.terrain = np.array([[terrain[int(t)]["attack_modifier"], terrain[int(t)]["defense_modifier"],
                                  terrain[int(t)]["cost"]] for t in tmap.reshape(-1)], np.float32)
        # the two halves of the board reinforcements may arrive on by default (define_board_sides, :1140-1158)
        mid = self.cols // 2
        if self.cols % 2:
            p1_last, p2_first = mid - 1, mid + 1
        else:
            p1_last, p2_first = max(0, mid - 2), min(self.cols - 1, mid + 1)
        arr = d["Reinforcements"]["arrival"]
        rows_u, rows_a = [], []
        for p, key in enumerate(("p1", "p2")):
            sched = d["Reinforcements"]["schedule"][key]
            if len(sched) != self.turns + 1:
                raise ValueError("Reinforcement schedule should have 'turns + 1' entries")
            k = 0
            for turn, ids in enumerate(sched):
                for uid in ids:
                    u = units[int(uid)]
                    a = np.zeros((self.rows, self.cols), np.uint8)
                    if arr["method"] == "Default":
                        if p == 0:
                            a[:, :p1_last + 1] = 1
                        else:
                            a[:, p2_first:] = 1
                    else:
                        for (r, c) in arr["locations"][key][k]:
                            a[r, c] = 1
                        k += 1
                    rows_u.append([p, turn, u["attack"], u["defense"], u["movement"]])
                    rows_a.append(a.reshape(-1))
        self.units = np.array(rows_u, np.int32)
        self.arrival = np.array(rows_a, np.uint8)
        vp = d["Victory_points"]["vp_locations"]
        self.n_vp = (len(vp["p1"]), len(vp["p2"]))
        self.vp = np.array(list(vp["p1"]) + list(vp["p2"]), np.int32).reshape(-1, 2)
        s = self.stacking
        self.planes = 1 + 6 * s + 1 + s + 1 + s + s
        self.num_actions = self.planes * self.rows * self.cols
        self.channels = 3 + 2 + 36 + 2 * 9 * s + 1 + s + 4 + 1 + 1

    def draw_game(self, rs):
        """One game's own map, drawn from the RandomState `rs` as SCS_Game(config) draws it from numpy's global stream
        (SCS_Game.py:1678-1738); `rs` is left where the game's play goes on.  Returns (terrain float32 [tiles, 3], victory
        points int32 [n_vp0 + n_vp1, 2]); parts the config gives in "Detailed" form are the config's."""
        assert self.per_game
        d, p1_last, p2_first = self._random
        rmap, rvps = randomized_map(d, self.rows, self.cols, p1_last, p2_first, rs)
        t = self.terrain if rmap is None else np.array(
            [[self._terrain_by_id[int(i)]["attack_modifier"], self._terrain_by_id[int(i)]["defense_modifier"],
              self._terrain_by_id[int(i)]["cost"]] for row in rmap for i in row], np.float32)
        v = self.vp if rvps is None else np.array(list(rvps[0]) + list(rvps[1]), np.int32).reshape(-1, 2)
        return t, v

    def draw_games(self, seeds):
        """`draw_game` for RandomState(seed) of every seed: (terrain [n, tiles, 3], vp [n, k, 2], the streams' states after
        the draws: MT19937 keys uint32 [n, 624] and positions int32 [n], the RandomStates themselves)."""
        n = len(seeds)
        terrain = np.empty((n, self.rows * self.cols, 3), np.float32)
        vp = np.empty((n, len(self.vp), 2), np.int32)
        keys, pos, streams = np.empty((n, 624), np.uint32), np.empty((n,), np.int32), []
        for i, s in enumerate(seeds):
            rs = np.random.RandomState(int(s))
            terrain[i], vp[i] = self.draw_game(rs)
            st = rs.get_state()
            assert st[3] == 0                      # (no cached Gaussian: choice() never draws one)
            keys[i], pos[i] = st[1], st[2]
            streams.append(rs)
        return terrain, vp, keys, pos, streams


class ScsBatch:
    def __init__(self, config, n_games, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        self.cfg = config if isinstance(config, ScsGameConfig) else ScsGameConfig(config)
        c = self.cfg
        self.device = torch.device("cuda", device)
        self.n_games = n_games
        self._keep = (np.ascontiguousarray(c.terrain), np.ascontiguousarray(c.vp), np.ascontiguousarray(c.units),
                      np.ascontiguousarray(c.arrival))
        d = _lib.ScsDesc(rows=c.rows, cols=c.cols, turns=c.turns, stacking=c.stacking,
                         terrain=self._keep[0].ctypes.data, n_vp=(c_int32 * 2)(*c.n_vp), vp=self._keep[1].ctypes.data,
                         n_units=len(c.units), units=self._keep[2].ctypes.data, arrival=self._keep[3].ctypes.data)
        self._h = c_void_p(0)
        st = lib.nz_scs_create(byref(self._h), byref(d), int(n_games), int(device))
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_scs_last_error(None) or b"").decode())

    def _check(self, st):
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_scs_last_error(self._h) or b"").decode())

    def _stream(self):
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    def close(self):
        if self._h.value:
            lib.nz_scs_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._check(lib.nz_scs_reset(self._h, self._stream()))

    def set_maps(self, terrain, vp):
        """Every game of the batch on its own map (float32 [G, tiles, 3], int32 [G, k, 2]; None: the config's); resets."""
        if terrain is None:
            self._check(lib.nz_scs_set_maps(self._h, None, None, self._stream()))
            return
        t = np.ascontiguousarray(terrain, np.float32)
        v = np.ascontiguousarray(vp, np.int32)
        assert t.shape == (self.n_games, self.cfg.rows * self.cfg.cols, 3) and v.shape == (self.n_games, len(self.cfg.vp), 2)
        self._check(lib.nz_scs_set_maps(self._h, c_void_p(t.ctypes.data), c_void_p(v.ctypes.data), self._stream()))

    def step(self, actions):
        a = torch.as_tensor(actions, dtype=torch.int32).to(self.device).contiguous()
        self._check(lib.nz_scs_step(self._h, c_void_p(a.data_ptr()), self._stream()))

    def legal_mask(self):
        c = self.cfg
        m = torch.empty((self.n_games, c.planes, c.rows, c.cols), dtype=torch.int8, device=self.device)
        self._check(lib.nz_scs_legal_mask(self._h, c_void_p(m.data_ptr()), self._stream()))
        return m

    def state_image(self):
        c = self.cfg
        x = torch.empty((self.n_games, c.channels, c.rows, c.cols), dtype=torch.float32, device=self.device)
        self._check(lib.nz_scs_state_image(self._h, c_void_p(x.data_ptr()), self._stream()))
        return x

    def status(self):
        """int32 [G, 7]: player, sub_phase, stage, turn, terminal, terminal_value, length."""
        s = torch.empty((self.n_games, 7), dtype=torch.int32, device=self.device)
        self._check(lib.nz_scs_status(self._h, c_void_p(s.data_ptr()), self._stream()))
        return s


class ScsSelfPlay:
    """MCTS self-play on SCS with the tree, rules and masks on the device and the leaf evaluations
    supplied by `evaluator(images) -> (probs [n, A] float32 post-softmax, values [n] float32)`
    (device tensors in, device or host tensors out) -- e.g. a PyTorch model followed by softmax.
    Lock-step: one host round trip per simulation wave (C ABI nz_scs_search_*)."""

    def __init__(self, config, search_config, n_games, training=True, device=0, nodes_per_game=None):
        from .search_config import to_struct
        if not torch.cuda.is_available():
            raise RuntimeError("nuzero_amd needs a ROCm GPU; there is no CPU fallback")
        self.cfg = config if isinstance(config, ScsGameConfig) else ScsGameConfig(config)
        c = self.cfg
        self.search_config, self.training = search_config, bool(training)
        self.device = torch.device("cuda", device)
        self.n_games = n_games
        sims = int(search_config["Simulation"]["mcts_simulations"])
        if nodes_per_game is None:
            nodes_per_game = 0          # the library's default: scaled with the simulations and the children bound
        self._keep = (np.ascontiguousarray(c.terrain), np.ascontiguousarray(c.vp), np.ascontiguousarray(c.units),
                      np.ascontiguousarray(c.arrival))
        d = _lib.ScsDesc(rows=c.rows, cols=c.cols, turns=c.turns, stacking=c.stacking,
                         terrain=self._keep[0].ctypes.data, n_vp=(c_int32 * 2)(*c.n_vp), vp=self._keep[1].ctypes.data,
                         n_units=len(c.units), units=self._keep[2].ctypes.data, arrival=self._keep[3].ctypes.data)
        sc = to_struct(search_config, training)
        self._h = c_void_p(0)
        st = lib.nz_scs_search_create(byref(self._h), byref(d), byref(sc), int(n_games), int(nodes_per_game),
                                      int(device))
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_scs_search_last_error(None) or b"").decode())
        G = n_games
        # decisions per game / children per node the records hold: bounded by the library from the game description
        mm, mc = c_int32(0), c_int32(0)
        self._check(lib.nz_scs_search_limits(self._h, byref(mm), byref(mc)))
        self.MAX_MOVES, self.MAX_CHILDREN = int(mm.value), int(mc.value)
        self._images = torch.empty((G, c.channels, c.rows, c.cols), dtype=torch.float32, device=self.device)
        self._leaf_game = torch.empty((G,), dtype=torch.int32, device=self.device)
        self.evaluations = 0

    def _check(self, st):
        if st != _lib.NZ_OK:
            raise _lib.NzError(st, (lib.nz_scs_search_last_error(self._h) or b"").decode())

    def _stream(self):
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    def close(self):
        if self._h.value:
            lib.nz_scs_search_destroy(self._h)
            self._h = c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self):
        s = torch.empty((self.n_games, 7), dtype=torch.int32, device=self.device)
        self._check(lib.nz_scs_search_status(self._h, c_void_p(s.data_ptr()), self._stream()))
        return s.cpu().numpy()

    def set_games(self, seeds):
        """Per-game maps (configs made with per_game=True): game i = `np.random.seed(seeds[i]); SCS_Game(config)`, its map
        drawn first and its play going on with the same stream (nz_scs_search_set_games).  Keeps the maps in
        `self.game_maps` = (terrain [n, tiles, 3], vp [n, k, 2]) for whoever replays the games.  Returns the streams."""
        terrain, vp, keys, pos, streams = self.cfg.draw_games(seeds)
        self._check(lib.nz_scs_search_set_games(self._h, len(seeds), c_void_p(terrain.ctypes.data), c_void_p(vp.ctypes.data),
                                                c_void_p(keys.ctypes.data), c_void_p(pos.ctypes.data)))
        self.game_maps = (terrain, vp)
        return streams

    def play(self, evaluator, seeds, max_moves=None):
        """Reset and play every game to the end (or for `max_moves` decisions); game g draws from
        RandomState(seeds[g]) in the reference's order (gamma x n_root_children, two uniforms, at most one
        more for choice).  `evaluator` may take a second argument: the game index of every leaf."""
        ex = self.search_config["Exploration"]
        G, A = self.n_games, self.cfg.num_actions
        if self.cfg.per_game:
            rngs = self.set_games(list(seeds))         # (resets; every stream stands behind its map's draws)
        else:
            rngs = [np.random.RandomState(int(s)) for s in seeds]
            self._check(lib.nz_scs_search_reset(self._h, self._stream()))
        nchild_dev = torch.empty((G,), dtype=torch.int32, device=self.device)
        import inspect
        wants_games = len(inspect.signature(evaluator).parameters) >= 2
        for _ in range(min(self.MAX_MOVES, max_moves or self.MAX_MOVES)):
            st = self.status()
            alive = st[:, 4] == 0
            if not alive.any():
                break
            noise = np.zeros((G, self.MAX_CHILDREN), np.float64)
            uni = np.zeros((G, 3), np.float64)
            if self.training:
                self._check(lib.nz_scs_search_root_children(self._h, c_void_p(nchild_dev.data_ptr()), self._stream()))
                nchild = nchild_dev.cpu().numpy()
                for g in np.nonzero(alive)[0]:
                    rs, n = rngs[g], int(nchild[g])
                    noise[g, :n] = rs.gamma(ex["root_dist_alpha"], ex["root_dist_beta"], n)
                    if st[g, 6] < ex["number_of_softmax_moves"]:
                        uni[g, 2] = rs.random_sample()
                    else:
                        u1, u2 = rs.random_sample(), rs.random_sample()
                        uni[g, 0], uni[g, 1] = u1, u2
                        if u1 < ex["epsilon_softmax_exploration"] or u2 < ex["epsilon_random_exploration"]:
                            uni[g, 2] = rs.random_sample()
            noise_d = torch.from_numpy(noise).to(self.device)
            uni_d = torch.from_numpy(uni).to(self.device)
            self._check(lib.nz_scs_search_begin_move(self._h, c_void_p(noise_d.data_ptr()), self._stream()))
            while True:
                n = c_int32(0)
                self._check(lib.nz_scs_search_select(self._h, c_void_p(self._images.data_ptr()),
                                                     c_void_p(self._leaf_game.data_ptr()), byref(n), self._stream()))
                if n.value == 0:
                    break
                if wants_games:
                    probs, values = evaluator(self._images[:n.value], self._leaf_game[:n.value])
                else:
                    probs, values = evaluator(self._images[:n.value])
                probs = torch.as_tensor(probs, dtype=torch.float32).to(self.device).contiguous()
                values = torch.as_tensor(values, dtype=torch.float32).to(self.device).contiguous()
                assert probs.shape == (n.value, A) and values.shape == (n.value,)
                self.evaluations += n.value
                self._check(lib.nz_scs_search_expand(self._h, c_void_p(probs.data_ptr()), c_void_p(values.data_ptr()),
                                                     self._stream()))
            self._check(lib.nz_scs_search_end_move(self._h, c_void_p(uni_d.data_ptr()), self._stream()))
        return self.export()

    # ---- one move at a time: evaluation matches (MctsAgent.choose_action / update_subtree, Tester.Test_using_agents) ----
    def reset(self):
        self._check(lib.nz_scs_search_reset(self._h, self._stream()))

    def search(self, evaluator, noise=None):
        """Root noise (training engines; float64 [G, MAX_CHILDREN]) and the move's simulations for every live game;
        no action yet."""
        noise_d = torch.as_tensor(noise, dtype=torch.float64).to(self.device).contiguous() if noise is not None else None
        self._check(lib.nz_scs_search_begin_move(self._h, c_void_p(noise_d.data_ptr()) if noise_d is not None else None,
                                                 self._stream()))
        while True:
            n = c_int32(0)
            self._check(lib.nz_scs_search_select(self._h, c_void_p(self._images.data_ptr()),
                                                 c_void_p(self._leaf_game.data_ptr()), byref(n), self._stream()))
            if n.value == 0:
                return
            probs, values = evaluator(self._images[:n.value])
            probs = torch.as_tensor(probs, dtype=torch.float32).to(self.device).contiguous()
            values = torch.as_tensor(values, dtype=torch.float32).to(self.device).contiguous()
            self.evaluations += n.value
            self._check(lib.nz_scs_search_expand(self._h, c_void_p(probs.data_ptr()), c_void_p(values.data_ptr()),
                                                 self._stream()))

    def apply(self, actions=None, uniforms=None):
        """Play `actions` (int32 [G]; None / -1 = the search's own choice), step and re-root."""
        a = torch.as_tensor(actions, dtype=torch.int32).to(self.device).contiguous() if actions is not None else None
        u = torch.as_tensor(uniforms, dtype=torch.float64).to(self.device).contiguous() if uniforms is not None else None
        self._check(lib.nz_scs_search_apply(self._h, c_void_p(a.data_ptr()) if a is not None else None,
                                            c_void_p(u.data_ptr()) if u is not None else None, self._stream()))

    def last_actions(self):
        out = torch.empty((self.n_games,), dtype=torch.int32, device=self.device)
        self._check(lib.nz_scs_search_last_actions(self._h, c_void_p(out.data_ptr()), self._stream()))
        return out

    def play_native(self, net, seeds, max_moves=None):
        """As play(), with the network on the device as well (`net`: nuzero_amd.boardnet.BoardNet with
        max_batch >= n_games): the whole move loop runs in the library (nz_scs_search_play), the
        host only draws the per-move random numbers.  Same games as play(net.evaluator(), seeds)."""
        seeds = np.ascontiguousarray(np.asarray(list(seeds), dtype=np.uint32))
        assert seeds.shape == (self.n_games,)
        if self.cfg.per_game:
            self.set_games(seeds.tolist())
        self._check(lib.nz_scs_search_play_moves(self._h, net._h, c_void_p(seeds.ctypes.data), int(max_moves or 0),
                                                 self._stream()))
        out = self.export()
        self.evaluations = out["expansions"]
        waves = ctypes.c_int64(0)
        self._check(lib.nz_scs_search_waves(self._h, byref(waves)))
        out["waves"] = int(waves.value)
        return out

    def persistent(self, enable=-1):
        """The library's move loop on the persistent kernel (one wavefront per game, a move's whole search in one
        launch): 1 require it, 0 never, -1 the default (where the network has a per-wavefront form and the inference
        cache is off).  Returns whether the LAST play ran on it."""
        used = ctypes.c_int32(0)
        self._check(lib.nz_scs_search_persistent(self._h, int(enable), byref(used)))
        return bool(used.value)

    def persist_profile(self, enable=-1):
        """HIP-event time of the persistent kernel (nz_scs_search_persist_profile): enable 1 / 0 switches, -1 reads."""
        out = (ctypes.c_double * 4)()
        self._check(lib.nz_scs_search_persist_profile(self._h, int(enable), out))
        return {"ms": out[0], "launches": int(out[1]), "mfma_per_position": int(out[2]), "flops_per_position": int(out[3])}

    def persist_ticks(self):
        """Diagnostic build (-DNZ_PERSIST_STAMPS) only: shader ticks per phase of the persistent kernel."""
        out = (ctypes.c_int64 * 13)()
        self._check(lib.nz_scs_search_persist_ticks(self._h, out))
        keys = ("clone", "descent", "mask_list", "planes_split", "network", "softmax_value", "expansion", "backup",
                "moves_total", "slowest_game_move", "net_k_loops", "net_epilogues", "net_pair_waits")
        return dict(zip(keys, (int(v) for v in out)))

    def record(self, games, capacity):
        """Test hook of the persistent route: keep the leaf evaluations of `games` (indices), up to `capacity` each,
        in the order each game's search consumes them (nz_scs_search_record); games = [] stops recording."""
        games = np.ascontiguousarray(np.asarray(list(games), dtype=np.int32))
        self._recorded = [int(g) for g in games]
        self._check(lib.nz_scs_search_record(self._h, c_void_p(games.ctypes.data), len(games), int(capacity)))
        self._record_capacity = int(capacity)

    def records(self):
        """{game: (digests uint64 [n, 2], probs float32 [n, A], values float32 [n])} of the recorded games."""
        out, A = {}, self.cfg.planes * self.cfg.rows * self.cfg.cols
        for slot, g in enumerate(self._recorded):
            count = ctypes.c_int32(0)
            self._check(lib.nz_scs_search_record_read(self._h, slot, byref(count), None, None, None))
            if count.value > self._record_capacity:
                raise RuntimeError(f"game {g} consumed {count.value} evaluations, capacity {self._record_capacity}")
            n = count.value
            dig, pr, va = np.empty((n, 2), np.uint64), np.empty((n, A), np.float32), np.empty((n,), np.float32)
            self._check(lib.nz_scs_search_record_read(self._h, slot, byref(count), c_void_p(dig.ctypes.data),
                                                      c_void_p(pr.ctypes.data), c_void_p(va.ctypes.data)))
            out[g] = (dig, pr, va)
        return out

    def play_round(self, net, seeds):
        """A round of len(seeds) >= n_games games over this engine's n_games slots (nz_scs_search_play_round): a slot
        whose game has ended starts the round's next game, as a Gamer actor plays its games back to back
        (Training/Gamer.py:45-98).  Game i is seeded with seeds[i] whichever slot plays it -- the same games as
        play_native on an engine of len(seeds) slots.  Returns export()'s dictionary with len(seeds) rows."""
        return {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in self.play_round_device(net, seeds).items()}

    def play_round_device(self, net, seeds):
        """play_round with the records left on the device (export_device()'s dictionary, len(seeds) rows)."""
        seeds = np.ascontiguousarray(np.asarray(list(seeds), dtype=np.uint32))
        n = int(seeds.shape[0])
        waves = ctypes.c_int64(0)
        if self.cfg.per_game:
            self.set_games(seeds.tolist())
        if n == self.n_games:
            self._check(lib.nz_scs_search_play_moves(self._h, net._h, c_void_p(seeds.ctypes.data), 0, self._stream()))
            t = self.export_device()
        else:
            self._check(lib.nz_scs_search_play_round(self._h, net._h, c_void_p(seeds.ctypes.data), n, self._stream()))
            M, C, dev = self.MAX_MOVES, self.MAX_CHILDREN, self.device
            t = {"actions": torch.empty((n, M), dtype=torch.int32, device=dev),
                 "tree_size": torch.empty((n, M), dtype=torch.int32, device=dev),
                 "n_children": torch.empty((n, M), dtype=torch.int32, device=dev),
                 "bias": torch.empty((n, M), dtype=torch.float64, device=dev),
                 "root_value_sum": torch.empty((n, M), dtype=torch.float64, device=dev),
                 "child_action": torch.empty((n, M, C), dtype=torch.int32, device=dev),
                 "child_visit": torch.empty((n, M, C), dtype=torch.int32, device=dev),
                 "child_prior": torch.empty((n, M, C), dtype=torch.float64, device=dev),
                 "child_value_sum": torch.empty((n, M, C), dtype=torch.float64, device=dev)}
            st = torch.empty((n, 2), dtype=torch.int32, device=dev)
            counters = (ctypes.c_int64 * 2)()
            self._check(lib.nz_scs_search_export_round(self._h, *[c_void_p(t[k].data_ptr()) for k in (
                "actions", "tree_size", "n_children", "bias", "root_value_sum", "child_action", "child_visit",
                "child_prior", "child_value_sum")], c_void_p(st.data_ptr()), counters, self._stream()))
            t["lengths"], t["outcomes"] = st[:, 0].contiguous(), st[:, 1].contiguous()
            t["simulations"], t["expansions"] = int(counters[0]), int(counters[1])
        self.evaluations = t["expansions"]
        self._check(lib.nz_scs_search_waves(self._h, byref(waves)))
        t["waves"] = int(waves.value)
        return t

    def cache(self, max_entries):
        """The reference's inference cache for play_native (KeylessCache(max_size), Utils/Caches/KeylessCache.py:24-160):
        a device hash table of the largest power of two <= max_entries, shared by the engine's games; 0 switches it
        off.  Results do not change (tests/test_gpu_scs.py)."""
        self._check(lib.nz_scs_search_cache(self._h, int(max_entries)))

    def cache_clear(self):
        """Empty the table (the reference builds new caches for every self-play round, AlphaZero.py:525-577)."""
        self._check(lib.nz_scs_search_cache(self._h, -1))

    def cache_stats(self):
        out = (ctypes.c_int64 * 4)()
        self._check(lib.nz_scs_search_cache_stats(self._h, out))
        return {"hits": int(out[0]), "misses": int(out[1]), "entries": int(out[2]), "size": int(out[3])}

    def phase_ticks(self):
        """Diagnostic build only (-DNZ_SCS_STAMPS): shader ticks per phase of the wave kernel, summed over games."""
        out = (ctypes.c_int64 * 6)()
        self._check(lib.nz_scs_search_phase_ticks(self._h, out))
        return dict(zip(("expand", "rules_copy", "clone", "descent", "leaf_mask_image", "terminal_sims"), [int(v) for v in out]))

    def export_device(self):
        """The finished round as device tensors (what the device replay buffer and the multi-GPU gather read)."""
        G, M, C = self.n_games, self.MAX_MOVES, self.MAX_CHILDREN
        dev = self.device
        t = {"actions": torch.empty((G, M), dtype=torch.int32, device=dev),
             "tree_size": torch.empty((G, M), dtype=torch.int32, device=dev),
             "n_children": torch.empty((G, M), dtype=torch.int32, device=dev),
             "bias": torch.empty((G, M), dtype=torch.float64, device=dev),
             "root_value_sum": torch.empty((G, M), dtype=torch.float64, device=dev),
             "child_action": torch.empty((G, M, C), dtype=torch.int32, device=dev),
             "child_visit": torch.empty((G, M, C), dtype=torch.int32, device=dev),
             "child_prior": torch.empty((G, M, C), dtype=torch.float64, device=dev),
             "child_value_sum": torch.empty((G, M, C), dtype=torch.float64, device=dev)}
        counters = (ctypes.c_int64 * 2)()
        self._check(lib.nz_scs_search_export(self._h, *[c_void_p(t[k].data_ptr()) for k in (
            "actions", "tree_size", "n_children", "bias", "root_value_sum", "child_action", "child_visit",
            "child_prior", "child_value_sum")], counters, self._stream()))
        st = torch.empty((G, 7), dtype=torch.int32, device=dev)
        self._check(lib.nz_scs_search_status(self._h, c_void_p(st.data_ptr()), self._stream()))
        t["lengths"], t["outcomes"] = st[:, 6].contiguous(), st[:, 5].contiguous()
        t["simulations"], t["expansions"] = int(counters[0]), int(counters[1])
        return t

    def export(self):
        return {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in self.export_device().items()}


def torch_evaluator(model, recurrent_iterations=2, pad_to=None):
    """Evaluator for ScsSelfPlay from a PyTorch policy/value module with the reference's calling
    convention (Network_Manager.inference, Neural_Networks/Network_Manager.py:46-64): recurrent
    models are called as model(x, iters) -> ((policy, value), thought), others as model(x) ->
    (policy, value).  Softmax over ALL logits (Explorer.py:159) is applied here, on the GPU.
    `pad_to` pads every leaf batch to one fixed size, so that MIOpen tunes its convolutions once
    instead of once per batch size (the number of leaves changes from wave to wave)."""
    model.eval()
    recurrent = bool(getattr(model, "recurrent", False))

    def ev(images):
        n = images.shape[0]
        if pad_to is not None and n < pad_to:
            images = torch.cat([images, images.new_zeros((pad_to - n,) + tuple(images.shape[1:]))], 0)
        with torch.no_grad():
            if recurrent:
                (p, v), _ = model(images, recurrent_iterations)
            else:
                p, v = model(images)
        probs = torch.softmax(p.reshape(p.shape[0], -1).float(), dim=1)
        return probs[:n].contiguous(), v.reshape(-1).float()[:n].contiguous()
    return ev


class ScsGameRecord:
    """A finished SCS game as ReplayBuffer.save_game and the trainer read it
    (Training/ReplayBuffer.py:31-33, SCS_Game.py:1517-1528)."""

    def __init__(self, states, child_actions, child_visits, n_children, length, terminal_value, num_actions, actions=None):
        self.length, self.terminal_value = int(length), int(terminal_value)
        self.action_history = [int(a) for a in actions[:self.length]] if actions is not None else None
        self.state_history = [torch.from_numpy(np.ascontiguousarray(states[m:m + 1])) for m in range(self.length)]
        self.child_policy = []
        for m in range(self.length):
            k = int(n_children[m])
            visits = [int(v) for v in child_visits[m, :k]]
            total = sum(visits)
            pol = [0] * num_actions
            for a, v in zip(child_actions[m, :k], visits):
                pol[int(a)] = v / total
            self.child_policy.append(pol)

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        return (self.terminal_value, self.child_policy[i])


def scs_game_records(selfplay, result):
    """GameRecords of a finished ScsSelfPlay round.  The per-move state images are regenerated by
    replaying the recorded actions through the device rules (ScsBatch), one image per decision."""
    lengths = result["lengths"]
    cfg, G = selfplay.cfg, int(lengths.shape[0])        # rows of the round (>= selfplay.n_games after play_round)
    batch = ScsBatch(cfg, G, device=selfplay.device.index or 0)
    if cfg.per_game:
        batch.set_maps(selfplay.game_maps[0][:G], selfplay.game_maps[1][:G])
    L = int(lengths.max())
    states = np.zeros((G, L, cfg.channels, cfg.rows, cfg.cols), np.float32)
    for m in range(L):
        states[:, m] = batch.state_image().cpu().numpy()
        batch.step(np.where(m < lengths, result["actions"][:, m], -1).astype(np.int32))
    batch.close()
    return [ScsGameRecord(states[g], result["child_action"][g], result["child_visit"][g], result["n_children"][g],
                          lengths[g], result["outcomes"][g], cfg.num_actions, result["actions"][g]) for g in range(G)]
