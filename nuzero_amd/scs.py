"""SCS rules on the GPU as batch operators (C ABI nz_scs_*), plus the game-config loader.

`ScsGameConfig` reads the reference's YAML game configs (Games/SCS/Game_configs/*.yml, parsed
by SCS_Game.load_game_from_config, SCS_Game.py:1570-1779) with the "Detailed" map and
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


class ScsGameConfig:
    def __init__(self, path_or_dict):
        if isinstance(path_or_dict, dict):
            d = path_or_dict
        else:
            with open(path_or_dict) as f:
                d = yaml.safe_load(f)
        self.rows, self.cols = int(d["Board_dimensions"]["rows"]), int(d["Board_dimensions"]["columns"])
        self.turns, self.stacking = int(d["Turns"]), int(d["Stacking_limit"])
        if d["Map"]["creation_method"] != "Detailed" or d["Victory_points"]["creation_method"] != "Detailed":
            raise NotImplementedError("only 'Detailed' maps and victory points (the 'Randomized' methods draw "
                                      "from numpy's global stream at load time, SCS_Game.py:1683-1738)")
        units = {int(p["id"]): p for p in d["Units"].values()}
        terrain = {int(p["id"]): p for p in d["Terrain"].values()}
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
