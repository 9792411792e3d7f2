"""Oracle: SCS (hex war-game) rules, legal-move mask and state image (test infrastructure only).

Restates /root/reference/Games/SCS/SCS_Game.py (with Unit.py, Tile.py, Terrain.py): the
10-stage turn machine (:687-831), possible_actions (:395-484), parse/play_action (:486-633),
end_movement / end_fighting (:927-946), resolve_combat (:997-1044, :1253-1285), the hex
neighbourhood with column-parity offsets (:1048-1094, :1199-1243), check_termination (:857-894)
and generate_state (:1348-1505).  Pinned by tests/golden/scs_kat.npz, which
tests/golden/make_golden_scs.py made by importing the genuine reference.

Only "Detailed" maps / victory points are supported (the "Randomized" creation methods draw from
the global numpy stream at load time, SCS_Game.py:1683-1738).
"""
import copy

import numpy as np
import yaml

N_STATS = 3
N_STATUSES = 3
N_REINF = 3          # reinforcements represented per player (SCS_Game.py:209)


class Unit:
    __slots__ = ("attack", "defense", "mov_allowance", "mov_points", "player", "status", "position", "arrival")

    def __init__(self, attack, defense, mov, player, arrival):
        self.attack, self.defense, self.mov_allowance, self.mov_points = attack, defense, mov, mov
        self.player, self.status, self.position, self.arrival = player, 0, None, arrival


class ScsConfig:
    """The game description of SCS_Game.load_game_from_config (:1570-1779).  "Randomized" maps and victory points
    (:1678-1738) need `map_seed`: what the reference draws from numpy's global stream after np.random.seed(map_seed).
    `map_seed` may also be a RandomState: the map is then drawn from it and the stream is left where the game's own
    draws continue (a game object built and played from one stream: a new map per game, Training/Gamer.py:52)."""

    def __init__(self, path, map_seed=None):
        with open(path) as f:
            d = yaml.safe_load(f)
        self.rows, self.cols = d["Board_dimensions"]["rows"], d["Board_dimensions"]["columns"]
        self.turns, self.stacking = d["Turns"], d["Stacking_limit"]
        units = {p["id"]: p for p in d["Units"].values()}
        # define_board_sides (:1140-1158)
        if self.cols % 2 != 0:
            mid = self.cols // 2
            p1_last, p2_first = mid - 1, mid + 1
        else:
            mid = self.cols // 2
            p1_last, p2_first = max(0, mid - 2), min(self.cols - 1, mid + 1)
        arrival = d["Reinforcements"]["arrival"]
        default_loc = [[], []]
        for i in range(self.rows):
            for j in range(self.cols):
                if j <= p1_last:
                    default_loc[0].append((i, j))
                elif j >= p2_first:
                    default_loc[1].append((i, j))
        self.schedule = [[], []]          # [player][turn] -> list of (attack, defense, movement, arrival)
        counters = [0, 0]
        for key, sched in d["Reinforcements"]["schedule"].items():
            p = int(key[-1]) - 1
            assert len(sched) == self.turns + 1
            for turn_units in sched:
                row = []
                for uid in turn_units:
                    u = units[uid]
                    if arrival["method"] == "Default":
                        loc = default_loc[p]
                    else:
                        loc = [tuple(pt) for pt in arrival["locations"]["p1" if p == 0 else "p2"][counters[p]]]
                        counters[p] += 1
                    row.append((u["attack"], u["defense"], u["movement"], loc))
                self.schedule[p].append(row)
        terrain = {t["id"]: t for t in d["Terrain"].values()}
        map_ids = d["Map"].get("map_configuration")
        vp = d["Victory_points"].get("vp_locations")
        if d["Map"]["creation_method"] != "Detailed" or d["Victory_points"]["creation_method"] != "Detailed":
            assert map_seed is not None, "Randomized methods draw from numpy's global stream: pass map_seed"
            rs = map_seed if isinstance(map_seed, np.random.RandomState) else np.random.RandomState(map_seed)
            ids = [t["id"] for t in d["Terrain"].values()]            # self.terrain_types, in the section's order (:1665-1676)
            for section, values in d.items():                        # the sections in file order (:1582)
                if section == "Map" and values["creation_method"] == "Randomized":          # :1679-1690
                    dist = values.get("distribution") or [1 / len(ids) for _ in ids]
                    map_ids = [[ids[rs.choice(len(ids), p=dist)] for _ in range(self.cols)] for _ in range(self.rows)]
                elif section == "Victory_points" and values["creation_method"] == "Randomized":   # :1709-1738
                    vp = {"p1": [], "p2": []}
                    cols_of = {"p1": range(p1_last + 1), "p2": range(p2_first, self.cols)}
                    for key in ("p1", "p2"):
                        for _ in range(values["number_vp"][key]):
                            point = (int(rs.choice(range(self.rows))), int(rs.choice(cols_of[key])))
                            while point in vp[key]:
                                point = (int(rs.choice(range(self.rows))), int(rs.choice(cols_of[key])))
                            vp[key].append(point)
        self.terrain = [[(terrain[t]["attack_modifier"], terrain[t]["defense_modifier"], terrain[t]["cost"])
                         for t in row] for row in map_ids]
        self.vp = [[tuple(p) for p in vp["p1"]], [tuple(p) for p in vp["p2"]]]
        s = self.stacking
        # action planes (:147-180) and their borders
        self.planes = 1 + 6 * s + 1 + s + 1 + s + s
        self.placement_limit = 1
        self.movement_limit = 1 + 6 * s
        self.target_limit = self.movement_limit + 1
        self.attackers_limit = self.target_limit + s
        self.confirm_limit = self.attackers_limit + 1
        self.no_move_limit = self.confirm_limit + s
        self.no_fight_limit = self.no_move_limit + s
        self.num_actions = self.planes * self.rows * self.cols
        self.channels = 3 + 2 + 2 * (2 * N_REINF * N_STATS) + 2 * (N_STATS * s * N_STATUSES) + 1 + s + 4 + 1 + 1


class ScsGame:
    def __init__(self, cfg):
        self.cfg = cfg
        c = cfg
        self.stacks = [[[] for _ in range(c.cols)] for _ in range(c.rows)]      # Tile.units
        self.owner = [[-1] * c.cols for _ in range(c.rows)]                     # Tile.player
        self.reinf = [[[Unit(a, d, m, p, loc) for (a, d, m, loc) in turn] for turn in c.schedule[p]]
                      for p in range(2)]
        self.available, self.moved, self.attacked = [[], []], [[], []], [[], []]
        self.target = None
        self.attackers = []
        self.player, self.sub_phase, self.stage, self.turn = 0, 0, -2, 0
        self.length, self.terminal, self.terminal_value = 0, False, 0
        self.state_history, self.child_policy = [], []
        self._update_env()

    def shallow_clone(self):
        """SCS_Game.shallow_clone (:1782-1793): everything but the four histories."""
        hist = (self.state_history, self.child_policy)
        self.state_history, self.child_policy = [], []
        g = copy.deepcopy(self)
        self.state_history, self.child_policy = hist
        return g

    def store_state(self, state):
        self.state_history.append(state)

    def store_search_statistics(self, root):                          # (:1517-1521)
        total = sum(c.visit_count for c in root.children)
        by_action = {c.action: c.visit_count for c in root.children}
        self.child_policy.append([by_action[a] / total if a in by_action else 0
                                  for a in range(self.cfg.num_actions)])

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):                                          # (:1523-1528)
        return (self.terminal_value, self.child_policy[i])

    # ---- board geometry (:1048-1094, :1199-1243): n, ne, se, s, sw, nw ---------------------
    def neighbours(self, pos):
        r, c = pos
        R, C = self.cfg.rows, self.cfg.cols
        even = c % 2 == 0
        out = [None] * 6
        if r - 1 != -1:
            out[0] = (r - 1, c)
        if r + 1 != R:
            out[3] = (r + 1, c)
        if not (c == 0 or (r == 0 and even)):
            out[5] = (r - 1, c - 1) if even else (r, c - 1)
        if not (c == 0 or (r == R - 1 and not even)):
            out[4] = (r, c - 1) if even else (r + 1, c - 1)
        if not (c == C - 1 or (r == 0 and even)):
            out[1] = (r - 1, c + 1) if even else (r, c + 1)
        if not (c == C - 1 or (r == R - 1 and not even)):
            out[2] = (r, c + 1) if even else (r + 1, c + 1)
        return out

    def _mobility(self, unit, consider_units):            # check_mobility (:1096-1111)
        res = [False] * 6
        for i, n in enumerate(self.neighbours(unit.position)):
            if n is None:
                continue
            if unit.mov_points - self.cfg.terrain[n[0]][n[1]][2] >= 0:
                res[i] = True
                if consider_units and (len(self.stacks[n[0]][n[1]]) == self.cfg.stacking
                                       or self.owner[n[0]][n[1]] == (unit.player ^ 1)):
                    res[i] = False
        return res

    def _adjacent_units(self, pos, player):               # check_adjacent_units (:1113-1126)
        out = []
        for n in self.neighbours(pos):
            if n is not None:
                out.extend(u for u in self.stacks[n[0]][n[1]] if u.player == player)
        return out

    # ---- duck-typed Game surface -----------------------------------------------------------
    def get_current_player(self):
        return self.player

    def is_terminal(self):
        return self.terminal

    def get_terminal_value(self):
        return self.terminal_value

    def get_length(self):
        return self.length

    def get_num_actions(self):
        return self.cfg.num_actions

    def possible_actions(self):
        """int8 [planes, rows, cols] (:395-484)."""
        c = self.cfg
        m = np.zeros((c.planes, c.rows, c.cols), np.int8)
        p, s_lim = self.player, c.stacking
        if self.sub_phase == 0:
            for (r, k) in self.reinf[p][self.turn][0].arrival:
                if not (self.owner[r][k] == (p ^ 1) or len(self.stacks[r][k]) == s_lim):
                    m[0, r, k] = 1
        elif self.sub_phase == 1:
            for u in self.available[p]:
                r, k = u.position
                s = self.stacks[r][k].index(u)
                m[c.confirm_limit + s, r, k] = 1
                nb, mob = self.neighbours(u.position), self._mobility(u, True)
                for i in range(6):
                    if nb[i] is not None and mob[i]:
                        m[c.placement_limit + i * s_lim + s, r, k] = 1
        elif self.sub_phase == 2:
            for u in self.moved[p]:
                r, k = u.position
                m[c.no_move_limit + self.stacks[r][k].index(u), r, k] = 1
                for e in self._adjacent_units(u.position, p ^ 1):
                    m[c.movement_limit, e.position[0], e.position[1]] = 1
        else:
            for u in self._adjacent_units(self.target, p):
                if u in self.attackers or u in self.attacked[p]:
                    continue
                r, k = u.position
                m[c.target_limit + self.stacks[r][k].index(u), r, k] = 1
            if len(self.attackers) > 0:
                m[c.attackers_limit, self.target[0], self.target[1]] = 1
        return m

    def step_index(self, action_i):
        """step (:375-391) with action_coords = unravel(action_i, (planes, rows, cols))."""
        c = self.cfg
        plane, rem = divmod(int(action_i), c.rows * c.cols)
        r, k = divmod(rem, c.cols)
        start = (r, k)
        if plane < c.placement_limit:                                  # placement
            u = self.reinf[self.player][self.turn].pop(0)
            u.position = start
            self.available[self.player].append(u)
            self._place(u, start)
        elif plane < c.movement_limit:                                 # movement
            idx = plane - c.placement_limit
            s, direction = idx % c.stacking, idx // c.stacking
            dest = self.neighbours(start)[direction]
            u = self.stacks[r][k][s]
            u.mov_points -= c.terrain[dest[0]][dest[1]][2]
            u.position = dest
            self._place(u, dest)
            self._remove(u, start)
            if not any(self._mobility(u, False)):
                self._end_movement(u)
        elif plane < c.target_limit:
            self.target = start
        elif plane < c.attackers_limit:
            self.attackers.append(self.stacks[r][k][plane - c.target_limit])
        elif plane < c.confirm_limit:
            self._resolve_combat()
            self.target = None
            self.attackers = []
        elif plane < c.no_move_limit:
            self._end_movement(self.stacks[r][k][plane - c.confirm_limit])
        else:
            self._end_fighting(self.stacks[r][k][plane - c.no_move_limit])
        self.length += 1
        self._update_env()

    def _place(self, u, pos):                                          # Tile.place_unit
        self.owner[pos[0]][pos[1]] = u.player
        self.stacks[pos[0]][pos[1]].append(u)

    def _remove(self, u, pos):                                         # Tile.remove_unit
        if len(self.stacks[pos[0]][pos[1]]) == 1:
            self.owner[pos[0]][pos[1]] = -1
        self.stacks[pos[0]][pos[1]].remove(u)

    def _end_movement(self, u):                                        # (:927-940)
        u.status = 1
        self.moved[u.player].append(u)
        self.available[u.player].remove(u)
        if len(self._adjacent_units(u.position, u.player ^ 1)) == 0:
            self._end_fighting(u)

    def _end_fighting(self, u):                                        # (:942-946)
        u.status = 2
        self.attacked[u.player].append(u)
        self.moved[u.player].remove(u)

    def _destroy(self, u):                                             # (:982-995)
        self._remove(u, u.position)
        [self.available, self.moved, self.attacked][u.status][u.player].remove(u)

    def _resolve_combat(self):                                         # (:997-1044)
        c = self.cfg
        tr, tk = self.target
        defenders = self.stacks[tr][tk]
        total_defense = sum(u.defense for u in defenders) * c.terrain[tr][tk][1]
        total_attack = 0
        for u in self.attackers:
            total_attack += u.attack * c.terrain[u.position[0]][u.position[1]][0]
            self._end_fighting(u)
        att_loss = 1 if total_attack <= total_defense else 0
        def_loss = 1 if total_attack >= total_defense else 0
        for _ in range(att_loss):
            self._destroy(self._strongest(self.attackers, ("attack", "defense", "mov_allowance")))
        for _ in range(def_loss):
            self._destroy(self._strongest(defenders, ("defense", "attack", "mov_allowance")))

    @staticmethod
    def _strongest(units, keys):                                       # (:1253-1285): first strict maximum
        best = units[0]
        for u in units:
            if tuple(getattr(u, k) for k in keys) > tuple(getattr(best, k) for k in keys):
                best = u
        return best

    def _update_env(self):                                             # update_game_env (:687-831)
        stage, done = self.stage, False
        while True:
            if stage == -2:
                if self.reinf[0][self.turn] == []:
                    stage += 1
                    continue
            elif stage == -1:
                if self.reinf[1][self.turn] == []:
                    self.turn += 1
                    stage += 1
                    continue
            elif stage in (0, 4):
                if self.reinf[stage // 4][self.turn] == []:
                    stage += 1
                    continue
            elif stage in (1, 5):
                if self.available[stage // 4] == []:
                    stage += 1
                    continue
            elif stage == 2:
                if self.moved[0] == []:
                    stage = 4
                    continue
                elif self.target is not None:
                    stage += 1
                    continue
            elif stage == 6:
                if self.moved[1] == []:
                    if self.turn + 1 > self.cfg.turns:
                        done = True
                        break
                    self.turn += 1
                    stage = 0
                    self._new_turn()
                    continue
                elif self.target is not None:
                    stage += 1
                    continue
            elif stage in (3, 7):
                if self.target is None:
                    stage -= 1
                    continue
            break
        self.player = 0 if stage in (-2, 0, 1, 2, 3) else 1
        if done:
            self.terminal = True
            self._check_termination()
        self.sub_phase = 0 if stage in (-2, -1, 0, 4) else 1 if stage in (1, 5) else 2 if stage in (2, 6) else 3
        self.stage = stage

    def _new_turn(self):                                               # (:845-855)
        self.available = [self.attacked[0], self.attacked[1]]
        self.attacked = [[], []]
        for p in (0, 1):
            for u in self.available[p]:
                u.mov_points = u.mov_allowance
                u.status = 0

    def _check_termination(self):                                      # (:857-894)
        vp = self.cfg.vp
        p2_cap = sum(1 for (r, k) in vp[0] if self.owner[r][k] == 1)
        p1_cap = sum(1 for (r, k) in vp[1] if self.owner[r][k] == 0)
        a, b = p1_cap / len(vp[1]), p2_cap / len(vp[0])
        self.terminal_value = 1 if a > b else -1 if a < b else 0

    def state_image(self):
        """float32 [1, C, rows, cols], channel order of generate_state (:1348-1505)."""
        c = self.cfg
        R, C, S = c.rows, c.cols, c.stacking
        img = np.zeros((c.channels, R, C), np.float32)
        for i in range(R):
            for j in range(C):
                img[0, i, j], img[1, i, j], img[2, i, j] = c.terrain[i][j]
        for p in (0, 1):
            for (r, k) in c.vp[p]:
                img[3 + p, r, k] = 1.0
        base = 5
        for p in (0, 1):
            shown = 0
            for turn, units in enumerate(self.reinf[p]):
                importance = ((c.turns + 1) - (turn - self.turn)) / (c.turns + 1)
                for u in units:
                    o = base + p * 2 * N_REINF * N_STATS + shown * 2 * N_STATS
                    for (r, k) in u.arrival:
                        img[o, r, k], img[o + 1, r, k], img[o + 2, r, k] = u.attack, u.defense, u.mov_points
                    img[o + 3:o + 6] = importance
                    shown += 1
                    if shown == N_REINF:
                        break
                if shown == N_REINF:
                    break
        base += 2 * 2 * N_REINF * N_STATS
        per_player = N_STATS * S * N_STATUSES
        for p in (0, 1):
            for status, lst in enumerate((self.available[p], self.moved[p], self.attacked[p])):
                for u in lst:
                    r, k = u.position
                    o = base + p * per_player + status * S * N_STATS + self.stacks[r][k].index(u) * N_STATS
                    img[o, r, k], img[o + 1, r, k], img[o + 2, r, k] = u.attack, u.defense, u.mov_points
        base += 2 * per_player
        if self.target is not None:
            img[base, self.target[0], self.target[1]] = 1.0
        base += 1
        for u in self.attackers:
            r, k = u.position
            img[base + self.stacks[r][k].index(u), r, k] = 1.0
        base += S
        img[base + self.sub_phase] = 1.0
        base += 4
        img[base] = self.turn / c.turns
        img[base + 1] = -1.0 if self.player == 1 else 1.0
        return img[None]

    generate_network_input = state_image
