// SCS (hex war-game) rules on the device: state transition, legal-move mask, state image.
//
// Restates Games/SCS/SCS_Game.py of the reference: the 10-stage turn machine (:687-831),
// possible_actions (:395-484), parse/play_action (:486-633), end_movement / end_fighting
// (:927-946), resolve_combat (:997-1044, :1253-1285), the hex neighbourhood with column-parity
// offsets (:1048-1094, :1199-1243), check_termination (:857-894), generate_state (:1348-1505).
//
// The reference keeps Python object lists; what its results depend on is
//   * the order of the units on a tile (stacking level = list index, Tile.py:24-28),
//   * the order in which attackers were selected (get_strongest_attacker walks that list),
//   * the reinforcement queues' schedule order (pop(0)),
// while the available / moved / attacked lists only matter as sets.  So a game is a flat
// record: per unit {status, tile, movement points}, per tile an ordered stack of unit ids and
// an owner, the ordered attacker list, the target tile and the turn-machine registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/nuzero_amd.h"

namespace nz {

constexpr int SCS_MAX_TILES = 100;    // 10 x 10
constexpr int SCS_MAX_STACK = 3;
constexpr int SCS_MAX_UNITS = 32;
constexpr int SCS_MAX_TURNS = 16;

// unit status: queued (not yet placed), the reference's 0/1/2, destroyed
enum : int8_t { SCS_QUEUED = -1, SCS_AVAILABLE = 0, SCS_MOVED = 1, SCS_ATTACKED = 2, SCS_DEAD = 3 };

struct ScsRules {                     // immutable game description (one per engine, device memory)
  int32_t rows, cols, tiles, turns, stacking, n_units, planes, channels;
  int32_t placement_limit, movement_limit, target_limit, attackers_limit, confirm_limit, no_move_limit,
      no_fight_limit;
  int8_t neighbour[SCS_MAX_TILES][6];          // n, ne, se, s, sw, nw; -1 = off board
  float terrain_f[SCS_MAX_TILES][3];           // attack modifier, defense modifier, cost as float32 (image)
  double attack_mod[SCS_MAX_TILES], defense_mod[SCS_MAX_TILES];
  int32_t cost[SCS_MAX_TILES];
  int32_t n_vp[2];
  int8_t vp[2][SCS_MAX_TILES];
  // units in schedule order: player 0's turn 0 .. turn T, then player 1's
  int8_t u_player[SCS_MAX_UNITS], u_turn[SCS_MAX_UNITS], u_attack[SCS_MAX_UNITS], u_defense[SCS_MAX_UNITS],
      u_mov[SCS_MAX_UNITS];
  uint8_t arrival[SCS_MAX_UNITS][SCS_MAX_TILES];
};

bool scs_fill_rules(const nz_scs_desc* d, ScsRules* out, std::string* err);   // scs.hip (host)
void scs_apply_map(ScsRules* r, const float* terrain, const int32_t* vp);      // scs.hip (host): a game's own map

struct ScsState {
  int16_t stage, turn, length;
  int8_t player, sub_phase, terminal, terminal_value, target, n_attackers;
  int8_t attackers[SCS_MAX_UNITS];
  int8_t status[SCS_MAX_UNITS], tile[SCS_MAX_UNITS], mov[SCS_MAX_UNITS];
  int8_t stack_n[SCS_MAX_TILES], owner[SCS_MAX_TILES];
  int8_t stack[SCS_MAX_TILES][SCS_MAX_STACK];
};

struct Scs {
  const ScsRules& r;
  ScsState& s;
  __device__ Scs(const ScsRules& rules, ScsState& state) : r(rules), s(state) {}

  // ---- helpers ------------------------------------------------------------------------------
  __device__ int level_of(int u) const {                 // Tile.get_stacking_level
    const int t = s.tile[u];
    for (int i = 0; i < s.stack_n[t]; ++i)
      if (s.stack[t][i] == u) return i;
    return 0;
  }
  __device__ void place(int u, int t) {                  // Tile.place_unit
    s.owner[t] = r.u_player[u];
    s.stack[t][s.stack_n[t]++] = (int8_t)u;
  }
  __device__ void remove(int u, int t) {                 // Tile.remove_unit
    if (s.stack_n[t] == 1) s.owner[t] = -1;
    int i = 0;
    while (s.stack[t][i] != u) ++i;
    for (; i + 1 < s.stack_n[t]; ++i) s.stack[t][i] = s.stack[t][i + 1];
    --s.stack_n[t];
  }
  __device__ int count(int player, int status) const {
    int n = 0;
    for (int u = 0; u < r.n_units; ++u) n += (r.u_player[u] == player && s.status[u] == status);
    return n;
  }
  __device__ int queue_head(int player, int turn) const {   // current_reinforcements[player][turn][0] or -1
    for (int u = 0; u < r.n_units; ++u)
      if (r.u_player[u] == player && r.u_turn[u] == turn && s.status[u] == SCS_QUEUED) return u;
    return -1;
  }
  __device__ bool can_move(int u, int dir, bool consider_units) const {   // check_mobility (:1096-1111)
    const int n = r.neighbour[s.tile[u]][dir];
    if (n < 0) return false;
    if (s.mov[u] - r.cost[n] < 0) return false;
    if (consider_units && (s.stack_n[n] == r.stacking || s.owner[n] == (r.u_player[u] ^ 1))) return false;
    return true;
  }
  __device__ bool enemy_adjacent(int t, int enemy) const {   // len(check_adjacent_units) > 0
    for (int d = 0; d < 6; ++d) {
      const int n = r.neighbour[t][d];
      if (n < 0) continue;
      for (int i = 0; i < s.stack_n[n]; ++i)
        if (r.u_player[s.stack[n][i]] == enemy) return true;
    }
    return false;
  }
  __device__ void end_fighting(int u) { s.status[u] = SCS_ATTACKED; }              // (:942-946)
  __device__ void end_movement(int u) {                                            // (:927-940)
    s.status[u] = SCS_MOVED;
    if (!enemy_adjacent(s.tile[u], r.u_player[u] ^ 1)) end_fighting(u);
  }
  __device__ void destroy(int u) {                                                 // (:982-995)
    remove(u, s.tile[u]);
    s.status[u] = SCS_DEAD;
  }
  // first strict maximum of (k1, k2, movement allowance) in list order (:1253-1285)
  __device__ int strongest(const int8_t* list, int n, bool attacker) const {
    int best = list[0];
    for (int i = 0; i < n; ++i) {
      const int u = list[i];
      const int a1 = attacker ? r.u_attack[u] : r.u_defense[u], b1 = attacker ? r.u_attack[best] : r.u_defense[best];
      const int a2 = attacker ? r.u_defense[u] : r.u_attack[u], b2 = attacker ? r.u_defense[best] : r.u_attack[best];
      if (a1 > b1 || (a1 == b1 && (a2 > b2 || (a2 == b2 && r.u_mov[u] > r.u_mov[best])))) best = u;
    }
    return best;
  }
  __device__ void resolve_combat() {                                               // (:997-1044)
    const int t = s.target;
    double total_defense = 0.0;
    int dsum = 0;
    for (int i = 0; i < s.stack_n[t]; ++i) dsum += r.u_defense[s.stack[t][i]];
    total_defense = (double)dsum * r.defense_mod[t];
    double total_attack = 0.0;
    for (int i = 0; i < s.n_attackers; ++i) {
      const int u = s.attackers[i];
      total_attack = total_attack + (double)r.u_attack[u] * r.attack_mod[s.tile[u]];
      end_fighting(u);
    }
    if (total_attack <= total_defense) destroy(strongest(s.attackers, s.n_attackers, true));
    if (total_attack >= total_defense) destroy(strongest(s.stack[t], s.stack_n[t], false));
  }
  __device__ void new_turn() {                                                     // (:845-855)
    for (int u = 0; u < r.n_units; ++u)
      if (s.status[u] == SCS_ATTACKED) {
        s.status[u] = SCS_AVAILABLE;
        s.mov[u] = r.u_mov[u];
      }
  }
  __device__ void check_termination() {                                            // (:857-894)
    int p1_cap = 0, p2_cap = 0;
    for (int i = 0; i < r.n_vp[0]; ++i) p2_cap += s.owner[r.vp[0][i]] == 1;
    for (int i = 0; i < r.n_vp[1]; ++i) p1_cap += s.owner[r.vp[1][i]] == 0;
    const double a = (double)p1_cap / (double)r.n_vp[1], b = (double)p2_cap / (double)r.n_vp[0];
    s.terminal_value = a > b ? 1 : (a < b ? -1 : 0);
  }
  __device__ void update_env() {                                                   // update_game_env (:687-831)
    int stage = s.stage;
    bool done = false;
    for (;;) {
      if (stage == -2) {
        if (queue_head(0, s.turn) < 0) { ++stage; continue; }
      } else if (stage == -1) {
        if (queue_head(1, s.turn) < 0) { ++s.turn; ++stage; continue; }
      } else if (stage == 0 || stage == 4) {
        if (queue_head(stage >> 2, s.turn) < 0) { ++stage; continue; }
      } else if (stage == 1 || stage == 5) {
        if (count(stage >> 2, SCS_AVAILABLE) == 0) { ++stage; continue; }
      } else if (stage == 2) {
        if (count(0, SCS_MOVED) == 0) { stage = 4; continue; }
        if (s.target >= 0) { ++stage; continue; }
      } else if (stage == 6) {
        if (count(1, SCS_MOVED) == 0) {
          if (s.turn + 1 > r.turns) { done = true; break; }
          ++s.turn;
          stage = 0;
          new_turn();
          continue;
        }
        if (s.target >= 0) { ++stage; continue; }
      } else {   // 3, 7
        if (s.target < 0) { --stage; continue; }
      }
      break;
    }
    s.player = (stage == -2 || (stage >= 0 && stage <= 3)) ? 0 : 1;
    if (done) {
      s.terminal = 1;
      check_termination();
    }
    s.sub_phase = (stage == -2 || stage == -1 || stage == 0 || stage == 4) ? 0
                  : (stage == 1 || stage == 5) ? 1 : (stage == 2 || stage == 6) ? 2 : 3;
    s.stage = (int16_t)stage;
  }

  // ---- public operations ----------------------------------------------------------------------
  __device__ void reset() {
    s.stage = -2; s.turn = 0; s.length = 0;
    s.player = 0; s.sub_phase = 0; s.terminal = 0; s.terminal_value = 0; s.target = -1; s.n_attackers = 0;
    for (int u = 0; u < SCS_MAX_UNITS; ++u) { s.status[u] = SCS_QUEUED; s.tile[u] = 0; s.mov[u] = u < r.n_units ? r.u_mov[u] : 0; s.attackers[u] = 0; }
    for (int t = 0; t < SCS_MAX_TILES; ++t) {
      s.stack_n[t] = 0; s.owner[t] = -1;
      for (int i = 0; i < SCS_MAX_STACK; ++i) s.stack[t][i] = -1;
    }
    update_env();
  }

  // calls f(action_index) for every legal action (possible_actions, :395-484); order is irrelevant
  template <typename F>
  __device__ void for_each_legal(F&& f) const {
    const int p = s.player, S = r.stacking, T = r.tiles;
    if (s.sub_phase == 0) {
      const int u = queue_head(p, s.turn);
      for (int t = 0; t < T; ++t)
        if (r.arrival[u][t] && !(s.owner[t] == (p ^ 1) || s.stack_n[t] == S)) f(t);
    } else if (s.sub_phase == 1) {
      for (int u = 0; u < r.n_units; ++u) {
        if (r.u_player[u] != p || s.status[u] != SCS_AVAILABLE) continue;
        const int t = s.tile[u], lvl = level_of(u);
        f((r.confirm_limit + lvl) * T + t);
        for (int d = 0; d < 6; ++d)
          if (can_move(u, d, true)) f((r.placement_limit + d * S + lvl) * T + t);
      }
    } else if (s.sub_phase == 2) {
      for (int u = 0; u < r.n_units; ++u) {
        if (r.u_player[u] != p || s.status[u] != SCS_MOVED) continue;
        const int t = s.tile[u];
        f((r.no_move_limit + level_of(u)) * T + t);
        for (int d = 0; d < 6; ++d) {
          const int n = r.neighbour[t][d];
          if (n < 0) continue;
          for (int i = 0; i < s.stack_n[n]; ++i)
            if (r.u_player[s.stack[n][i]] == (p ^ 1)) { f(r.movement_limit * T + n); break; }
        }
      }
    } else {
      for (int d = 0; d < 6; ++d) {
        const int n = r.neighbour[s.target][d];
        if (n < 0) continue;
        for (int i = 0; i < s.stack_n[n]; ++i) {
          const int u = s.stack[n][i];
          if (r.u_player[u] != p || s.status[u] == SCS_ATTACKED) continue;
          bool chosen = false;
          for (int k = 0; k < s.n_attackers; ++k) chosen |= s.attackers[k] == u;
          if (!chosen) f((r.target_limit + i) * T + n);
        }
      }
      if (s.n_attackers > 0) f(r.attackers_limit * T + s.target);
    }
  }

  __device__ void step(int action) {                                               // step (:375-391)
    const int T = r.tiles, S = r.stacking;
    const int plane = action / T, t = action % T;
    if (plane < r.placement_limit) {
      const int u = queue_head(s.player, s.turn);
      s.status[u] = SCS_AVAILABLE;
      s.tile[u] = (int8_t)t;
      place(u, t);
    } else if (plane < r.movement_limit) {
      const int idx = plane - r.placement_limit, lvl = idx % S, dir = idx / S;
      const int u = s.stack[t][lvl], dest = r.neighbour[t][dir];
      s.mov[u] = (int8_t)(s.mov[u] - r.cost[dest]);
      s.tile[u] = (int8_t)dest;
      place(u, dest);
      remove(u, t);
      bool any = false;
      for (int d = 0; d < 6; ++d) any |= can_move(u, d, false);
      if (!any) end_movement(u);
    } else if (plane < r.target_limit) {
      s.target = (int8_t)t;
    } else if (plane < r.attackers_limit) {
      s.attackers[s.n_attackers++] = s.stack[t][plane - r.target_limit];
    } else if (plane < r.confirm_limit) {
      resolve_combat();
      s.target = -1;
      s.n_attackers = 0;
    } else if (plane < r.no_move_limit) {
      end_movement(s.stack[t][plane - r.confirm_limit]);
    } else {
      end_fighting(s.stack[t][plane - r.no_move_limit]);
    }
    ++s.length;
    update_env();
  }

  // generate_state (:1348-1505); img is [channels][tiles] float32, fully written
  __device__ void state_image(float* __restrict__ img) const {
    const int T = r.tiles, S = r.stacking;
    for (int i = 0; i < r.channels * T; ++i) img[i] = 0.0f;
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < 3; ++k) img[k * T + t] = r.terrain_f[t][k];
    for (int p = 0; p < 2; ++p)
      for (int i = 0; i < r.n_vp[p]; ++i) img[(3 + p) * T + r.vp[p][i]] = 1.0f;
    int base = 5;
    for (int p = 0; p < 2; ++p) {
      int shown = 0;
      for (int u = 0; u < r.n_units && shown < 3; ++u) {       // schedule order = turn order within a player
        if (r.u_player[u] != p || s.status[u] != SCS_QUEUED) continue;
        const double importance = (double)((r.turns + 1) - (r.u_turn[u] - s.turn)) / (double)(r.turns + 1);
        const int o = base + p * 18 + shown * 6;
        for (int t = 0; t < T; ++t) {
          if (r.arrival[u][t]) {
            img[o * T + t] = (float)r.u_attack[u];
            img[(o + 1) * T + t] = (float)r.u_defense[u];
            img[(o + 2) * T + t] = (float)s.mov[u];
          }
          img[(o + 3) * T + t] = img[(o + 4) * T + t] = img[(o + 5) * T + t] = (float)importance;
        }
        ++shown;
      }
    }
    base += 36;
    const int per_player = 3 * S * 3;
    for (int u = 0; u < r.n_units; ++u) {
      const int st = s.status[u];
      if (st < SCS_AVAILABLE || st > SCS_ATTACKED) continue;
      const int o = base + r.u_player[u] * per_player + st * S * 3 + level_of(u) * 3, t = s.tile[u];
      img[o * T + t] = (float)r.u_attack[u];
      img[(o + 1) * T + t] = (float)r.u_defense[u];
      img[(o + 2) * T + t] = (float)s.mov[u];
    }
    base += 2 * per_player;
    if (s.target >= 0) img[base * T + s.target] = 1.0f;
    base += 1;
    for (int i = 0; i < s.n_attackers; ++i) {
      const int u = s.attackers[i];
      if (s.status[u] == SCS_DEAD) continue;
      img[(base + level_of(u)) * T + s.tile[u]] = 1.0f;
    }
    base += S;
    const float turn_v = (float)((double)s.turn / (double)r.turns);
    const float player_v = s.player == 1 ? -1.0f : 1.0f;
    for (int t = 0; t < T; ++t) {
      img[(base + s.sub_phase) * T + t] = 1.0f;
      img[(base + 4) * T + t] = turn_v;
      img[(base + 5) * T + t] = player_v;
    }
  }
};

// ---- wavefront-cooperative forms (one 64-lane wavefront per game, state and rules in LDS) ------
// Same results as Scs::for_each_legal / Scs::state_image, spread over the lanes: lane u owns
// unit u (SCS_MAX_UNITS <= 64), lanes stride over tiles.  All 64 lanes must call these together.

// The lanes of the wavefront meet: ONEWAVE = the workgroup is this one wavefront (__syncthreads is then no more than
// that); otherwise the workgroup holds several independent wavefronts (the persistent self-play kernel: one game each)
// and only this wavefront's own LDS accesses are ordered -- they issue in order, so a compiler fence is all it takes.
template <bool ONEWAVE>
__device__ __forceinline__ void scs_sync() {
  if constexpr (ONEWAVE) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// mask: MASK words in LDS, zeroed here; bit a set <=> action a is legal.
template <int WORDS, bool ONEWAVE = true>
__device__ __forceinline__ void scs_legal_mask_wave(const ScsRules& r, const ScsState& s, uint32_t* mask, int lane) {
  for (int i = lane; i < WORDS; i += 64) mask[i] = 0u;
  scs_sync<ONEWAVE>();
  auto f = [&](int a) { atomicOr(&mask[a >> 5], 1u << (a & 31)); };
  const Scs game(r, const_cast<ScsState&>(s));
  const int p = s.player, S = r.stacking, T = r.tiles;
  if (s.sub_phase == 0) {
    const int u = game.queue_head(p, s.turn);
    for (int t = lane; t < T; t += 64)
      if (r.arrival[u][t] && !(s.owner[t] == (p ^ 1) || s.stack_n[t] == S)) f(t);
  } else if (s.sub_phase == 1) {
    const int u = lane;
    if (u < r.n_units && r.u_player[u] == p && s.status[u] == SCS_AVAILABLE) {
      const int t = s.tile[u], lvl = game.level_of(u);
      f((r.confirm_limit + lvl) * T + t);
      for (int d = 0; d < 6; ++d)
        if (game.can_move(u, d, true)) f((r.placement_limit + d * S + lvl) * T + t);
    }
  } else if (s.sub_phase == 2) {
    const int u = lane;
    if (u < r.n_units && r.u_player[u] == p && s.status[u] == SCS_MOVED) {
      const int t = s.tile[u];
      f((r.no_move_limit + game.level_of(u)) * T + t);
      for (int d = 0; d < 6; ++d) {
        const int n = r.neighbour[t][d];
        if (n < 0) continue;
        for (int i = 0; i < s.stack_n[n]; ++i)
          if (r.u_player[s.stack[n][i]] == (p ^ 1)) { f(r.movement_limit * T + n); break; }
      }
    }
  } else if (lane < 6) {
    const int n = r.neighbour[s.target][lane];
    if (n >= 0)
      for (int i = 0; i < s.stack_n[n]; ++i) {
        const int u = s.stack[n][i];
        if (r.u_player[u] != p || s.status[u] == SCS_ATTACKED) continue;
        bool chosen = false;
        for (int k = 0; k < s.n_attackers; ++k) chosen |= s.attackers[k] == u;
        if (!chosen) f((r.target_limit + i) * T + n);
      }
    if (lane == 0 && s.n_attackers > 0) f(r.attackers_limit * T + s.target);
  }
  scs_sync<ONEWAVE>();
}

// img: [channels][tiles] float32 in global memory, fully written.
// ROWS = false: img is [channels][tiles] (the NCHW image of one position).  ROWS = true: img points at the
// position's first row of the network's input rows (boardnet.hip: row = cell * 16 + ..., `stride` floats per row,
// channels contiguous), so element (c, t) lives at img[t * 16 * stride + c] -- the leaf batch is then already in
// the layout the conv kernels read and needs no conversion pass.
template <bool ROWS, bool LDS_IMG = false>       // LDS_IMG: the image is in LDS (one wavefront's LDS operations keep their order)
__device__ __forceinline__ void scs_state_image_wave(const ScsRules& r, const ScsState& s, float* __restrict__ img, int stride,
                                                     int lane) {
  auto at = [&](int c, int t) -> float& { return ROWS ? img[(size_t)t * 16 * stride + c] : img[c * r.tiles + t]; };
  const int T = r.tiles, S = r.stacking;
  const int per_player = 3 * S * 3;
  const int unit_base = 5 + 36, target_plane = unit_base + 2 * per_player, att_base = target_plane + 1,
            phase_base = att_base + S;
  // Everything the image needs from the state and the rules is read HERE, before anything depends on it: section by section
  // the reads were a chain of some forty dependent LDS round trips (7.7 k cycles per leaf on the persistent route).
  const int sub_phase = s.sub_phase, turn = s.turn, turns = r.turns, player = s.player, target = s.target;
  const int nu = r.n_units, n_att = s.n_attackers, nv0 = r.n_vp[0], nv1 = r.n_vp[1];
  const bool isu = lane < nu;
  const int u_st = isu ? (int)s.status[lane] : -1, u_pl = isu ? (int)r.u_player[lane] : -1;
  const int u_tile_raw = isu ? (int)s.tile[lane] : 0, u_tile = u_tile_raw >= 0 ? u_tile_raw : 0;
  const int u_att = isu ? (int)r.u_attack[lane] : 0, u_def = isu ? (int)r.u_defense[lane] : 0, u_mov = isu ? (int)s.mov[lane] : 0;
  const int u_turn = isu ? (int)r.u_turn[lane] : 0;
  const int vp0 = lane < nv0 ? (int)r.vp[0][lane] : -1, vp1 = lane < nv1 ? (int)r.vp[1][lane] : -1;
  const int a_u = lane < n_att ? (int)s.attackers[lane] : -1;
  auto level_in = [&](int t, int u) {            // Tile.get_stacking_level: the first place of unit u in tile t's stack
    const int n = s.stack_n[t];
    int st_[SCS_MAX_STACK];
#pragma unroll
    for (int i = 0; i < SCS_MAX_STACK; ++i) st_[i] = s.stack[t][i];
    int lvl = 0;
#pragma unroll
    for (int i = SCS_MAX_STACK - 1; i >= 0; --i) lvl = (i < n && st_[i] == u) ? i : lvl;
    return lvl;
  };
  const int u_lvl = level_in(u_tile, lane);
  const int a_us = a_u >= 0 ? a_u : 0;
  const int a_st = (int)s.status[a_us], a_tile_raw = (int)s.tile[a_us], a_tile = a_tile_raw >= 0 ? a_tile_raw : 0;
  const int a_lvl = level_in(a_tile, a_us);
  const float turn_frac = (float)((double)turn / (double)turns);
  // planes that are dense: written whole; every other plane is zero-filled first
  auto dense = [&](int c) {                      // the value of a plane that is the same on every tile (terrain: per tile)
    float v = 0.0f;
    if (c >= phase_base) {
      const int k = c - phase_base;
      if (k < 4) v = k == sub_phase ? 1.0f : 0.0f;
      else if (k == 4) v = turn_frac;
      else v = player == 1 ? -1.0f : 1.0f;
    }
    return v;
  };
  const int nc4 = (r.channels + 3) >> 2;               // 16-byte chunks of a tile's row
  if (ROWS && T <= 64 && 2 * nc4 <= 64) {
    // Two tiles' rows per pass, a 16-byte chunk per lane: the planes that depend on the state only are made once (the
    // same on every tile), tile t's terrain sits in lane t and is handed round with v_readlane -- the pass is readlanes,
    // selects and one store, with no LDS read to wait for (a tile-by-tile loop with its terrain reads was 4 k cycles of
    // the persistent route's 9 k per leaf).
    float tr[3] = {0.f, 0.f, 0.f};
    if (lane < T) {
#pragma unroll
      for (int k = 0; k < 3; ++k) tr[k] = r.terrain_f[lane][k];
    }
    const int second = lane >= nc4 ? 1 : 0, c = (lane - second * nc4) * 4;
    typedef float img4 __attribute__((ext_vector_type(4)));
    img4 dv;
#pragma unroll
    for (int i = 0; i < 4; ++i) dv[i] = c + i < r.channels ? dense(c + i) : 0.0f;
    for (int t0 = 0; t0 < T; t0 += 2) {
      const int t1 = t0 + 1 < T ? t0 + 1 : t0;
      img4 v = dv;
      if (c == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const uint32_t bits = __builtin_bit_cast(uint32_t, tr[k]);
          const uint32_t a = __builtin_amdgcn_readlane(bits, t0), b = __builtin_amdgcn_readlane(bits, t1);
          v[k] = __builtin_bit_cast(float, second ? b : a);
        }
      }
      const int t = t0 + second;
      if (lane < 2 * nc4 && t < T) *reinterpret_cast<img4*>(img + (size_t)t * 16 * stride + c) = v;
    }
  } else if constexpr (ROWS) {
    // a tile's channels are contiguous: lane = channel (and channel + 64), tile by tile -- no division per element
    const int c0 = lane, c1 = lane + 64;
    const float v0 = dense(c0), v1 = dense(c1);
    for (int t = 0; t < T; ++t) {
      float* row = img + (size_t)t * 16 * stride;
      if (c0 < r.channels) row[c0] = c0 < 3 ? r.terrain_f[t][c0] : v0;
      if (c1 < r.channels) row[c1] = v1;
    }
    for (int c = lane + 128; c < r.channels; c += 64) {        // (more than 128 planes: none of the shipped games)
      const float v = dense(c);
      for (int t = 0; t < T; ++t) img[(size_t)t * 16 * stride + c] = v;
    }
  } else {
    for (int i = lane; i < r.channels * T; i += 64) {
      const int c = i / T, t = i - c * T;
      at(c, t) = c < 3 ? r.terrain_f[t][c] : dense(c);
    }
  }
  if constexpr (LDS_IMG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // the scattered writes below land after the fill
  if (vp0 >= 0) at(3, vp0) = 1.0f;
  if (vp1 >= 0) at(4, vp1) = 1.0f;
  for (int p = 0; p < 2; ++p) {                 // the next three reinforcements of each player, schedule order
    const bool queued = isu && u_pl == p && u_st == SCS_QUEUED;
    unsigned long long m = __ballot(queued);
    for (int shown = 0; shown < 3 && m; ++shown) {
      const int u = __ffsll((long long)m) - 1;              // (wave-uniform: the unit's numbers come out of its lane)
      m &= m - 1;
      const int ut = __builtin_amdgcn_readlane(u_turn, u);
      const float ua = (float)__builtin_amdgcn_readlane(u_att, u), ud = (float)__builtin_amdgcn_readlane(u_def, u),
                  um = (float)__builtin_amdgcn_readlane(u_mov, u);
      const double importance = (double)((turns + 1) - (ut - turn)) / (double)(turns + 1);
      const int o = 5 + p * 18 + shown * 6;
      for (int t = lane; t < T; t += 64) {
        if (r.arrival[u][t]) {
          at(o, t) = ua;
          at(o + 1, t) = ud;
          at(o + 2, t) = um;
        }
        at(o + 3, t) = at(o + 4, t) = at(o + 5, t) = (float)importance;
      }
    }
  }
  if (isu && u_st >= SCS_AVAILABLE && u_st <= SCS_ATTACKED) {
    const int o = unit_base + u_pl * per_player + u_st * S * 3 + u_lvl * 3;
    at(o, u_tile) = (float)u_att;
    at(o + 1, u_tile) = (float)u_def;
    at(o + 2, u_tile) = (float)u_mov;
  }
  if (lane == 0 && target >= 0) at(target_plane, target) = 1.0f;
  if (a_u >= 0 && a_st != SCS_DEAD) at(att_base + a_lvl, a_tile) = 1.0f;
}

// Scs::step by a whole wavefront: lane 0 applies the action (a handful of stores), the turn
// machine's "is any unit of player p queued / available / moved" scans (update_game_env,
// :687-831) become one ballot each instead of a loop over the units.
// (`tiles_magic` = ceil(2^32 / tiles), or 0: action / tiles by one multiply-high instead of an integer division -- exact
// for the action indices of a board of <= 100 tiles; a tree descent makes this step once per level)
template <bool ONEWAVE = true>
__device__ __forceinline__ void scs_step_wave(const ScsRules& r, ScsState& s, int action, int lane, uint32_t tiles_magic = 0u) {
  Scs game(r, s);
  const int T = r.tiles, S = r.stacking;
  const int plane = tiles_magic ? (int)__umulhi((uint32_t)action, tiles_magic) : action / T, t = action - plane * T;
  const bool unit_lane = lane < r.n_units;
  const int pl = unit_lane ? r.u_player[lane] : -1, tu = unit_lane ? r.u_turn[lane] : -1;
  int st = unit_lane ? s.status[lane] : 99;
  if (plane < r.placement_limit) {
    const unsigned long long q = __ballot(pl == s.player && tu == s.turn && st == SCS_QUEUED);
    const int u = __ffsll((long long)q) - 1;
    if (lane == 0) {
      s.status[u] = SCS_AVAILABLE;
      s.tile[u] = (int8_t)t;
      game.place(u, t);
    }
  } else {
    // lane 0 applies the action; the two scans a movement ends with -- "can the unit still move anywhere" (six directions,
    // check_mobility :1096-1111) and "is an enemy adjacent" (six neighbours x stack, end_movement :927-940) -- are one
    // lane per direction / per (neighbour, stack entry) and a ballot instead of loops on one lane
    int ended = -1;                                   // the unit whose movement ends with this action (wave-uniform)
    if (plane < r.movement_limit) {
      const int idx = plane - r.placement_limit, lvl = idx % S, dir = idx / S;
      const int u = s.stack[t][lvl], dest = r.neighbour[t][dir];
      if (lane == 0) {
        s.mov[u] = (int8_t)(s.mov[u] - r.cost[dest]);
        s.tile[u] = (int8_t)dest;
        game.place(u, dest);
        game.remove(u, t);
      }
      scs_sync<ONEWAVE>();
      const bool can = lane < 6 && game.can_move(u, lane, false);
      if (__ballot(can) == 0ull) ended = u;
    } else if (plane < r.target_limit) {
      if (lane == 0) s.target = (int8_t)t;
    } else if (plane < r.attackers_limit) {
      if (lane == 0) s.attackers[s.n_attackers++] = s.stack[t][plane - r.target_limit];
    } else if (plane < r.confirm_limit) {
      if (lane == 0) {
        game.resolve_combat();
        s.target = -1;
        s.n_attackers = 0;
      }
    } else if (plane < r.no_move_limit) {
      ended = s.stack[t][plane - r.confirm_limit];
    } else {
      if (lane == 0) game.end_fighting(s.stack[t][plane - r.no_move_limit]);
    }
    if (ended >= 0) {                                 // end_movement(ended)
      const int tile = s.tile[ended], enemy = r.u_player[ended] ^ 1;
      bool adj = false;
      if (lane < 6 * SCS_MAX_STACK) {
        const int d = lane / SCS_MAX_STACK, i = lane - d * SCS_MAX_STACK;
        const int n = r.neighbour[tile][d];
        adj = n >= 0 && i < s.stack_n[n] && r.u_player[s.stack[n][i]] == enemy;
      }
      const bool any_adj = __ballot(adj) != 0ull;
      if (lane == 0) s.status[ended] = any_adj ? SCS_MOVED : SCS_ATTACKED;
    }
  }
  if (lane == 0) ++s.length;
  scs_sync<ONEWAVE>();
  st = unit_lane ? s.status[lane] : 99;
  int stage = s.stage, turn = s.turn;
  const int target = s.target;
  bool done = false;
  auto none = [&](bool pred) { return __ballot(pred) == 0ull; };
  for (;;) {
    if (stage == -2) {
      if (none(pl == 0 && tu == turn && st == SCS_QUEUED)) { ++stage; continue; }
    } else if (stage == -1) {
      if (none(pl == 1 && tu == turn && st == SCS_QUEUED)) { ++turn; ++stage; continue; }
    } else if (stage == 0 || stage == 4) {
      if (none(pl == (stage >> 2) && tu == turn && st == SCS_QUEUED)) { ++stage; continue; }
    } else if (stage == 1 || stage == 5) {
      if (none(pl == (stage >> 2) && st == SCS_AVAILABLE)) { ++stage; continue; }
    } else if (stage == 2) {
      if (none(pl == 0 && st == SCS_MOVED)) { stage = 4; continue; }
      if (target >= 0) { ++stage; continue; }
    } else if (stage == 6) {
      if (none(pl == 1 && st == SCS_MOVED)) {
        if (turn + 1 > r.turns) { done = true; break; }
        ++turn;
        stage = 0;
        if (st == SCS_ATTACKED) {                       // new_turn (:845-855)
          st = SCS_AVAILABLE;
          s.status[lane] = SCS_AVAILABLE;
          s.mov[lane] = r.u_mov[lane];
        }
        continue;
      }
      if (target >= 0) { ++stage; continue; }
    } else {   // 3, 7
      if (target < 0) { --stage; continue; }
    }
    break;
  }
  scs_sync<ONEWAVE>();
  if (lane == 0) {
    s.turn = (int16_t)turn;
    s.player = (stage == -2 || (stage >= 0 && stage <= 3)) ? 0 : 1;
    if (done) {
      s.terminal = 1;
      game.check_termination();
    }
    s.sub_phase = (stage == -2 || stage == -1 || stage == 0 || stage == 4) ? 0
                  : (stage == 1 || stage == 5) ? 1 : (stage == 2 || stage == 6) ? 2 : 3;
    s.stage = (int16_t)stage;
  }
  scs_sync<ONEWAVE>();
}

}  // namespace nz
