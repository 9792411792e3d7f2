// Persistent self-play kernel (gfx950): one workgroup owns 16 games for their
// whole life and alternates, with no kernel boundary and no inter-workgroup
// traffic, between
//   tree phase  4 waves x 4 games, one 16-lane row per game: finish the pending
//               expansion, simulate until the next leaf that needs the network,
//               and, when a move's simulations are used up, pick the action,
//               step the game, re-root and mix the next root's noise;
//   net phase   the fused RecurrentNet forward for the 16 pending leaves on
//               FP32 MFMA (net_dev.hpp), inputs and outputs in LDS.
// A slot that finishes its game takes the next unplayed game of the round from a
// global counter (the reference's ActorPool hands the next game to whichever
// actor is free, Training/AlphaZero.py:525-577), so a round of n_games games
// runs on n_slots concurrent trees.
// A game's tree is only ever touched by its own row, so games progress through
// their moves independently of each other: the whole
// select -> inference -> expand -> backup -> move cycle (Training/Gamer.py:64-79
// around Search/Explorer.py:40-67) stays on the GPU.  With 4096 games the grid is
// 256 workgroups, one per CU.
//
// A game runs at most sims_per_cycle simulations between two network passes: late in a game
// most simulations end in terminal positions, and one such row would otherwise hold the other
// fifteen (and the matrix pipes) for dozens of simulations; it simply skips a pass instead.
//
// Randomness.  The reference's draws per move (gamma x n_root_children, two
// uniforms, at most one more inside np.random.choice; SURVEY.md appendix A rule
// 13) come from a per-game host stream.  The host pre-draws every move's values
// assuming the root of move m >= 1 is expanded with 9 - m children (move 0's
// root is never expanded).  The kernel checks the assumption at every re-root;
// a game that violates it (its chosen child was never visited) is flagged
// `desync`, stops, and is replayed by the lock-step path, which asks the device
// for the child count before drawing.
#include "net_dev.hpp"
#include "tree_dev.hpp"

namespace nz {
namespace {

// The register copy of the root's children after an expansion: the root itself got children (path of one node) ->
// load them; a child of the root got children (path of two) -> only its link word changed, and the lane that holds it
// knows the new one (what expand_row wrote); deeper expansions do not touch the copy.
__device__ __forceinline__ void expanded_near_root(RootCache& rc, Arena t, int root, int sub, int path_len, int win0,
                                                   int base0, int node_count, uint32_t leaf_meta, uint32_t leaf_sb) {
  if (path_len == 1) {
    root_cache_load(rc, t, root, sub);
  } else if (path_len == 2 && sub == win0 && node_count > base0) {
    rc.lk = make_uint2((uint32_t)base0, pack_meta((uint32_t)(node_count - base0), meta_action(leaf_meta),
                                                  (uint32_t)ttt_player(leaf_sb), 0u));
  }
}

constexpr int LDS_TAB = 960;     // entries of the (sqrt, bias) tables kept in LDS: enough for 100 simulations per move

// The tree phase: 16 games x 16 lanes, two games per wave (the first two DPP rows of each of the eight waves; with four
// games per wave the rows that are in different branches -- expansion, descent, end of move -- take turns).
constexpr int TREE_THREADS = POS * LANES_PER_GAME;
static_assert(NET_THREADS == 2 * TREE_THREADS, "two of a wave's four rows carry a game");
__device__ __forceinline__ bool tree_lane(int tid) { return (tid & 32) == 0; }
__device__ __forceinline__ int tree_slot(int tid) { return (tid >> 6) * 2 + ((tid >> 4) & 1); }      // game slot = network row
__device__ __forceinline__ int tree_index(int tid) { return tree_slot(tid) * LANES_PER_GAME + (tid & 15); }   // into park_lane
// Tree-phase state of one row (game slot); see `park` below.
struct RowState {
  bool alive = false, pending = false;
  int g = 0, root = 0, node_count = 1, sims_left = 0, move = 0;
  uint32_t board = 0u;
  int n_sim = 0, n_exp = 0, n_lvl = 0, n_kid = 0;
  int path_len = 0, leaf = 0, win0 = -1;
  uint32_t leaf_sb = 0u, leaf_meta = 0u;
  int outcome = 0;
  int my_node = 0, my_n = 0;      // per lane: path node i of the parked simulation
  double my_vs = 0.0;
};
constexpr int PARK_ROW_WORDS = 23, PARK_LANE_WORDS = 13;
__device__ __forceinline__ void park(const RowState& s, const RootCache& rc, int32_t (*row)[POS],
                                     int32_t (*lane)[TREE_THREADS], int slot, int sub, int tid) {
  if (sub == 0) {
    row[0][slot] = (s.alive ? 1 : 0) | (s.pending ? 2 : 0);
    row[1][slot] = s.g; row[2][slot] = s.root; row[3][slot] = s.node_count; row[4][slot] = s.sims_left;
    row[5][slot] = s.move; row[6][slot] = (int32_t)s.board; row[7][slot] = s.n_sim; row[8][slot] = s.n_exp;
    row[9][slot] = s.n_lvl; row[10][slot] = s.n_kid; row[11][slot] = s.path_len; row[12][slot] = s.leaf;
    row[13][slot] = s.win0; row[14][slot] = (int32_t)s.leaf_sb; row[15][slot] = (int32_t)s.leaf_meta;
    row[16][slot] = s.outcome; row[17][slot] = rc.k; row[18][slot] = rc.base; row[19][slot] = rc.root_n;
    row[20][slot] = __double2loint(rc.root_vs); row[21][slot] = __double2hiint(rc.root_vs);
    row[22][slot] = (int32_t)rc.root_meta;
  }
  lane[0][tid] = s.my_node; lane[1][tid] = s.my_n;
  lane[2][tid] = __double2loint(s.my_vs); lane[3][tid] = __double2hiint(s.my_vs);
  lane[4][tid] = rc.n;
  lane[5][tid] = __double2loint(rc.vs); lane[6][tid] = __double2hiint(rc.vs);
  lane[7][tid] = __double2loint(rc.q); lane[8][tid] = __double2hiint(rc.q);
  lane[9][tid] = __double2loint(rc.pr); lane[10][tid] = __double2hiint(rc.pr);
  lane[11][tid] = (int32_t)rc.lk.x; lane[12][tid] = (int32_t)rc.lk.y;
}
__device__ __forceinline__ void unpark(RowState& s, RootCache& rc, int32_t (*row)[POS], int32_t (*lane)[TREE_THREADS],
                                       int slot, int sub, int tid) {
  const int f = row[0][slot];
  s.alive = (f & 1) != 0; s.pending = (f & 2) != 0;
  s.g = row[1][slot]; s.root = row[2][slot]; s.node_count = row[3][slot]; s.sims_left = row[4][slot];
  s.move = row[5][slot]; s.board = (uint32_t)row[6][slot]; s.n_sim = row[7][slot]; s.n_exp = row[8][slot];
  s.n_lvl = row[9][slot]; s.n_kid = row[10][slot]; s.path_len = row[11][slot]; s.leaf = row[12][slot];
  s.win0 = row[13][slot]; s.leaf_sb = (uint32_t)row[14][slot]; s.leaf_meta = (uint32_t)row[15][slot];
  s.outcome = row[16][slot]; rc.k = row[17][slot]; rc.base = row[18][slot]; rc.root_n = row[19][slot];
  rc.root_vs = __hiloint2double(row[21][slot], row[20][slot]);
  rc.root_meta = (uint32_t)row[22][slot];
  s.my_node = lane[0][tid]; s.my_n = lane[1][tid];
  s.my_vs = __hiloint2double(lane[3][tid], lane[2][tid]);
  rc.n = lane[4][tid];
  rc.vs = __hiloint2double(lane[6][tid], lane[5][tid]);
  rc.q = __hiloint2double(lane[8][tid], lane[7][tid]);
  rc.pr = __hiloint2double(lane[10][tid], lane[9][tid]);
  rc.lk = make_uint2((uint32_t)lane[11][tid], (uint32_t)lane[12][tid]);
  (void)sub;
}

// STAMPS = true is the diagnostic build: thread 0 of every workgroup adds up the
// shader-clock ticks it spends in each phase ([block][4] = cycles of the loop,
// tree ticks, net ticks, total ticks).  The product build carries no stamp.
struct SelfplayArgs {
  TreeParams p;
  const NetProgram* prog;
  const float* W;
  const double* noise;        // [G][T][A]
  const double* uniforms;     // [G][T][3]
  unsigned long long* stamps;
};
// The kernel's arguments live in the kernarg segment (constant memory).  Each phase reads what it needs from there
// through a pointer the compiler cannot see through, instead of holding ~110 scalar registers of pointers (and
// per-lane addresses derived from them) alive across the network phase, where they would be spilled.
__device__ __forceinline__ const SelfplayArgs& kernel_args() {
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *(const SelfplayArgs*)(const void*)kp;
}
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// The network phase as a real call: its registers are allocated on their own, as in the stand-alone network kernel
// (inlined into the persistent loop the allocator spills ~140 registers around the job loop).
__device__ __forceinline__ void net_phase(const NetProgram* prog, const float* W, float* lds, const float* inp,
                                                    float* out_logits, float* out_value) {
  // arguments of a device function arrive in vector registers; these two are uniform
  auto uniform = [](const void* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
  };
  net_tile(static_cast<const NetProgram*>(uniform(prog)), static_cast<const float*>(uniform(W)), lds, inp, 1, POS,
           out_logits, out_value);
}

template <bool STAMPS>
__global__ __launch_bounds__(NET_THREADS) void selfplay_kernel(SelfplayArgs) {
  __shared__ __attribute__((aligned(16))) float lds[NET_LDS_FLOATS + POS * TTT_ACTIONS + POS];
  float* const inp = lds + NET_BUFFERS * ACT_FLOATS;
  float* const out_logits = inp + INP_FLOATS + VAL_FLOATS;   // [16][9], after net_tile's own areas
  float* const out_value = out_logits + POS * TTT_ACTIONS;

  // K groups may reach past a narrow layer's channels (their weights are zero): no NaN bit patterns in LDS
  for (int idx = threadIdx.x; idx < NET_BUFFERS * ACT_FLOATS; idx += NET_THREADS) lds[idx] = 0.0f;

  // The network phase needs the whole register file (256 + 256 per lane); the tree phase's state waits in LDS
  // meanwhile -- one word per game for what the 16 lanes of a row share, one per lane for the rest -- and lives in
  // registers only between `unpark` and `park`, so nothing is spilled to scratch memory around net_tile.
  __shared__ int32_t park_row[PARK_ROW_WORDS][POS];
  __shared__ int32_t park_lane[PARK_LANE_WORDS][TREE_THREADS];
  // sqrt(N) and log((N + base + 1) / base) + init of the parent count (Explorer.py:103-112), first LDS_TAB entries:
  // the root's children are scored from registers, with these here a descent's first level touches no memory at all
  __shared__ double2 tab[LDS_TAB];
  {
    const TreeParams& p = kernel_args().p;
    for (int i = threadIdx.x; i < LDS_TAB && i < p.tab_len; i += NET_THREADS) tab[i] = make_double2(p.sqrt_tab[i], p.bias_tab[i]);
    const int tid = threadIdx.x, slot = tree_slot(tid), sub = tid & (LANES_PER_GAME - 1);
    if (tree_lane(tid)) {
      const int gslot = blockIdx.x * p.slots_per_wg + slot;
      RootCache rc0;
      root_cache_load(rc0, arena_of(p, gslot < p.n_slots ? gslot : 0), 0, sub);
      RowState st0;
      st0.alive = slot < p.slots_per_wg && gslot < p.n_slots && gslot < p.n_games;
      st0.g = gslot;                                     // game being played in this slot
      st0.sims_left = p.sims;
      park(st0, rc0, park_row, park_lane, slot, sub, tree_index(tid));
    }
  }
  unsigned long long t_tree = 0, t_net = 0, n_cycles = 0, t_begin = 0, t0 = 0, t_finish = 0;
  if constexpr (STAMPS) t_begin = t0 = __builtin_amdgcn_s_memtime();

  __shared__ int s_max_sims;
  unsigned long long crit_sims = 0;
  int cycle = 0;
  for (;;) {
    // ------------------------------ tree phase ---------------------------------
    // (per-lane indices and the arguments are re-derived here every cycle: loop-invariant values would be kept in
    // registers across the network phase)
    const int tid = opaque(threadIdx.x);
    const int slot = tree_slot(tid);                     // game slot in the tile = network row
    const int sub = tid & (LANES_PER_GAME - 1);
    const SelfplayArgs& ka = kernel_args();
    const TreeParams& p = ka.p;
    const int gslot = blockIdx.x * p.slots_per_wg + slot;   // global slot = tree arena (rows past slots_per_wg: never alive)
    const double* const noise = ka.noise;
    const double* const uniforms = ka.uniforms;
    const int tab_n = p.tab_len < LDS_TAB ? p.tab_len : LDS_TAB;
    int cyc_sims = 0;
    if constexpr (STAMPS) {
      if (tid == 0) s_max_sims = 0;
      __syncthreads();
    }
    bool row_alive = false, row_pending = false;
    if (tree_lane(tid)) {                                // lanes 32-63 of every wave only work in the network phase
    RowState st;
    RootCache rc;
    unpark(st, rc, park_row, park_lane, slot, sub, tree_index(tid));
    const Arena t = arena_of(p, gslot < p.n_slots ? gslot : 0);
    bool& alive = st.alive;
    bool& pending = st.pending;
    int &g = st.g, &root = st.root, &node_count = st.node_count, &sims_left = st.sims_left, &move = st.move;
    uint32_t& board = st.board;
    int &n_sim = st.n_sim, &n_exp = st.n_exp, &n_lvl = st.n_lvl, &n_kid = st.n_kid;
    int &my_node = st.my_node, &path_len = st.path_len, &leaf = st.leaf, &my_n = st.my_n, &win0 = st.win0;
    double& my_vs = st.my_vs;
    uint32_t &leaf_sb = st.leaf_sb, &leaf_meta = st.leaf_meta;
    int& outcome = st.outcome;
    if (alive) {
      if (pending) {
        const float logit = sub < 9 ? out_logits[slot * TTT_ACTIONS + sub] : 0.0f;
        const float prob = row_softmax9(logit, sub);
        const double value = (double)out_value[slot];
        const int base0 = node_count;
        node_count = expand_row(p, t, leaf, leaf_meta, leaf_sb, prob, sub, node_count);
        backup_cached(t, rc, my_node, my_n, my_vs, path_len, value, sub, win0);
        row_memory_fence();
        expanded_near_root(rc, t, root, sub, path_len, win0, base0, node_count, leaf_meta, leaf_sb);
        --sims_left;
        ++cyc_sims;
        ++n_sim;
        ++n_exp;
        pending = false;
      }
      while (alive && !pending && cyc_sims < p.sims_per_cycle) {
        if (move < 0) {
          // ---- the slot's game is over: record it, take the next one of the round ----
          if (sub == 0) {
            p.length[g] = -move - 1;
            p.outcome[g] = outcome;
            p.sim_count[g] = n_sim;
            p.exp_count[g] = n_exp;
            p.sel_nodes[g] = n_lvl;
            p.sel_children[g] = n_kid;
            p.new_nodes[g] = node_count - 1;
          }
          int ng = 0;
          if (sub == 0) ng = atomicAdd(p.next_game, 1);
          ng = row_geti(ng, 0);
          if (ng >= p.n_games) { alive = false; break; }
          g = ng;
          if (sub == 0) arena_reset(t);
          row_memory_fence();
          root = 0; node_count = 1; sims_left = p.sims; move = 0; board = 0u; outcome = 0;
          root_cache_load(rc, t, root, sub);
          n_sim = n_exp = n_lvl = n_kid = 0;
          continue;
        }
        if (sims_left == 0) {
          // ---- the move is searched: act, step, re-root (Gamer.py:71-79) ----------
          int chosen = -1, new_root = root, term = 0, new_children = 0;
          uint32_t new_board = board;
          unsigned long long tf0 = 0;
          if constexpr (STAMPS) tf0 = __builtin_amdgcn_s_memtime();
          if (sub == 0) {
            const MoveResult r = finish_move_one(p, t, g, root, board, move,
                                                 uniforms ? uniforms + ((size_t)g * TTT_MAX_MOVES + move) * 3 : nullptr);
            chosen = r.chosen; new_root = r.new_root; term = r.term; new_children = r.new_children;
            new_board = r.new_board;
          }
          chosen = row_geti(chosen, 0);
          new_root = row_geti(new_root, 0);
          term = row_geti(term, 0);
          new_children = row_geti(new_children, 0);
          new_board = (uint32_t)row_geti((int)new_board, 0);
          if constexpr (STAMPS) t_finish += __builtin_amdgcn_s_memtime() - tf0;
          if (chosen < 0) { alive = false; break; }        // error flag already raised (whole round fails)
          board = new_board;
          root = new_root;
          ++move;
          sims_left = p.sims;
          if (term != 0) {
            outcome = term_value(term);
            move = -move - 1;                               // game over after `move` moves
            continue;
          }
          if (p.training) {
            if (new_children != TTT_ACTIONS - move) {       // the host's draw-count assumption failed
              if (sub == 0) p.desync[g] = 1;
              move = -move - 1;                             // abandon; the lock-step route replays it
              continue;
            }
            noise_row(p, t, root, noise + ((size_t)g * TTT_MAX_MOVES + move) * TTT_ACTIONS, sub);
            row_memory_fence();
          }
          root_cache_load(rc, t, root, sub);
          continue;
        }
        const Descent d = descend_cached(p, t, rc, root, board, sub, my_node, my_n, my_vs, win0, tab, tab_n);
        n_lvl += d.levels;
        n_kid += d.children;
        const int term = ttt_terminal(d.sb);
        if (term != 0) {
          if (sub == 0)
            t[d.node].meta = pack_meta(0u, meta_action(d.lk.y), (uint32_t)ttt_player(d.sb), (uint32_t)term);
          backup_cached(t, rc, my_node, my_n, my_vs, d.path_len, (double)term_value(term), sub, win0);
          row_memory_fence();
          --sims_left;
          ++cyc_sims;
          ++n_sim;
          continue;
        }
        if (p.table != nullptr) {
          const float* row = p.table + (size_t)ttt_code(d.sb) * 10;
          const float prob = sub < 9 ? row[sub] : 0.0f;
          const double value = (double)row[9];
          const int base0 = node_count;
          node_count = expand_row(p, t, d.node, d.lk.y, d.sb, prob, sub, node_count);
          backup_cached(t, rc, my_node, my_n, my_vs, d.path_len, value, sub, win0);
          row_memory_fence();
          expanded_near_root(rc, t, root, sub, d.path_len, win0, base0, node_count, d.lk.y, d.sb);
          --sims_left;
          ++cyc_sims;
          ++n_sim;
          ++n_exp;
          continue;
        }
        // leaf needs the network
        pending = true;
        leaf = d.node;
        leaf_meta = d.lk.y;
        leaf_sb = d.sb;
        path_len = d.path_len;
      }
    }
    // this row's input planes: inp[cell][slot][4] = (P1 stone, P2 stone, 0, 0)
    if (sub < CELLS) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pending) {
        v.x = (float)((leaf_sb >> sub) & 1u);
        v.y = (float)((leaf_sb >> (16 + sub)) & 1u);
      }
      *reinterpret_cast<float4*>(inp + (sub * POS + slot) * 4) = v;
    }
    if constexpr (STAMPS) {
      if (sub == 0) atomicMax(&s_max_sims, cyc_sims);
    }
    park(st, rc, park_row, park_lane, slot, sub, tree_index(tid));
    row_alive = alive;
    row_pending = pending;
    }
    const int any_alive = __syncthreads_or(row_alive ? 1 : 0);
    const int any_pending = __syncthreads_or(row_pending ? 1 : 0);
    if constexpr (STAMPS) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      t_tree += t1 - t0;
      t0 = t1;
      crit_sims += (unsigned long long)s_max_sims;
    }
    if (!any_alive) break;
    if (++cycle > p.max_cycles) {     // uniform over the workgroup: never spin forever
      if (tid == 0) atomicOr(p.error_flag, 64);
      break;
    }
    if (!any_pending) continue;       // every live row used up its simulations for this cycle

    // ------------------------------ net phase ----------------------------------
    {
      const SelfplayArgs& na = kernel_args();
      net_phase(na.prog, na.W, lds, inp, out_logits, out_value);
    }
    // net_tile ends with a barrier: outputs are visible to every row
    if constexpr (STAMPS) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      t_net += t1 - t0;
      t0 = t1;
      ++n_cycles;
    }
  }
  const SelfplayArgs& ea = kernel_args();
  const TreeParams& p = ea.p;
  unsigned long long* const stamps = ea.stamps;
  const int tid = opaque(threadIdx.x), slot = tree_slot(tid), sub = tid & (LANES_PER_GAME - 1);   // (not the prologue's copies)
  const int gslot = blockIdx.x * p.slots_per_wg + slot;
  if constexpr (STAMPS) {
    if (tid == 0) {
      stamps[blockIdx.x * 4 + 0] = n_cycles;
      stamps[blockIdx.x * 4 + 1] = t_tree;
      stamps[blockIdx.x * 4 + 2] = t_net;
      stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    // wave 0's own time inside move bookkeeping and inside pending expansions (subset of tree ticks)
    if (tid == 0) {
      stamps[gridDim.x * 4 + blockIdx.x * 2 + 0] = t_finish;
      stamps[gridDim.x * 4 + blockIdx.x * 2 + 1] = crit_sims;   // sum over cycles of the slowest row's simulations
    }
  }

  if (tree_lane(tid) && slot < p.slots_per_wg && gslot < p.n_slots && sub == 0) {
    p.alive[gslot] = 0;
    p.pending[gslot] = -1;
    p.n_root_children[gslot] = 0;
  }
}

}  // namespace

int selfplay_blocks(int n_slots, int slots_per_wg) { return (n_slots + slots_per_wg - 1) / slots_per_wg; }

void launch_selfplay(const TreeParams& p, const NetProgram* prog_dev, int n_layers, const float* weights,
                     const double* noise, const double* uniforms, unsigned long long* stamps, hipStream_t s) {
  const int blocks = selfplay_blocks(p.n_slots, p.slots_per_wg);
  (void)n_layers;
  const SelfplayArgs a{p, prog_dev, weights, noise, uniforms, stamps};
  if (stamps != nullptr) hipLaunchKernelGGL(selfplay_kernel<true>, dim3(blocks), dim3(NET_THREADS), 0, s, a);
  else hipLaunchKernelGGL(selfplay_kernel<false>, dim3(blocks), dim3(NET_THREADS), 0, s, a);
}

}  // namespace nz
