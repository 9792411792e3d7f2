// Tree kernels of the self-play engine (gfx950): batched AlphaZero MCTS over
// thousands of independent game trees.  One 16-lane DPP row owns one game.
//
// Reference semantics restated here (all paths relative to the reference repo):
//   select   Search/Explorer.py:99-130  (PUCT score, ties -> larger action)
//   expand   Search/Explorer.py:137-181 (mask, renormalise, children ascending)
//   backup   Search/Explorer.py:132-135 (same value added along the path)
//   noise    Search/Explorer.py:201-210
//   action   Search/Explorer.py:70-97,183-199
//   move     Training/Gamer.py:64-79
//   rules    Games/Tic_Tac_Toe/tic_tac_toe.py:121-133,161-167,198-262
//
// Exactness: the reference does its tree arithmetic in IEEE double, one
// operation at a time.  This file is compiled with -ffp-contract=off so no
// multiply-add is fused, log() and sqrt() of the parent count come from tables
// the host fills with glibc's libm (what CPython's math.log/sqrt call), and
// sums follow numpy's pairwise order.  Given the same leaf evaluations the
// visit counts are therefore bit-identical to the reference's.
//
// Storage: structure-of-arrays per game arena; a node's children are
// contiguous in ascending action order, so lane j of the row reads child j
// with one coalesced access per array.
#include "engine.h"

namespace nz {
namespace {

constexpr int BLOCK = 256;
constexpr int GAMES_PER_BLOCK = BLOCK / LANES_PER_GAME;

// ---- Tic-Tac-Toe on bitboards ------------------------------------------------
__device__ __forceinline__ bool ttt_line(uint32_t m) {
  return (m & 0007u) == 0007u || (m & 0070u) == 0070u || (m & 0700u) == 0700u ||
         (m & 0111u) == 0111u || (m & 0222u) == 0222u || (m & 0444u) == 0444u ||
         (m & 0421u) == 0421u || (m & 0124u) == 0124u;
}
__device__ __forceinline__ uint32_t ttt_p1(uint32_t b) { return b & 0x1ffu; }
__device__ __forceinline__ uint32_t ttt_p2(uint32_t b) { return (b >> 16) & 0x1ffu; }
__device__ __forceinline__ int ttt_length(uint32_t b) { return __popc(ttt_p1(b) | ttt_p2(b)); }
__device__ __forceinline__ uint32_t ttt_empty(uint32_t b) { return ~(ttt_p1(b) | ttt_p2(b)) & 0x1ffu; }
// player to move: (length % 2) + 1            (tic_tac_toe.py:165)
__device__ __forceinline__ int ttt_player(uint32_t b) { return (ttt_length(b) & 1) + 1; }
__device__ __forceinline__ uint32_t ttt_step(uint32_t b, int action) {
  return b | ((1u << action) << (ttt_player(b) == 1 ? 0 : 16));
}
// 0 = not terminal, 1 = draw, 2 = player one won (+1), 3 = player two won (-1)
__device__ __forceinline__ int ttt_terminal(uint32_t b) {
  if (ttt_line(ttt_p1(b))) return 2;
  if (ttt_line(ttt_p2(b))) return 3;
  return ttt_length(b) == 9 ? 1 : 0;
}
__device__ __forceinline__ int term_value(int code) { return code == 2 ? 1 : (code == 3 ? -1 : 0); }
__device__ __forceinline__ int ttt_code(uint32_t b) {   // sum cell[a] * 3^a
  int code = 0;
#pragma unroll
  for (int a = 8; a >= 0; --a)
    code = code * 3 + (int)((b >> a) & 1u) + 2 * (int)((b >> (16 + a)) & 1u);
  return code;
}

// ---- row (16-lane) helpers ---------------------------------------------------
__device__ __forceinline__ double row_get(double v, int lane) { return __shfl(v, lane, LANES_PER_GAME); }
__device__ __forceinline__ float row_getf(float v, int lane) { return __shfl(v, lane, LANES_PER_GAME); }

// numpy's pairwise sum for n = 9: eight running sums, then the tail
// (verified against np.sum in tests/test_rng_host.py::test_pairwise_order)
__device__ __forceinline__ double sum9(double v, int /*sub*/) {
  const double a0 = row_get(v, 0), a1 = row_get(v, 1), a2 = row_get(v, 2), a3 = row_get(v, 3);
  const double a4 = row_get(v, 4), a5 = row_get(v, 5), a6 = row_get(v, 6), a7 = row_get(v, 7);
  const double a8 = row_get(v, 8);
  double r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  return r + a8;
}
__device__ __forceinline__ float sum9f(float v) {
  const float a0 = row_getf(v, 0), a1 = row_getf(v, 1), a2 = row_getf(v, 2), a3 = row_getf(v, 3);
  const float a4 = row_getf(v, 4), a5 = row_getf(v, 5), a6 = row_getf(v, 6), a7 = row_getf(v, 7);
  const float a8 = row_getf(v, 8);
  float r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  return r + a8;
}

// scipy.special.softmax over the nine logits of a row, float32
// (Explorer.py:159: exp(x - max) / sum, numpy's summation order)
__device__ __forceinline__ float row_softmax9(float logit, int sub) {
  float m = sub < 9 ? logit : -INFINITY;
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1) m = fmaxf(m, __shfl_xor(m, w, LANES_PER_GAME));
  const float e = sub < 9 ? expf(logit - m) : 0.0f;
  const float s = sum9f(e);
  return e / s;
}

struct Arena {
  int32_t* visit;
  double* value_sum;
  double* prior;
  uint2* link;
};
__device__ __forceinline__ Arena arena_of(const TreeParams& p, int g) {
  const size_t off = (size_t)g * (size_t)p.cap;
  return Arena{p.visit + off, p.value_sum + off, p.prior + off, p.link + off};
}

// Explorer.evaluate's expansion (Explorer.py:165-179).  `prob` is this lane's
// post-softmax float32 probability for action `sub`.
__device__ __forceinline__ int expand_row(const TreeParams& p, const Arena& t, int leaf, uint32_t leaf_meta,
                                          uint32_t sb, float prob, int sub, int node_count) {
  const uint32_t empty = ttt_empty(sb);
  const bool legal = sub < 9 && ((empty >> sub) & 1u);
  const double mask = legal ? 1.0 : 0.0;
  double pd = sub < 9 ? (double)prob * mask : 0.0;
  double total = sum9(pd, sub);
  if (total == 0.0) {                       // network put no mass on legal moves
    pd = pd + mask;
    total = sum9(pd, sub);
  }
  const int k = __popc(empty);
  const int base = node_count;
  if (base + k > p.cap) {
    if (sub == 0) atomicOr(p.error_flag, 1);
    return node_count;
  }
  if (legal) {
    const int c = base + __popc(empty & ((1u << sub) - 1u));
    t.visit[c] = 0;
    t.value_sum[c] = 0.0;
    t.prior[c] = pd / total;
    t.link[c] = make_uint2(0u, pack_meta(0u, (uint32_t)sub, TO_PLAY_UNSET, 0u));
  }
  if (sub == 0)
    t.link[leaf] = make_uint2((uint32_t)base,
                              pack_meta((uint32_t)k, meta_action(leaf_meta), (uint32_t)ttt_player(sb), 0u));
  return node_count + k;
}

// Explorer.backpropagate (Explorer.py:132-135): lane i owns path node i.
__device__ __forceinline__ void backup_row(const Arena& t, int my_node, int path_len, double value, int sub) {
  if (sub < path_len) {
    t.visit[my_node] += 1;
    t.value_sum[my_node] = t.value_sum[my_node] + value;
  }
}

// ---- reset -------------------------------------------------------------------
__global__ void reset_kernel(TreeParams p) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0) {
    p.leaf_count[0] = 0;
    p.leaf_count[1] = 0;
    *p.error_flag = 0;
  }
  if (g >= p.n_games) return;
  const Arena t = arena_of(p, g);
  t.visit[0] = 0;
  t.value_sum[0] = 0.0;
  t.prior[0] = 0.0;
  t.link[0] = make_uint2(0u, pack_meta(0u, 0u, TO_PLAY_UNSET, 0u));
  p.board[g] = 0u;
  p.length[g] = 0;
  p.alive[g] = 1;
  p.outcome[g] = 0;
  p.root[g] = 0;
  p.node_count[g] = 1;
  p.sims_left[g] = p.sims;
  p.pending[g] = -1;
  p.leaf_board[g] = 0u;
  p.path_len[g] = 0;
  p.sim_count[g] = 0;
  p.exp_count[g] = 0;
  p.sel_nodes[g] = 0;
  p.sel_children[g] = 0;
  p.n_root_children[g] = 0;
  for (int m = 0; m < TTT_MAX_MOVES; ++m) {
    const int gm = g * TTT_MAX_MOVES + m;
    p.hist_board[gm] = 0u;
    p.hist_action[gm] = -1;
    p.hist_tree_size[gm] = 0;
    p.hist_children[gm] = 0;
    p.hist_bias[gm] = 0.0;
    p.hist_root_value_sum[gm] = 0.0;
    for (int a = 0; a < TTT_ACTIONS; ++a) {
      p.hist_visits[gm * TTT_ACTIONS + a] = 0;
      p.hist_prior[gm * TTT_ACTIONS + a] = 0.0;
      p.hist_value_sum[gm * TTT_ACTIONS + a] = 0.0;
    }
  }
}

// ---- root noise (Explorer.py:201-210) ----------------------------------------
__global__ __launch_bounds__(BLOCK) void noise_kernel(TreeParams p, const double* __restrict__ noise) {
  const int g = (blockIdx.x * BLOCK + threadIdx.x) / LANES_PER_GAME;
  const int sub = threadIdx.x & (LANES_PER_GAME - 1);
  if (g >= p.n_games || !p.alive[g]) return;
  const Arena t = arena_of(p, g);
  const uint2 lk = t.link[p.root[g]];
  const int k = (int)meta_children(lk.y);
  if (sub < k) {
    const int c = (int)lk.x + sub;
    const double n = noise[(size_t)g * TTT_ACTIONS + sub];
    const double a = t.prior[c] * p.one_minus_frac;
    const double b = n * p.frac;
    t.prior[c] = a + b;
  }
}

// ---- advance: finish the pending expansion, then simulate until the next leaf
// that needs the network (or until this move's simulations are used up) -------
__global__ __launch_bounds__(BLOCK) void advance_kernel(TreeParams p, int iteration) {
  __shared__ int s_count;
  __shared__ int s_base;
  const int tid = threadIdx.x;
  const int g = (blockIdx.x * BLOCK + tid) / LANES_PER_GAME;
  const int sub = tid & (LANES_PER_GAME - 1);
  if (tid == 0) s_count = 0;
  if (blockIdx.x == 0 && tid == 0) p.leaf_count[(iteration + 1) & 1] = 0;
  __syncthreads();

  bool want_slot = false;
  uint32_t slot_board = 0u;
  int my_rank = 0;

  if (g < p.n_games && p.alive[g]) {
    const Arena t = arena_of(p, g);
    int sims_left = p.sims_left[g];
    int node_count = p.node_count[g];
    const int root = p.root[g];
    const uint32_t board = p.board[g];
    const int pend = p.pending[g];
    int n_sim = 0, n_exp = 0, n_lvl = 0, n_kid = 0;
    int my_node = 0;       // lane i remembers path node i
    int path_len = 0;

    if (pend >= 0) {
      // network result for the leaf queued by the previous launch
      path_len = p.path_len[g];
      my_node = p.path[g * MAX_PATH + sub];
      const int leaf = p.path[g * MAX_PATH + path_len - 1];
      const uint32_t sb = p.leaf_board[g];
      const float logit = sub < 9 ? p.leaf_logits[(size_t)pend * TTT_ACTIONS + sub] : 0.0f;
      const float prob = row_softmax9(logit, sub);
      const double value = (double)p.leaf_value[pend];
      node_count = expand_row(p, t, leaf, t.link[leaf].y, sb, prob, sub, node_count);
      backup_row(t, my_node, path_len, value, sub);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      --sims_left;
      ++n_sim;
      ++n_exp;
    }

    while (sims_left > 0) {
      // -- descend (Explorer.py:51-58) -----------------------------------------
      int node = root;
      uint32_t sb = board;
      path_len = 1;
      if (sub == 0) my_node = root;
      uint2 lk = t.link[node];
      while (meta_children(lk.y) != 0u) {
        const int k = (int)meta_children(lk.y);
        const int base = (int)lk.x;
        ++n_lvl;
        n_kid += k;
        const int n_parent = t.visit[node];
        if (n_parent >= p.tab_len) {
          if (sub == 0) atomicOr(p.error_flag, 2);
          break;
        }
        const double sq = p.sqrt_tab[n_parent];
        const double cb = p.bias_tab[n_parent];
        const bool negate = (int)meta_to_play(lk.y) == p.negate_player;
        double score = -INFINITY;
        int action = -1;
        int child = base;
        uint2 clk = make_uint2(0u, 0u);
        if (sub < k) {
          child = base + sub;
          const int n = t.visit[child];
          const double vs = t.value_sum[child];
          const double pr = t.prior[child];
          clk = t.link[child];
          const double u = sq / (double)(n + 1);
          double conf = pr * u;
          conf = conf * cb;
          double q = (n == 0) ? 0.0 : vs / (double)n;
          if (negate) q = -q;
          q = q * p.value_factor;
          score = conf + q;
          action = (int)meta_action(clk.y);
        }
        // max over (score, action): larger action wins a tie (Explorer.py:100)
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1) {
          const double os = __shfl_xor(score, w, LANES_PER_GAME);
          const int oa = __shfl_xor(action, w, LANES_PER_GAME);
          const int oc = __shfl_xor(child, w, LANES_PER_GAME);
          const uint32_t ox = __shfl_xor(clk.x, w, LANES_PER_GAME);
          const uint32_t oy = __shfl_xor(clk.y, w, LANES_PER_GAME);
          if (os > score || (os == score && oa > action)) {
            score = os; action = oa; child = oc; clk.x = ox; clk.y = oy;
          }
        }
        sb = ttt_step(sb, action);
        node = child;
        lk = clk;
        if (sub == path_len) my_node = node;
        ++path_len;
      }

      // -- evaluate (Explorer.py:137-181) --------------------------------------
      const int term = ttt_terminal(sb);
      if (term != 0) {
        if (sub == 0)
          t.link[node] = make_uint2(0u, pack_meta(0u, meta_action(lk.y), (uint32_t)ttt_player(sb), (uint32_t)term));
        backup_row(t, my_node, path_len, (double)term_value(term), sub);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        --sims_left;
        ++n_sim;
        continue;
      }
      if (p.table != nullptr) {
        const float* row = p.table + (size_t)ttt_code(sb) * 10;
        const float prob = sub < 9 ? row[sub] : 0.0f;
        const double value = (double)row[9];
        node_count = expand_row(p, t, node, lk.y, sb, prob, sub, node_count);
        backup_row(t, my_node, path_len, value, sub);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        --sims_left;
        ++n_sim;
        ++n_exp;
        continue;
      }
      // leaf needs the network: park the simulation until the next launch
      want_slot = true;
      slot_board = sb;
      p.path[g * MAX_PATH + sub] = my_node;
      if (sub == 0) {
        p.path_len[g] = path_len;
        p.leaf_board[g] = sb;
      }
      break;
    }

    if (sub == 0) {
      p.sims_left[g] = sims_left;
      p.node_count[g] = node_count;
      p.sim_count[g] += n_sim;
      p.exp_count[g] += n_exp;
      p.sel_nodes[g] += n_lvl;
      p.sel_children[g] += n_kid;
      if (!want_slot) p.pending[g] = -1;
    }
  }

  // one global atomic per block for the leaf queue
  if (want_slot && sub == 0) my_rank = atomicAdd(&s_count, 1);
  __syncthreads();
  if (tid == 0 && s_count > 0) s_base = atomicAdd(&p.leaf_count[iteration & 1], s_count);
  __syncthreads();
  if (want_slot && sub == 0) {
    const int slot = s_base + my_rank;
    p.leaf_boards[slot] = slot_board;
    p.pending[g] = slot;
  }
}

// ---- end of a move: select_action, step, statistics, re-root -----------------
// numpy pairwise sum for n <= 9 values held in an array
__device__ inline double np_sum(const double* v, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r = r + v[i];
    return r;
  }
  double r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  for (int i = 8; i < n; ++i) r = r + v[i];
  return r;
}
// np.random.choice(p=...): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
__device__ inline int np_choice(const double* prob, int n, double u) {
  double cdf[TTT_ACTIONS];
  double run = 0.0;
  for (int i = 0; i < n; ++i) {
    run = (i == 0) ? prob[0] : run + prob[i];
    cdf[i] = run;
  }
  const double last = cdf[n - 1];
  int idx = 0;
  while (idx < n && !(cdf[idx] / last > u)) ++idx;
  return idx < n ? idx : n - 1;
}

__global__ void finish_move_kernel(TreeParams p, const double* __restrict__ uniforms) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  if (!p.alive[g]) {
    p.n_root_children[g] = 0;
    return;
  }
  const Arena t = arena_of(p, g);
  const int root = p.root[g];
  const uint2 lk = t.link[root];
  const int k = (int)meta_children(lk.y);
  const int base = (int)lk.x;
  const int move = p.length[g];
  const uint32_t board = p.board[g];
  const int gm = g * TTT_MAX_MOVES + move;
  if (k == 0 || p.sims_left[g] != 0 || p.pending[g] >= 0) {
    atomicOr(p.error_flag, 4);      // search did not complete
    return;
  }

  int counts[TTT_ACTIONS], actions[TTT_ACTIONS];
  for (int j = 0; j < k; ++j) {
    const int c = base + j;
    counts[j] = t.visit[c];
    actions[j] = (int)meta_action(t.link[c].y);
    p.hist_visits[gm * TTT_ACTIONS + actions[j]] = counts[j];
    p.hist_prior[gm * TTT_ACTIONS + actions[j]] = t.prior[c];
    p.hist_value_sum[gm * TTT_ACTIONS + actions[j]] = t.value_sum[c];
  }
  const int root_visits = t.visit[root];
  p.hist_board[gm] = board;
  p.hist_tree_size[gm] = root_visits;
  p.hist_children[gm] = k;
  p.hist_bias[gm] = root_visits < p.tab_len ? p.bias_tab[root_visits] : 0.0;
  p.hist_root_value_sum[gm] = t.value_sum[root];

  // Explorer.select_action (Explorer.py:70-97)
  int mode = 0;       // 0 max, 1 softmax over visit counts, 2 uniform over legal
  double u3 = 0.0;
  if (p.training) {
    const double u1 = uniforms[g * 3 + 0], u2 = uniforms[g * 3 + 1];
    u3 = uniforms[g * 3 + 2];
    if (move < p.softmax_moves) mode = 1;
    else if (u1 < p.eps_softmax) mode = 1;
    else if (u2 < p.eps_random) mode = 2;
  }
  int chosen;
  if (mode == 0) {            // max_action: first maximum in child order
    int best = 0;
    for (int j = 1; j < k; ++j)
      if (counts[j] > counts[best]) best = j;
    chosen = actions[best];
  } else if (mode == 1) {     // softmax_action (Explorer.py:187-199)
    int mx = counts[0];
    for (int j = 1; j < k; ++j) mx = counts[j] > mx ? counts[j] : mx;
    double e[TTT_ACTIONS];
    for (int j = 0; j < k; ++j) e[j] = exp((double)(counts[j] - mx));
    const double s = np_sum(e, k);
    for (int j = 0; j < k; ++j) e[j] = e[j] / s;
    const double s2 = np_sum(e, k);
    for (int j = 0; j < k; ++j) e[j] = e[j] / s2;
    chosen = actions[np_choice(e, k, u3)];
  } else {                    // uniform over legal actions (Explorer.py:86-89)
    const uint32_t empty = ttt_empty(board);
    double m[TTT_ACTIONS];
    for (int a = 0; a < TTT_ACTIONS; ++a) m[a] = ((empty >> a) & 1u) ? 1.0 : 0.0;
    const double n_valid = np_sum(m, TTT_ACTIONS);
    for (int a = 0; a < TTT_ACTIONS; ++a) m[a] = m[a] / n_valid;
    chosen = np_choice(m, TTT_ACTIONS, u3);
  }
  p.hist_action[gm] = chosen;

  // game.step (tic_tac_toe.py:161-167) and re-rooting (Gamer.py:74-79)
  const uint32_t nb = ttt_step(board, chosen);
  p.board[g] = nb;
  p.length[g] = move + 1;
  int new_root = base;
  for (int j = 0; j < k; ++j)
    if (actions[j] == chosen) new_root = base + j;
  p.root[g] = new_root;
  p.sims_left[g] = p.sims;
  p.pending[g] = -1;
  const int term = ttt_terminal(nb);
  if (term != 0) {
    p.alive[g] = 0;
    p.outcome[g] = term_value(term);
    p.n_root_children[g] = 0;
  } else {
    p.n_root_children[g] = (int)meta_children(t.link[new_root].y);
  }
}

// ---- export ------------------------------------------------------------------
__global__ void export_states_kernel(TreeParams p, float* __restrict__ states) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over G*T*18
  const int total = p.n_games * TTT_MAX_MOVES * 18;
  if (idx >= total) return;
  const int cell = idx % 9, plane = (idx / 9) % 2, gm = idx / 18;
  const int g = gm / TTT_MAX_MOVES, m = gm % TTT_MAX_MOVES;
  float v = 0.0f;
  if (m < p.length[g]) v = (float)((p.hist_board[gm] >> (cell + 16 * plane)) & 1u);
  states[idx] = v;
}

__global__ void export_moves_kernel(TreeParams p, int32_t* visits, int32_t* actions, int32_t* tree_size,
                                    int32_t* n_children, double* bias) {
  const int gm = blockIdx.x * blockDim.x + threadIdx.x;
  if (gm >= p.n_games * TTT_MAX_MOVES) return;
  if (visits)
    for (int a = 0; a < TTT_ACTIONS; ++a) visits[gm * TTT_ACTIONS + a] = p.hist_visits[gm * TTT_ACTIONS + a];
  if (actions) actions[gm] = p.hist_action[gm];
  if (tree_size) tree_size[gm] = p.hist_tree_size[gm];
  if (n_children) n_children[gm] = p.hist_children[gm];
  if (bias) bias[gm] = p.hist_bias[gm];
}

}  // namespace

void launch_reset(const TreeParams& p, hipStream_t s) {
  hipLaunchKernelGGL(reset_kernel, dim3((p.n_games + 255) / 256), dim3(256), 0, s, p);
}
void launch_noise(const TreeParams& p, const double* noise, hipStream_t s) {
  const int blocks = (p.n_games + GAMES_PER_BLOCK - 1) / GAMES_PER_BLOCK;
  hipLaunchKernelGGL(noise_kernel, dim3(blocks), dim3(BLOCK), 0, s, p, noise);
}
void launch_advance(const TreeParams& p, int iteration, hipStream_t s) {
  const int blocks = (p.n_games + GAMES_PER_BLOCK - 1) / GAMES_PER_BLOCK;
  hipLaunchKernelGGL(advance_kernel, dim3(blocks), dim3(BLOCK), 0, s, p, iteration);
}
void launch_finish_move(const TreeParams& p, const double* uniforms, hipStream_t s) {
  hipLaunchKernelGGL(finish_move_kernel, dim3((p.n_games + 255) / 256), dim3(256), 0, s, p, uniforms);
}
void launch_export_states(const TreeParams& p, float* states, hipStream_t s) {
  const int total = p.n_games * TTT_MAX_MOVES * 18;
  hipLaunchKernelGGL(export_states_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, states);
}
void launch_export_visits(const TreeParams& p, int32_t* visits, int32_t* actions, int32_t* tree_size,
                          int32_t* n_children, double* bias, hipStream_t s) {
  const int total = p.n_games * TTT_MAX_MOVES;
  hipLaunchKernelGGL(export_moves_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, visits, actions,
                     tree_size, n_children, bias);
}

}  // namespace nz
