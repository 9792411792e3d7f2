// Device-side building blocks of the tree search, shared by the lock-step
// kernels (tree.hip) and the persistent self-play kernel (selfplay.hip).
// One 16-lane DPP row owns one game tree.
//
// Reference semantics restated here (paths relative to the reference repo):
//   select   Search/Explorer.py:99-130  (PUCT score, ties -> larger action)
//   expand   Search/Explorer.py:137-181 (mask, renormalise, children ascending)
//   backup   Search/Explorer.py:132-135 (same value added along the path)
//   noise    Search/Explorer.py:201-210
//   action   Search/Explorer.py:70-97,183-199
//   move     Training/Gamer.py:64-79
//   rules    Games/Tic_Tac_Toe/tic_tac_toe.py:121-133,161-167,198-262
//
// Exactness: the reference does its tree arithmetic in IEEE double, one
// operation at a time.  Every translation unit that includes this header is
// compiled with -ffp-contract=off so no multiply-add is fused, log() and sqrt()
// of the parent count come from tables the host fills with glibc's libm (what
// CPython's math.log/sqrt call), and sums follow numpy's pairwise order.  Given
// the same leaf evaluations the visit counts are bit-identical to the
// reference's.
//
// Storage: one 48-byte record per node (engine.h TNode), a node's children contiguous in
// ascending action order, so lane j of the row reads child j as three 16-byte loads from
// consecutive addresses (k children = k * 48 contiguous bytes: 4 cache lines for 9).  A record
// carries Q = value_sum / visit_count (Node.value(), Search/Node.py:19-22) beside the two:
// Q changes for exactly one child per level per simulation -- the one backed up -- so the
// backup rewrites it (one division per path node, lane-parallel) and select reads it, instead
// of every level dividing for every child.  Same operands, same IEEE division: same double.
#pragma once
#include "engine.h"

namespace nz {

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
__device__ __forceinline__ int row_geti(int v, int lane) { return __shfl(v, lane, LANES_PER_GAME); }

// numpy's pairwise sum for n = 9: eight running sums, then the tail
// (checked against np.sum in tests/test_rng_host.py::test_pairwise_order)
__device__ __forceinline__ double sum9(double v) {
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

// A game tree is only ever read and written by the lanes of its own row, i.e. by one
// wavefront, and a wavefront's memory operations take effect in program order: ordering
// a simulation's stores before the next simulation's loads needs no wait, only that the
// compiler keeps the order.
__device__ __forceinline__ void row_memory_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

typedef TNode* Arena;
__device__ __forceinline__ Arena arena_of(const TreeParams& p, int g) { return p.nodes + (size_t)g * (size_t)p.cap; }
__device__ __forceinline__ TNode fresh_node(uint32_t action) {
  TNode n;
  n.value_sum = 0.0; n.q = 0.0; n.prior = 0.0; n.visit = 0; n.first = 0u;
  n.meta = pack_meta(0u, action, TO_PLAY_UNSET, 0u);
  n.pad[0] = n.pad[1] = n.pad[2] = 0u;
  return n;
}
__device__ __forceinline__ void arena_reset(Arena t) { t[0] = fresh_node(0u); }   // Node(0), Gamer.py:59

// Explorer.evaluate's expansion (Explorer.py:165-179).  `prob` is this lane's
// post-softmax float32 probability for action `sub`.  Returns the new node count.
__device__ __forceinline__ int expand_row(const TreeParams& p, Arena t, int leaf, uint32_t leaf_meta,
                                          uint32_t sb, float prob, int sub, int node_count) {
  const uint32_t empty = ttt_empty(sb);
  const bool legal = sub < 9 && ((empty >> sub) & 1u);
  const double mask = legal ? 1.0 : 0.0;
  double pd = sub < 9 ? (double)prob * mask : 0.0;
  double total = sum9(pd);
  if (total == 0.0) {                       // network put no mass on legal moves
    pd = pd + mask;
    total = sum9(pd);
  }
  const int k = __popc(empty);
  const int base = node_count;
  if (base + k > p.cap) {
    if (sub == 0) atomicOr(p.error_flag, 1);
    return node_count;
  }
  if (legal) {
    const int c = base + __popc(empty & ((1u << sub) - 1u));
    t[c].value_sum = 0.0;                     // (the three padding words are never read: not written either)
    t[c].q = 0.0;
    t[c].prior = pd / total;
    t[c].visit = 0;
    t[c].first = 0u;
    t[c].meta = pack_meta(0u, (uint32_t)sub, TO_PLAY_UNSET, 0u);
  }
  if (sub == 0) {
    t[leaf].first = (uint32_t)base;
    t[leaf].meta = pack_meta((uint32_t)k, meta_action(leaf_meta), (uint32_t)ttt_player(sb), 0u);
  }
  return node_count + k;
}

// Explorer.backpropagate (Explorer.py:132-135): lane i owns path node i.
__device__ __forceinline__ void backup_row(Arena t, int my_node, int path_len, double value, int sub) {
  if (sub < path_len) {
    const int n = t[my_node].visit + 1;
    const double vs = t[my_node].value_sum + value;
    t[my_node].visit = n;
    t[my_node].value_sum = vs;
    t[my_node].q = vs / (double)n;
  }
}

// PUCT score of one child (Explorer.py:103-130), shared by both descents.  `q` is the child's stored
// value_sum / visit_count (0.0 while unvisited, Node.py:19-22).
__device__ __forceinline__ double puct_score(const TreeParams& p, double sq, double cb, bool negate, int n, double q,
                                             double pr) {
  const double u = sq / (double)(n + 1);
  double conf = pr * u;
  conf = conf * cb;
  if (negate) q = -q;
  q = q * p.value_factor;
  return conf + q;
}

// rotate a value by N lanes inside its 16-lane row (DPP row_ror: no LDS round trip)
template <int N>
__device__ __forceinline__ int row_ror(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ double row_ror(double v) {
  return __hiloint2double(row_ror<N>(__double2hiint(v)), row_ror<N>(__double2loint(v)));
}
// Row-wide argmax of (score, action): the larger action wins a tie (Explorer.py:100: max over (score, action, child)
// tuples).  Children sit in ascending action order, lane j = child j, so that is: the highest lane among those that
// hold the maximum.  Two butterflies of four DPP rotations: the maximum score (v_max_f64), then the highest lane
// whose score equals it (v_max_i32).  Lanes that hold no child pass score = -inf.
__device__ __forceinline__ int row_argmax(double score, int sub) {
  double m = score;
  m = fmax(m, row_ror<8>(m));
  m = fmax(m, row_ror<4>(m));
  m = fmax(m, row_ror<2>(m));
  m = fmax(m, row_ror<1>(m));
  int win = (score == m) ? sub : -1;
  win = max(win, row_ror<8>(win));
  win = max(win, row_ror<4>(win));
  win = max(win, row_ror<2>(win));
  win = max(win, row_ror<1>(win));
  return win;
}

// One descent from `root` (Explorer.py:51-58).  On return `node`/`lk` is the
// leaf and its link word, `sb` the scratch position there, `path_len` the
// number of nodes on the path and lane i's `my_node` is path node i.
struct Descent {
  int node;
  uint2 lk;
  uint32_t sb;
  int path_len;
  int levels, children;   // work done, for the byte accounting
};
__device__ __forceinline__ Descent descend_row(const TreeParams& p, Arena t, int root, uint32_t board,
                                               int sub, int& my_node) {
  Descent d;
  d.node = root;
  d.sb = board;
  d.path_len = 1;
  d.levels = 0;
  d.children = 0;
  if (sub == 0) my_node = root;
  d.lk = make_uint2(t[root].first, t[root].meta);
  int n_parent = t[root].visit;     // below the root it is carried down from the chosen child
  while (meta_children(d.lk.y) != 0u) {
    const int k = (int)meta_children(d.lk.y);
    const int base = (int)d.lk.x;
    ++d.levels;
    d.children += k;
    if (n_parent >= p.tab_len) {
      if (sub == 0) atomicOr(p.error_flag, 2);
      break;
    }
    const double sq = p.sqrt_tab[n_parent];
    const double cb = p.bias_tab[n_parent];
    const bool negate = (int)meta_to_play(d.lk.y) == p.negate_player;
    double score = -INFINITY;
    int n = 0;
    uint2 clk = make_uint2(0u, 0u);
    if (sub < k) {
      const TNode c = t[base + sub];
      n = c.visit;
      clk = make_uint2(c.first, c.meta);
      score = puct_score(p, sq, cb, negate, n, c.q, c.prior);
    }
    const int win = row_argmax(score, sub);
    clk.x = (uint32_t)row_geti((int)clk.x, win);
    clk.y = (uint32_t)row_geti((int)clk.y, win);
    n_parent = row_geti(n, win);
    d.sb = ttt_step(d.sb, (int)meta_action(clk.y));
    d.node = base + win;
    d.lk = clk;
    if (sub == d.path_len) my_node = d.node;
    ++d.path_len;
  }
  return d;
}

// ---- root-resident variant (persistent kernel) ---------------------------------
// Every simulation starts at the root, so the persistent kernel keeps the root's
// statistics and, in lane j, those of root child j in registers for the whole
// move: level 0 of each descent then needs no memory access, and because every
// selection also hands the chosen child's (visit, value_sum) to the lane that
// owns that path node, the backup is a plain store.  Memory stays current
// (write-through), so the end-of-move code and the lock-step kernels read the
// same values.  The arithmetic is the one of descend_row / backup_row.
struct RootCache {
  int k, base;           // children of the root (k = 0: not expanded yet)
  int root_n;
  double root_vs;
  uint32_t root_meta;
  int n;                 // lane j < k: child j
  double vs, q, pr;
  uint2 lk;
};
__device__ __forceinline__ void root_cache_load(RootCache& c, Arena t, int root, int sub) {
  const TNode r = t[root];
  c.k = (int)meta_children(r.meta);
  c.base = (int)r.first;
  c.root_meta = r.meta;
  c.root_n = r.visit;
  c.root_vs = r.value_sum;
  c.n = 0;
  c.vs = 0.0;
  c.q = 0.0;
  c.pr = 0.0;
  c.lk = make_uint2(0u, 0u);
  if (sub < c.k) {
    const TNode ch = t[c.base + sub];
    c.n = ch.visit;
    c.vs = ch.value_sum;
    c.q = ch.q;
    c.pr = ch.prior;
    c.lk = make_uint2(ch.first, ch.meta);
  }
}
// As descend_row; additionally lane i receives path node i's (visit, value_sum) as read
// during the descent (my_n, my_vs) and win0 is the lane of the chosen root child (-1: none).
// `tab` (may be null): the first `tab_n` entries of (sqrt_tab, bias_tab) interleaved, in LDS.
__device__ __forceinline__ Descent descend_cached(const TreeParams& p, Arena t, const RootCache& c, int root,
                                                  uint32_t board, int sub, int& my_node, int& my_n, double& my_vs,
                                                  int& win0, const double2* tab = nullptr, int tab_n = 0) {
  Descent d;
  d.node = root;
  d.sb = board;
  d.path_len = 1;
  d.levels = 0;
  d.children = 0;
  d.lk = make_uint2((uint32_t)c.base, c.root_meta);
  win0 = -1;
  if (sub == 0) {
    my_node = root;
    my_n = c.root_n;
    my_vs = c.root_vs;
  }
  int n_parent = c.root_n;
  while (meta_children(d.lk.y) != 0u) {
    const int k = (int)meta_children(d.lk.y);
    const int base = (int)d.lk.x;
    ++d.levels;
    d.children += k;
    if (n_parent >= p.tab_len) {
      if (sub == 0) atomicOr(p.error_flag, 2);
      break;
    }
    double sq, cb;
    if (n_parent < tab_n) {
      const double2 e = tab[n_parent];
      sq = e.x; cb = e.y;
    } else {
      sq = p.sqrt_tab[n_parent];
      cb = p.bias_tab[n_parent];
    }
    const bool negate = (int)meta_to_play(d.lk.y) == p.negate_player;
    double score = -INFINITY;
    int n = 0;
    double vs = 0.0;
    uint2 clk = make_uint2(0u, 0u);
    if (sub < k) {
      double pr, q;
      if (d.path_len == 1) {          // children of the root: registers
        n = c.n; vs = c.vs; q = c.q; pr = c.pr; clk = c.lk;
      } else {
        const TNode ch = t[base + sub];
        n = ch.visit; vs = ch.value_sum; q = ch.q; pr = ch.prior;
        clk = make_uint2(ch.first, ch.meta);
      }
      score = puct_score(p, sq, cb, negate, n, q, pr);
    }
    const int win = row_argmax(score, sub);
    if (d.path_len == 1) win0 = win;
    clk.x = (uint32_t)row_geti((int)clk.x, win);
    clk.y = (uint32_t)row_geti((int)clk.y, win);
    n_parent = row_geti(n, win);
    const double vs_win = row_get(vs, win);
    d.sb = ttt_step(d.sb, (int)meta_action(clk.y));
    d.node = base + win;
    d.lk = clk;
    if (sub == d.path_len) {
      my_node = d.node;
      my_n = n_parent;
      my_vs = vs_win;
    }
    ++d.path_len;
  }
  return d;
}
// Explorer.backpropagate without loads: lane i stores path node i from the values the descent read
__device__ __forceinline__ void backup_cached(Arena t, RootCache& c, int my_node, int my_n, double my_vs,
                                              int path_len, double value, int sub, int win0) {
  if (sub < path_len) {
    const double vs = my_vs + value;
    t[my_node].visit = my_n + 1;
    t[my_node].value_sum = vs;
    t[my_node].q = vs / (double)(my_n + 1);
  }
  c.root_n += 1;
  c.root_vs = c.root_vs + value;
  if (path_len >= 2 && sub == win0) {
    c.n += 1;
    c.vs = c.vs + value;
    c.q = c.vs / (double)c.n;
  }
}

// Root noise (Explorer.py:201-210): lane j mixes child j's prior.
__device__ __forceinline__ void noise_row(const TreeParams& p, Arena t, int root, const double* noise_row9,
                                          int sub) {
  const int k = (int)meta_children(t[root].meta);
  if (sub < k) {
    const int c = (int)t[root].first + sub;
    const double a = t[c].prior * p.one_minus_frac;
    const double b = noise_row9[sub] * p.frac;
    t[c].prior = a + b;
  }
}

// ---- end of a move, one thread per game ----------------------------------------
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
// exp() behind a real call: inlined, its polynomial's constants are hoisted to the top of the persistent kernel and kept
// (in the self-play kernel: in scratch memory, written and read back every cycle) for a function a game calls once a move
__device__ __attribute__((noinline)) double exp_call(double x) { return exp(x); }
// np.random.choice(p=...): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
__device__ inline int np_choice(const double* prob, int n, double u) {
  double cdf[TTT_ACTIONS] = {};      // fully initialised for the same reason as finish_move_one's arrays
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

struct MoveResult {
  int chosen;        // action index, -1 if the search had not completed
  int new_root;
  uint32_t new_board;
  int term;          // terminal code of the new position
  int new_children;  // children of the new root
};
// Explorer.select_action (Explorer.py:70-97), the per-move records Gamer keeps
// (Gamer.py:71-77,81-92), game.step (tic_tac_toe.py:161-167) and re-rooting
// (Gamer.py:78-79).  `record` is the game's row in the hist_* arrays.
// `forced` >= 0 plays that action instead of the search's own choice (the opponent's move in
// MctsAgent.update_subtree, Testing/Agents/Generic/MctsAgent.py:35-39).
__device__ inline MoveResult finish_move_one(const TreeParams& p, Arena t, int record, int root,
                                             uint32_t board, int move, const double* uni3, int forced = -1) {
  MoveResult r{-1, root, board, 0, 0};
  const int k = (int)meta_children(t[root].meta);
  const int base = (int)t[root].first;
  const int gm = record * TTT_MAX_MOVES + move;
  if (k == 0) {
    atomicOr(p.error_flag, 4);      // search did not complete
    return r;
  }
  // (fully initialised: these arrays become vector registers, and the entries a partial fill leaves alone would
  // otherwise be carried -- as live registers -- around the caller's loop, network phase included)
  int counts[TTT_ACTIONS] = {}, actions[TTT_ACTIONS] = {};
  for (int j = 0; j < k; ++j) {
    const TNode c = t[base + j];
    counts[j] = c.visit;
    actions[j] = (int)meta_action(c.meta);
    p.hist_visits[gm * TTT_ACTIONS + actions[j]] = counts[j];
    p.hist_prior[gm * TTT_ACTIONS + actions[j]] = c.prior;
    p.hist_value_sum[gm * TTT_ACTIONS + actions[j]] = c.value_sum;
  }
  const int root_visits = t[root].visit;
  p.hist_board[gm] = board;
  p.hist_tree_size[gm] = root_visits;
  p.hist_children[gm] = k;
  p.hist_bias[gm] = root_visits < p.tab_len ? p.bias_tab[root_visits] : 0.0;
  p.hist_root_value_sum[gm] = t[root].value_sum;

  int mode = 0;       // 0 max, 1 softmax over visit counts, 2 uniform over legal, 3 forced
  double u3 = 0.0;
  if (forced >= 0) {
    mode = 3;
  } else if (p.training) {
    const double u1 = uni3[0], u2 = uni3[1];
    u3 = uni3[2];
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
    double e[TTT_ACTIONS] = {};
    for (int j = 0; j < k; ++j) e[j] = exp_call((double)(counts[j] - mx));
    const double s = np_sum(e, k);
    for (int j = 0; j < k; ++j) e[j] = e[j] / s;
    const double s2 = np_sum(e, k);
    for (int j = 0; j < k; ++j) e[j] = e[j] / s2;
    chosen = actions[np_choice(e, k, u3)];
  } else if (mode == 3) {
    chosen = -1;
    for (int j = 0; j < k; ++j)
      if (actions[j] == forced) chosen = forced;
    if (chosen < 0) {           // not a legal move of this position
      atomicOr(p.error_flag, 8);
      return r;
    }
  } else {                    // uniform over legal actions (Explorer.py:86-89)
    const uint32_t empty = ttt_empty(board);
    double m[TTT_ACTIONS];
    for (int a = 0; a < TTT_ACTIONS; ++a) m[a] = ((empty >> a) & 1u) ? 1.0 : 0.0;
    const double n_valid = np_sum(m, TTT_ACTIONS);
    for (int a = 0; a < TTT_ACTIONS; ++a) m[a] = m[a] / n_valid;
    chosen = np_choice(m, TTT_ACTIONS, u3);
  }
  p.hist_action[gm] = chosen;
  r.chosen = chosen;
  r.new_board = ttt_step(board, chosen);
  r.new_root = base;
  for (int j = 0; j < k; ++j)
    if (actions[j] == chosen) r.new_root = base + j;
  r.term = ttt_terminal(r.new_board);
  r.new_children = (int)meta_children(t[r.new_root].meta);
  return r;
}

__device__ inline void hist_clear(const TreeParams& p, int record) {
  for (int m = 0; m < TTT_MAX_MOVES; ++m) {
    const int gm = record * TTT_MAX_MOVES + m;
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

}  // namespace nz
