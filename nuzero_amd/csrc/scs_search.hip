// MCTS over SCS game trees on the device, lock-step, with the leaf evaluations supplied from
// outside (C ABI nz_scs_search_* / nz_scs_select / nz_scs_expand ...).  One wavefront owns one
// game: lane j scores child j of the node being left (up to 64 children), lane 0 runs the rules
// (scs_dev.hpp) on the game's scratch state.
//
// Reference semantics (paths relative to the reference repo), as tree_dev.hpp, with the numeric
// types SCS produces under numpy >= 2 (SURVEY.md section 8a row 3, appendix A rule 11):
//   * expansion priors are float32: softmax probs (float32) * int8 mask, np.sum in float32
//     (pairwise), float32 division                                  (Search/Explorer.py:165-179)
//   * score() with a float32 prior runs in float32: the Python floats u, c and q are cast to
//     float32 before each operation                                  (Search/Explorer.py:114-130)
//   * root noise makes the root's children's priors float64 (float32 * (1 - frac) rounded to
//     float32, plus the float64 noise term); from then on their score is float64
//                                                                    (Search/Explorer.py:201-210)
//   * SCS players are 0 and 1, so `parent.to_play == 2` never negates Q (appendix A rule 6)
// The tree arithmetic must not be contracted: this file is compiled with -ffp-contract=off.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/nuzero_amd.h"
#include "scs_dev.hpp"
#include "boardnet_internal.h"

using namespace nz;

namespace {

// Children per node and decisions per game are bounded from the game description when the engine is created
// (nz_scs_search_create: `scs_limits`), not by constants: a wavefront handles a node's children in chunks of 64 lanes.
constexpr int MAXC_CHUNKS = 4;                 // chunks of 64 children a node may have
constexpr int MAXC_LIMIT = 64 * MAXC_CHUNKS;   // 256: more legal actions than that in one position are rejected at create
constexpr int MAX_ACTIONS = 21 * SCS_MAX_TILES;
constexpr int MASK_WORDS = (MAX_ACTIONS + 31) / 32;

struct SNode {                    // array-of-structures: one 32-byte record per node
  double prior;                   // float32 value unless prior_f64
  double value_sum;
  int32_t visit;
  int32_t child_base;
  uint16_t n_children;
  uint16_t action;
  int8_t to_play;                 // -1 until evaluated
  int8_t prior_f64;               // 1: prior went through the root-noise mix (float64 arithmetic)
  int8_t terminal;
  int8_t pad;
};

struct SearchParams {
  int32_t n_games, cap, sims, training, softmax_moves, negate_player, tab_len, max_path;
  double frac, one_minus_frac, value_factor, eps_softmax, eps_random;
  const double* bias_tab;
  const double* sqrt_tab;
  const ScsRules* rules;   // the game description; with per-game maps (nz_scs_search_set_games) one row per game of the round
  const int32_t* rules_row;  // [G] the row of `rules` each slot's game reads, or nullptr: all read row 0
  ScsState* real;          // [G]
  ScsState* scratch;       // [G]
  SNode* nodes;            // [G][2][half_cap]: two halves per game, the live tree is in half `half[g]`
  int8_t* half;            // [G]
  int32_t half_cap;
  int32_t* node_count;     // [G]
  int32_t* root;           // [G]
  int32_t* sims_left;      // [G]
  int32_t* pending;        // [G] leaf slot or -1
  int32_t* path;           // [G][max_path]
  int32_t* path_len;       // [G]
  uint32_t* leaf_mask;     // [G][MASK_WORDS] legal actions at the pending leaf
  int32_t* leaf_count;     // [1]
  int32_t* active_count;   // [1] games that are not done with their move after a wave
  int32_t* clear_counters; // the (leaf, active) pair of the NEXT wave, zeroed by this one (nullptr: the host zeroes)
  // numpy's pairwise float32 sum over num_actions entries as a straight program: leaf blocks [pw_hi[b-1], pw_hi[b])
  // in index order, eight strided partial sums up to pw_be[b] and the rest added one by one, then pw_merge[b]
  // "left + right" merges of the block-sum stack (built on the host for this game's num_actions)
  int32_t num_actions;
  int32_t pw_blocks;
  uint16_t pw_hi[48], pw_be[48];
  uint8_t pw_merge[48];
  int32_t maxc;            // children per node the records hold (a multiple of 64, <= MAXC_LIMIT)
  int32_t max_moves;       // decisions per game the records hold
  int32_t terminal_budget; // simulations ending in terminal leaves one game may run per wave
  int32_t image_row_stride; // > 0: leaf images are written as input rows of a board net (floats per row), else NCHW
  // inference cache (keyless hash table shared by the games of the engine, see cache_put_kernel); c_bits == 0: off
  int32_t c_bits;          // log2(entries)
  int32_t c_wave;          // serial number of this simulation wave (arbitration of same-wave writers)
  int32_t c_hit_base;      // evaluation slots [c_hit_base, c_hit_base + G) receive this wave's cache hits
  uint64_t* c_id;          // [entries][2] the 128-bit hash of the state an entry holds (0, 0: empty)
  float* c_probs;          // [entries][A] post-softmax probabilities
  float* c_value;          // [entries]
  int32_t* c_writer;       // [entries] last wave that wrote the entry (wave-by-wave route)
  uint64_t* c_check;       // [entries] check word over (key, probabilities, value) (persistent route: self-validating entries)
  uint64_t* leaf_key;      // [G][2] hash of each queued (missed) leaf, by evaluation slot
  int32_t* hit_count;      // [1] hits of this wave
  float* eval_probs;       // [3 G][A] evaluations: the network's [0, G), cache hits [G, 2 G) / [2 G, 3 G) by wave parity
  float* eval_value;       // [3 G]
  int32_t* error_flag;
  int64_t* counters;       // [16] simulations, expansions; [8] cache hits, [9] misses, [10] entries in use; [2..7] shader ticks per phase of the NZ_SCS_STAMPS diagnostic build
  // records [G][max_moves]...
  int32_t* rec_action;
  int32_t* rec_tree_size;
  int32_t* rec_children;
  double* rec_bias;
  double* rec_root_value_sum;
  int32_t* rec_child_action;   // [G][max_moves][maxc]
  int32_t* rec_child_visit;
  double* rec_child_prior;
  double* rec_child_value_sum;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
template <typename T>
__device__ __forceinline__ T wave_get(T v, int lane) { return __shfl(v, lane, 64); }

// Wavefront-wide max over (score, key) tuples, the larger key winning a tie (Explorer.py:100: max over (score, action,
// child) tuples; key = action << 8 | child index).  Inside each 16-lane row two butterflies of DPP rotations (the maximum
// score, then the largest key among the lanes that hold it), then the four rows' results meet through v_readlane: no
// trip through the LDS crossbar (six rounds of three ds_bpermute were a thousand cycles per tree level).  Lanes without
// a child pass score = -inf, key = -1.  Returns the winning key in every lane.
template <int N>
__device__ __forceinline__ int dpp_row_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false); }
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
  return __hiloint2double(dpp_row_ror<N>(__double2hiint(v)), dpp_row_ror<N>(__double2loint(v)));
}
__device__ __forceinline__ int wave_argmax_key(double score, int key) {
  double m = score;
  m = fmax(m, dpp_row_ror<8>(m));
  m = fmax(m, dpp_row_ror<4>(m));
  m = fmax(m, dpp_row_ror<2>(m));
  m = fmax(m, dpp_row_ror<1>(m));
  int win = (score == m) ? key : -1;
  win = max(win, dpp_row_ror<8>(win));
  win = max(win, dpp_row_ror<4>(win));
  win = max(win, dpp_row_ror<2>(win));
  win = max(win, dpp_row_ror<1>(win));
  double best = -INFINITY;
  int best_key = -1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double mr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 16 * r),
                                       __builtin_amdgcn_readlane(__double2loint(m), 16 * r));
    const int kr = __builtin_amdgcn_readlane(win, 16 * r);
    if (mr > best || (mr == best && kr > best_key)) { best = mr; best_key = kr; }
  }
  return best_key;
}

// score of one child (Explorer.score, :114-130) in the reference's dtypes
__device__ __forceinline__ double child_score(const SearchParams& p, const SNode& c, double sq, double cb, bool negate) {
  const double u = sq / (double)(c.visit + 1);
  double q = (c.visit == 0) ? 0.0 : c.value_sum / (double)c.visit;
  if (negate) q = -q;
  q = q * p.value_factor;
  if (c.prior_f64) {
    double conf = c.prior * u;
    conf = conf * cb;
    return conf + q;
  }
  float conf = (float)c.prior * (float)u;
  conf = conf * (float)cb;
  return (double)(conf + (float)q);
}

// numpy's float32 pairwise sum (np.sum over all num_actions entries of `probs`, Explorer.py:169) of an array that
// is zero except at k sorted positions; adding a zero is exact, so only the non-zero entries and numpy's block
// structure matter.  Wave-uniform: lane i holds the i-th non-zero entry (idx ascending, val), every lane walks the
// block program of SearchParams (built on the host) with the entries broadcast by v_readlane -- no memory accesses.
// Entry j lives in lane j & 63 of register j >> 6 (up to MAXC_CHUNKS registers per lane).
template <int CH = MAXC_CHUNKS>
__device__ __forceinline__ float np_sum_sparse_f32_wave(const SearchParams& p, const int (&idx)[CH],
                                                        const float (&val)[CH], int k) {
  float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f, st4 = 0.f, st5 = 0.f, st6 = 0.f, st7 = 0.f;   // block-sum stack
  int sp = 0, i = 0, lo = 0;
  auto entry_idx = [&](int j) {
    const int c = __builtin_amdgcn_readfirstlane(j >> 6), l = __builtin_amdgcn_readfirstlane(j & 63);
    int r = idx[0];
    if constexpr (CH > 1) r = c == 0 ? idx[0] : c == 1 ? idx[1] : c == 2 ? idx[2 % CH] : idx[3 % CH];
    return __builtin_amdgcn_readlane(r, l);
  };
  auto entry_val = [&](int j) {
    const int c = __builtin_amdgcn_readfirstlane(j >> 6), l = __builtin_amdgcn_readfirstlane(j & 63);
    float r = val[0];
    if constexpr (CH > 1) r = c == 0 ? val[0] : c == 1 ? val[1] : c == 2 ? val[2 % CH] : val[3 % CH];
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), l));
  };
  for (int b = 0; b < p.pw_blocks; ++b) {
    const int hi = p.pw_hi[b], be = p.pw_be[b];
    // numpy's eight running sums r[0..7] of the block live in lanes 0..7 (one compare and one add per entry instead of
    // eight of each); adding 0.0f to the other seven is exact
    float racc = 0.f;
    const int lane8 = (int)(threadIdx.x & 63);
    while (i < k) {
      const int ix = entry_idx(i);
      if (ix >= be) break;
      const float v = entry_val(i);
      racc += lane8 == ((ix - lo) & 7) ? v : 0.f;
      ++i;
    }
    // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), read from lane 0: lane i takes lane i + 1 / + 2 / + 4 of its
    // row by DPP shifts (only lane 0's chain matters; no trip through the LDS crossbar)
    const float t1 = racc + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, racc), 0x101, 0xF, 0xF, false));   // row_shl:1
    const float t2 = t1 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0x102, 0xF, 0xF, false));       // row_shl:2
    const float t3 = t2 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t2), 0x104, 0xF, 0xF, false));       // row_shl:4
    float res = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t3)));
    while (i < k) {
      if (entry_idx(i) >= hi) break;
      res = res + entry_val(i);
      ++i;
    }
    for (int mrg = p.pw_merge[b];; --mrg) {            // push, then merge: left + right
      switch (sp) {
        case 0: st0 = res; break; case 1: st1 = res; break; case 2: st2 = res; break; case 3: st3 = res; break;
        case 4: st4 = res; break; case 5: st5 = res; break; case 6: st6 = res; break; default: st7 = res; break;
      }
      ++sp;
      if (mrg == 0) break;
      float right, left;
      switch (sp) {
        case 2: left = st0; right = st1; break; case 3: left = st1; right = st2; break; case 4: left = st2; right = st3; break;
        case 5: left = st3; right = st4; break; case 6: left = st4; right = st5; break; case 7: left = st5; right = st6; break;
        default: left = st6; right = st7; break;
      }
      res = left + right;
      sp -= 2;
    }
    lo = hi;
  }
  return st0;
}
// numpy's pairwise sum of a dense float64 vector (the children of one node: n <= MAXC_LIMIT = 256): blocks of at most 128
// entries on eight running sums; a longer vector is halved (the left half a multiple of eight) and the halves' sums are
// added -- two levels of halving reach every n <= 256 (no recursion on the device: DEPTH counts them down)
__device__ inline double np_sum_f64_block(const double* v, int n) {     // n <= 128
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r = r + v[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = v[j];
  int i = 8;
  for (; i < n - n % 8; i += 8)
    for (int j = 0; j < 8; ++j) r[j] = r[j] + v[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res = res + v[i];
  return res;
}
template <int DEPTH = 2>
__device__ inline double np_sum_f64(const double* v, int n) {
  if constexpr (DEPTH > 0) {
    if (n > 128) {
      int n2 = n / 2;
      n2 -= n2 % 8;
      const double left = np_sum_f64<DEPTH - 1>(v, n2);
      return left + np_sum_f64<DEPTH - 1>(v + n2, n - n2);
    }
  }
  return np_sum_f64_block(v, n);
}
static_assert(MAXC_LIMIT <= 256, "np_sum_f64 halves a vector at most twice");

__device__ __forceinline__ const ScsRules& rules_of(const SearchParams& p, int g) {
  return p.rules[p.rules_row ? p.rules_row[g] : 0];
}

// the half of game g's arena that holds its live tree
__device__ __forceinline__ SNode* arena(const SearchParams& p, int g) {
  return p.nodes + (size_t)g * p.cap + (size_t)p.half[g] * p.half_cap;
}

__global__ void search_reset_kernel(SearchParams p) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0) {
    *p.leaf_count = 0; *p.active_count = 0; *p.error_flag = 0;
    for (int i = 0; i < 16; ++i)
      if (i < 8 || i > 10) p.counters[i] = 0;      // [8..10] belong to the inference cache (nz_scs_search_cache)
  }
  if (g >= p.n_games) return;
  Scs(rules_of(p, g), p.real[g]).reset();
  p.half[g] = 0;
  SNode& n = p.nodes[(size_t)g * p.cap];
  n.prior = 0.0; n.value_sum = 0.0; n.visit = 0; n.child_base = 0; n.n_children = 0; n.action = 0;
  n.to_play = -1; n.prior_f64 = 0; n.terminal = 0; n.pad = 0;
  p.node_count[g] = 1;
  p.root[g] = 0;
  p.sims_left[g] = 0;
  p.pending[g] = -1;
  p.path_len[g] = 0;
  for (int m = 0; m < p.max_moves; ++m) {
    p.rec_action[(size_t)g * p.max_moves + m] = -1;
    p.rec_children[(size_t)g * p.max_moves + m] = 0;
    p.rec_tree_size[(size_t)g * p.max_moves + m] = 0;
  }
}

// ---- rounds of more games than slots (nz_scs_search_play_round) ----------------------------------------------------
// A finished game's records leave its slot for the round's store (one row per game of the round), then the slot starts
// the next game of the round: Gamer actors play their games back to back (Gamer.py:45-98), and a round ends no later
// than its longest game instead of idling every slot whose game was short.
struct RoundStore {
  int32_t *action, *tree_size, *children, *child_action, *child_visit, *status;   // status [n][2]: length, terminal value
  double *bias, *root_value_sum, *child_prior, *child_value_sum;
};
// one workgroup per slot; `to_record[g]` = row of the store that takes slot g's game, or -1
__global__ void archive_kernel(SearchParams p, RoundStore st, const int32_t* __restrict__ to_record) {
  const int g = blockIdx.x;
  const int rec = to_record[g];
  if (rec < 0) return;
  const ScsState& real = p.real[g];
  const int M = p.max_moves, MAXC = p.maxc;
  const int len = real.length < M ? real.length : M;
  const size_t src = (size_t)g * M, dst = (size_t)rec * M;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const bool played = m < len;
    st.action[dst + m] = played ? p.rec_action[src + m] : -1;
    st.tree_size[dst + m] = played ? p.rec_tree_size[src + m] : 0;
    st.children[dst + m] = played ? p.rec_children[src + m] : 0;
    st.bias[dst + m] = played ? p.rec_bias[src + m] : 0.0;
    st.root_value_sum[dst + m] = played ? p.rec_root_value_sum[src + m] : 0.0;
  }
  for (int i = threadIdx.x; i < len * MAXC; i += blockDim.x) {
    const int m = i / MAXC, j = i - m * MAXC;
    if (j < p.rec_children[src + m]) {
      st.child_action[dst * MAXC + i] = p.rec_child_action[src * MAXC + i];
      st.child_visit[dst * MAXC + i] = p.rec_child_visit[src * MAXC + i];
      st.child_prior[dst * MAXC + i] = p.rec_child_prior[src * MAXC + i];
      st.child_value_sum[dst * MAXC + i] = p.rec_child_value_sum[src * MAXC + i];
    }
  }
  if (threadIdx.x == 0) {
    st.status[rec * 2] = real.length;
    st.status[rec * 2 + 1] = real.terminal_value;
  }
}
// search_reset_kernel for the slots with restart[g] != 0 (counters and flags are the round's: untouched)
__global__ void restart_kernel(SearchParams p, const int32_t* __restrict__ restart) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games || !restart[g]) return;
  Scs(rules_of(p, g), p.real[g]).reset();
  p.half[g] = 0;
  SNode& n = p.nodes[(size_t)g * p.cap];
  n.prior = 0.0; n.value_sum = 0.0; n.visit = 0; n.child_base = 0; n.n_children = 0; n.action = 0;
  n.to_play = -1; n.prior_f64 = 0; n.terminal = 0; n.pad = 0;
  p.node_count[g] = 1;
  p.root[g] = 0;
  p.sims_left[g] = 0;
  p.pending[g] = -1;
  p.path_len[g] = 0;
  for (int m = 0; m < p.max_moves; ++m) {
    p.rec_action[(size_t)g * p.max_moves + m] = -1;
    p.rec_children[(size_t)g * p.max_moves + m] = 0;
    p.rec_tree_size[(size_t)g * p.max_moves + m] = 0;
  }
}

// children of each live game's root (the number of gamma draws of the next move)
__global__ void root_children_kernel(SearchParams p, int32_t* out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  out[g] = p.real[g].terminal ? 0 : arena(p, g)[p.root[g]].n_children;
}

// add_exploration_noise (Explorer.py:201-210) and the start of a move's search
__global__ void begin_move_kernel(SearchParams p, const double* __restrict__ noise) {
  const int g = blockIdx.x;
  const int lane = lane_id();
  if (p.real[g].terminal) return;
  SNode* nodes = arena(p, g);
  const SNode& root = nodes[p.root[g]];
  for (int j = lane; p.training && j < root.n_children; j += 64) {
    SNode& c = nodes[root.child_base + j];
    const double n = noise[(size_t)g * p.maxc + j];
    double a;
    if (c.prior_f64) a = c.prior * p.one_minus_frac;
    else a = (double)((float)c.prior * (float)p.one_minus_frac);
    const double b = n * p.frac;
    c.prior = a + b;
    c.prior_f64 = 1;
  }
  if (lane == 0) {
    p.sims_left[g] = p.sims;
    p.pending[g] = -1;
  }
}

// ---- inference cache ------------------------------------------------------------------------------------------
// The reference's optional cache (Explorer.py:146-155; Utils/Caches/KeylessCache.py:24-160: hash of the state tensor,
// index bits select the slot, the remaining bits are stored as the entry's id; no keys kept) as one device table
// shared by all games of the engine.  Key: a 128-bit hash of the game STATE the tensor is generated from (equal states
// give equal tensors, so a hit returns what the network would compute; the reference's metrohash is a third-party
// package that is not available here, and which hash is used only matters for which entries collide).  The table
// is only READ inside wave_kernel and only WRITTEN by cache_put_kernel between two wave kernels, so no entry is
// ever seen half-written.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {          // murmur3 finaliser
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
// every lane returns the same (hi, lo): position-tagged words mixed per lane, combined across the wavefront
__device__ __forceinline__ void state_hash_wave(const ScsState& st, int lane, uint64_t& hi, uint64_t& lo) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&st);
  uint64_t a = 0, b = 0;
  for (int i = lane; i < (int)(sizeof(ScsState) / 4); i += 64) {
    const uint64_t v = ((uint64_t)(uint32_t)i << 32) | w[i];
    a += mix64(v ^ 0x9e3779b97f4a7c15ull);
    b += mix64(v * 0xd6e8feb86659fd93ull + 0x2545f4914f6cdd1dull);
  }
  for (int o = 32; o; o >>= 1) {
    a += __shfl_xor((unsigned long long)a, o, 64);
    b += __shfl_xor((unsigned long long)b, o, 64);
  }
  hi = mix64(a ^ (b >> 7));
  lo = mix64(b ^ (a << 9));
  if (hi == 0 && lo == 0) lo = 1;                                 // (0, 0) marks an empty entry
}
__global__ void cache_count_kernel(const uint64_t* __restrict__ c_id, int64_t entries, int64_t* __restrict__ out) {
  long long n = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < entries; e += (int64_t)gridDim.x * blockDim.x)
    n += (c_id[2 * e] != 0 || c_id[2 * e + 1] != 0) ? 1 : 0;
  if (n) atomicAdd((unsigned long long*)out, (unsigned long long)n);
}
// After the network: the wave's evaluated leaves go into the table (KeylessCache.put: the newest entry replaces
// what the slot held).  One wavefront per leaf; of two leaves of the same wave that map to the same entry the first
// to arrive writes it, the other is dropped (both are valid values for their keys).
__global__ __launch_bounds__(64) void cache_put_kernel(SearchParams p, const int32_t* __restrict__ n_leaves) {
  const int slot = blockIdx.x, lane = threadIdx.x;
  if (slot >= *n_leaves) return;
  const uint64_t hi = p.leaf_key[2 * slot], lo = p.leaf_key[2 * slot + 1];
  const size_t e = (size_t)(lo & ((1ull << p.c_bits) - 1));
  int old = 0;
  if (lane == 0) old = atomicExch(&p.c_writer[e], p.c_wave);
  old = __shfl(old, 0, 64);
  if (old == p.c_wave) return;
  if (lane == 0) {
    if (p.c_id[2 * e] == 0 && p.c_id[2 * e + 1] == 0) atomicAdd((unsigned long long*)&p.counters[10], 1ull);
    p.c_id[2 * e] = hi;
    p.c_id[2 * e + 1] = lo;
    p.c_value[e] = p.eval_value[slot];
  }
  const int A = p.num_actions;
  for (int i = lane; i < A; i += 64) p.c_probs[e * A + i] = p.eval_probs[(size_t)slot * A + i];
}

// One simulation wave of one game, one wavefront per game (Explorer.run_mcts, :49-61):
//   mode & 1  finish the pending expansion with the supplied evaluation (Explorer.py:162-181) and
//             back its value up           probs [n_leaves][A] float32 post-softmax, value [n_leaves]
//   mode & 2  simulate until the game waits on a leaf that needs an evaluation, its simulations
//             are used up, or `budget` simulations that ended in terminal leaves have run
// The rules, the scratch game and the legal-move mask live in LDS: the rules run on lane 0
// (serial, latency-bound code), the children are scored, the mask and the state image are built
// and the new children are written by all lanes.
__global__ __launch_bounds__(64) void wave_kernel(SearchParams p, int mode, const float* __restrict__ probs,
                                                   const float* __restrict__ value, float* __restrict__ images,
                                                   int32_t* __restrict__ leaf_game) {
  __shared__ __attribute__((aligned(16))) ScsRules R;
  __shared__ ScsState sc;
  __shared__ uint32_t smask[MASK_WORDS];
  __shared__ int sidx[MAXC_LIMIT];
  const int g = blockIdx.x;
  const int lane = lane_id();
#ifdef NZ_SCS_STAMPS     // diagnostic build: where a game's wave time goes (nz_scs_search_phase_ticks)
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, ts = __builtin_amdgcn_s_memtime();
#define NZ_STAMP(slot) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tk[slot] += now - ts; ts = now; }
#else
#define NZ_STAMP(slot)
#endif
  if (g == 0 && lane == 0 && p.clear_counters != nullptr) { p.clear_counters[0] = 0; p.clear_counters[1] = 0; p.clear_counters[2] = 0; }
  // everything this wave needs from the game's records in ONE round of loads (a chain of dependent loads, each
  // behind the branch on the previous one, costs a memory round trip per link)
  const uint32_t* m = p.leaf_mask + (size_t)g * MASK_WORDS;
  int32_t* path = p.path + (size_t)g * p.max_path;
  const int terminal = p.real[g].terminal;
  const int pend = p.pending[g];
  const int plen_in = p.path_len[g];
  const int base_in = p.node_count[g];
  const int sims_in = p.sims_left[g];
  const int root = p.root[g];
  uint32_t mword[(MASK_WORDS + 63) / 64];
#pragma unroll
  for (int w = 0; w < (MASK_WORDS + 63) / 64; ++w) mword[w] = w * 64 + lane < MASK_WORDS ? m[w * 64 + lane] : 0u;
  const int my_path0 = path[lane];                     // max_path > 64
  if (terminal) return;
  const int A = p.num_actions;
  SNode* nodes = arena(p, g);

  const bool expanding = (mode & 1) && pend >= 0;
  if (expanding) {
    const int slot = pend;
    const int plen = plen_in;
    const int leaf = plen <= 64 ? __shfl(my_path0, plen - 1, 64) : path[plen - 1];
    // second round: the evaluation and the path's statistics
    const double v = (double)value[slot];
    const int my_node = lane < plen ? my_path0 : 0;
    const int my_visit = lane < plen ? nodes[my_node].visit : 0;
    const double my_vs = lane < plen ? nodes[my_node].value_sum : 0.0;
    // legal actions in ascending order: lane w enumerates mask word w (and word 64 + w)
    int k = 0;
#pragma unroll
    for (int w = 0; w < (MASK_WORDS + 63) / 64; ++w) {
      uint32_t bits = mword[w];
      int inc = __popc(bits);
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      int off = k + inc - __popc(bits);
      while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        if (off < MAXC_LIMIT) sidx[off] = (w * 64 + lane) * 32 + b;
        ++off;
      }
      k += __shfl(inc, 63, 64);
    }
    const int base = base_in;
    const bool overflow = k > p.maxc;
    if (overflow || base + k > p.half_cap) {
      if (lane == 0) atomicOr(p.error_flag, overflow ? 16 : 1);
    } else {
      __syncthreads();
      int my_idx[MAXC_CHUNKS];                // child 64 c + lane: its action and its masked probability
      float my_val[MAXC_CHUNKS];
#pragma unroll
      for (int c = 0; c < MAXC_CHUNKS; ++c) {
        const int j = c * 64 + lane;
        my_idx[c] = j < k ? sidx[j] : 0x7fffffff;
        my_val[c] = j < k ? probs[(size_t)slot * A + my_idx[c]] : 0.0f;
      }
      float total = np_sum_sparse_f32_wave(p, my_idx, my_val, k);
      if (total == 0.0f) {                  // probs += mask (Explorer.py:171-173)
#pragma unroll
        for (int c = 0; c < MAXC_CHUNKS; ++c) my_val[c] = my_val[c] + 1.0f;
        total = np_sum_sparse_f32_wave(p, my_idx, my_val, k);
      }
#pragma unroll
      for (int c = 0; c < MAXC_CHUNKS; ++c) {
        const int j = c * 64 + lane;
        if (j < k) {
          SNode n;
          n.prior = (double)(my_val[c] / total);
          n.value_sum = 0.0; n.visit = 0; n.child_base = 0; n.n_children = 0; n.action = (uint16_t)my_idx[c];
          n.to_play = -1; n.prior_f64 = 0; n.terminal = 0; n.pad = 0;
          nodes[base + j] = n;
        }
      }
      if (lane == 0) {
        nodes[leaf].child_base = base;
        nodes[leaf].n_children = (uint16_t)k;
        p.node_count[g] = base + k;
      }
    }
    if (lane == 0) {
      p.pending[g] = -1;
      p.sims_left[g] = sims_in - 1;
    }
    // backup (Explorer.py:132-135): the path's statistics were read above; the leaf's new children do not touch them
    if (lane < plen) {
      nodes[my_node].visit = my_visit + 1;
      nodes[my_node].value_sum = my_vs + v;
    }
    for (int i = 64 + lane; i < plen; i += 64) {
      SNode& n = nodes[path[i]];
      n.visit += 1;
      n.value_sum = n.value_sum + v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  NZ_STAMP(0);                                  // expansion + backup of the evaluated leaf
  // The run's simulation / expansion counters take ONE atomic each per game and wave, as the wavefront leaves: a thousand
  // games add to the same two words, and an atomic still on its way holds up every later wait for a load (vmcnt counts
  // in order) -- the descent's first node reads stood behind the expansion's.
  const unsigned long long n_expanded = expanding ? 1ull : 0ull;
  auto leave = [&](unsigned long long sims_done) {
    if (lane == 0) {
      if (sims_done) atomicAdd((unsigned long long*)&p.counters[0], sims_done);
      if (n_expanded) atomicAdd((unsigned long long*)&p.counters[1], n_expanded);
    }
  };
  if (!(mode & 2) || (pend >= 0 && !expanding)) { leave(n_expanded); return; }
  int sims_left = expanding ? sims_in - 1 : sims_in;
  if (sims_left <= 0) { leave(n_expanded); return; }

  {   // rules -> LDS: 16 bytes per lane and load, all loads in flight before the first store
    static_assert(sizeof(ScsRules) % 4 == 0 && sizeof(ScsState) % 4 == 0, "copied as dwords");
    constexpr int N16 = (int)(sizeof(ScsRules) / 16), PER_LANE = (N16 + 63) / 64;
    const ScsRules* const my_rules = &rules_of(p, g);
    const uint4* src = reinterpret_cast<const uint4*>(my_rules);
    uint4* dst = reinterpret_cast<uint4*>(&R);
    uint4 tmp[PER_LANE];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j)
      if (j * 64 + lane < N16) tmp[j] = src[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j)
      if (j * 64 + lane < N16) dst[j * 64 + lane] = tmp[j];
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(my_rules);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(&R);
    for (int i = N16 * 4 + lane; i < (int)(sizeof(ScsRules) / 4); i += 64) d4[i] = s4[i];
  }
  NZ_STAMP(1);                                  // rules -> LDS
  long n_sim = 0;
  int budget = p.terminal_budget;
  bool queued = false;

  while (sims_left > 0 && budget > 0) {
    {   // scratch_game = game.shallow_clone()
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&p.real[g]);
      uint32_t* dst = reinterpret_cast<uint32_t*>(&sc);
      for (int i = lane; i < (int)(sizeof(ScsState) / 4); i += 64) dst[i] = src[i];
      __syncthreads();
    }
    NZ_STAMP(2);                                // scratch game = real game
    int node = root, plen = 1;
    if (lane == 0) path[0] = root;
    bool bad = false;
    // One dependent load per level: the chosen child's record was read when it was scored, so the lane that holds it
    // hands the four words the next level needs (visit count, first child, children count, to_play) to the wavefront
    // (v_readlane: the winner is wave-uniform) instead of every level re-reading its parent from memory.
    int par_visit, par_base, par_k, par_to_play;
    {
      const SNode r0 = nodes[root];
      par_visit = r0.visit; par_base = r0.child_base; par_k = r0.n_children; par_to_play = r0.to_play;
    }
    while (par_k > 0) {
      if (par_visit >= p.tab_len || plen >= p.max_path) { bad = true; break; }
      const double sq = p.sqrt_tab[par_visit], cb = p.bias_tab[par_visit];
      const bool negate = par_to_play == p.negate_player;
      double score = -INFINITY;
      int key = -1;
      int b_visit = 0, b_base = 0, b_k = 0, b_to_play = 0;        // this lane's best child so far
      for (int j = lane; j < par_k; j += 64) {                     // one chunk of 64 children per round, usually one
        const SNode c = nodes[par_base + j];
        const double sc = child_score(p, c, sq, cb, negate);
        const int ky = ((int)c.action << 8) | j;                   // j < 256; children are in ascending action order
        if (sc > score || (sc == score && ky > key)) {
          score = sc; key = ky;
          b_visit = c.visit; b_base = c.child_base; b_k = c.n_children; b_to_play = c.to_play;
        }
      }
      const int my_key = key;
      key = wave_argmax_key(score, key);          // max over (score, action): the larger action wins a tie (Explorer.py:100)
      // the lane that scored the winner (keys are unique: they carry the child index)
      const unsigned long long holders = __ballot(my_key == key);
      const int wl = __builtin_amdgcn_readfirstlane(__ffsll((long long)holders) - 1);
      node = par_base + (key & 0xff);
      par_visit = __builtin_amdgcn_readlane(b_visit, wl);
      par_base = __builtin_amdgcn_readlane(b_base, wl);
      par_k = __builtin_amdgcn_readlane(b_k, wl);
      par_to_play = __builtin_amdgcn_readlane(b_to_play, wl);
      if (lane == 0) path[plen] = node;
      scs_step_wave(R, sc, key >> 8, lane);
      ++plen;
    }
    if (bad) {
      if (lane == 0) atomicOr(p.error_flag, 2);
      sims_left = 0;
      break;
    }
    __syncthreads();                          // lane 0's steps on the scratch game are visible to every lane
    NZ_STAMP(3);                                // descent: scores, argmax, rule steps
    // evaluate (Explorer.py:137-181)
    const int to_play = sc.player, term = sc.terminal;
    if (term) {
      if (lane == 0) {
        nodes[node].to_play = (int8_t)to_play;
        nodes[node].terminal = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const double v = (double)sc.terminal_value;
      for (int i = lane; i < plen; i += 64) {
        SNode& n = nodes[path[i]];
        n.visit += 1;
        n.value_sum = n.value_sum + v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      --sims_left;
      --budget;
      ++n_sim;
      continue;
    }
    // leaf needs an evaluation: from the cache if its state is there (Explorer.py:147-149), else queue its state
    // image for the network; either way remember its legal actions for the expansion at the start of the next wave
    uint64_t key_hi = 0, key_lo = 0;
    bool hit = false;
    size_t entry = 0;
    if (p.c_bits > 0) {
      state_hash_wave(sc, lane, key_hi, key_lo);
      entry = (size_t)(key_lo & ((1ull << p.c_bits) - 1));
      hit = p.c_id[2 * entry] == key_hi && p.c_id[2 * entry + 1] == key_lo;      // uniform: every lane reads the same words
    }
    int slot = 0;
    if (lane == 0) {
      if (hit) {
        slot = p.c_hit_base + atomicAdd(p.hit_count, 1);
        atomicAdd((unsigned long long*)&p.counters[8], 1ull);
      } else {
        slot = atomicAdd(p.leaf_count, 1);
        leaf_game[slot] = g;
        if (p.c_bits > 0) {
          p.leaf_key[2 * slot] = key_hi;
          p.leaf_key[2 * slot + 1] = key_lo;
          atomicAdd((unsigned long long*)&p.counters[9], 1ull);
        }
      }
      p.pending[g] = slot;
      p.path_len[g] = plen;
      nodes[node].to_play = (int8_t)to_play;
    }
    slot = __shfl(slot, 0, 64);
#ifndef NZ_ABLATE_SCS_MASK    // timing experiment: no legal mask (results wrong)
    scs_legal_mask_wave<MASK_WORDS>(R, sc, smask, lane);
#endif
    uint32_t* m = p.leaf_mask + (size_t)g * MASK_WORDS;
    for (int i = lane; i < MASK_WORDS; i += 64) m[i] = smask[i];
    if (hit) {                                  // the cached evaluation takes the place of the network's
      for (int i = lane; i < A; i += 64) p.eval_probs[(size_t)slot * A + i] = p.c_probs[entry * A + i];
      if (lane == 0) p.eval_value[slot] = p.c_value[entry];
      queued = true;
      NZ_STAMP(4);
      break;
    }
#ifndef NZ_ABLATE_SCS_IMAGE   // timing experiment: no state image (results wrong)
    if (p.image_row_stride > 0)       // straight into the network's input rows: group of 16 slots, then cell, then slot
      scs_state_image_wave<true>(R, sc, images + ((size_t)(slot >> 4) * R.tiles * 16 + (slot & 15)) * p.image_row_stride,
                                 p.image_row_stride, lane);
    else
      scs_state_image_wave<false>(R, sc, images + (size_t)slot * R.channels * R.tiles, 0, lane);
#endif
    queued = true;
    NZ_STAMP(4);                                // legal mask + state image of the queued leaf
    break;
  }
  NZ_STAMP(5);                                  // terminal-leaf simulations
#ifdef NZ_SCS_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long*)&p.counters[2 + i], tk[i]);
#endif
  if (lane == 0) {
    p.sims_left[g] = sims_left;
    if (queued || sims_left > 0) atomicAdd(p.active_count, 1);
  }
  leave((unsigned long long)n_sim + n_expanded);
}


// ---- persistent self-play: one wavefront per game, the whole move's search in one launch --------------------------------
// wave_kernel + a network launch per simulation make every game wait for the slowest of all games, a thousand times a
// move.  Here a game's wavefront runs its simulations on its own (Explorer.run_mcts, :49-61: one in flight per tree):
// descent and rules as in wave_kernel, then the NETWORK FOR ITS OWN LEAF -- the split-bf16 arithmetic of
// fused16_net_kernel (boardnet.hip) for one position: 25 cells = two row tiles, every layer's weights streamed from L2,
// activations in the wavefront's own block of LDS -- softmax, expansion, backup, next simulation.  No barrier, no other
// game, no host until the move's simulations are used up.  Four games (wavefronts) per workgroup share the rules block.
struct PersistArgs {
  const Fused16Program* prog;
  int32_t net_floats, stage_off, stage_floats, inp, in_channels;
  int32_t wave_bytes;             // a game's LDS block: real game, scratch game, legal mask, legal list, flags, network
  int32_t rules_per_game;         // 1: every game slot keeps its own description in LDS (per-game maps), 0: one for the workgroup
  // leaf evaluations of chosen games, for the oracle replay of tests/scs_replay.py (null: none)
  const int32_t* rec_slot;        // [G] slot or -1
  int32_t rec_cap;
  int32_t* rec_count;             // [slots]
  uint64_t* rec_digest;           // [slots][cap][2] image_hash_wave of the leaf's float32 planes
  float* rec_probs;               // [slots][cap][A]
  float* rec_value;               // [slots][cap]
};
constexpr int PERSIST_GAMES = 4;                 // games per workgroup
constexpr int PERSIST_WAVES = 2 * PERSIST_GAMES;  // a leader and a helper wavefront per game
constexpr int PERSIST_THREADS = PERSIST_WAVES * 64;
constexpr int PERSIST_FLAG_BYTES = 32;           // per game: the step numbers of its (up to four) wavefronts, the pass number / exit word
constexpr int PERSIST_EXIT = 0x7fffffff;
constexpr int PERSIST_STATE_BYTES = (int)((sizeof(ScsState) + 15) / 16 * 16);
constexpr int PERSIST_MASK_BYTES = (MASK_WORDS * 4 + 15) / 16 * 16;
constexpr int PERSIST_GAME_BYTES = 2 * PERSIST_STATE_BYTES + PERSIST_MASK_BYTES + MAXC_LIMIT * 4 + PERSIST_FLAG_BYTES;
constexpr int PERSIST_RULES_BYTES = (int)((sizeof(ScsRules) + 15) / 16 * 16);
typedef const __attribute__((address_space(1))) u32x4* gptr4u;

// 128-bit digest of a leaf's float32 planes [channels][tiles] as they sit in the staging rows ([tile][inp]); order-free
// sums of per-element mixes, so any lane order gives the same value (tests/scs_replay.py image_mix_digest is its twin)
__device__ __forceinline__ void image_hash_wave(const float* stage, int inp, int channels, int tiles, int lane, uint64_t& hi, uint64_t& lo) {
  uint64_t a = 0, b = 0;
  for (int i = lane; i < channels * tiles; i += 64) {
    const int c = i / tiles, t = i - c * tiles;
    const uint64_t v = ((uint64_t)(uint32_t)i << 32) | __builtin_bit_cast(uint32_t, stage[t * inp + c]);
    a += mix64(v ^ 0x9e3779b97f4a7c15ull);
    b += mix64(v * 0xd6e8feb86659fd93ull + 0x2545f4914f6cdd1dull);
  }
  for (int o = 32; o; o >>= 1) {
    a += __shfl_xor((unsigned long long)a, o, 64);
    b += __shfl_xor((unsigned long long)b, o, 64);
  }
  hi = mix64(a ^ (b >> 7));
  lo = mix64(b ^ (a << 9));
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// the check word of a cache entry: per-lane sums of per-probability mixes (acc), the value and the key
__device__ __forceinline__ uint64_t cache_check_word(uint64_t acc, float value, uint64_t key_hi, uint64_t key_lo) {
  for (int o = 32; o; o >>= 1) acc += __shfl_xor((unsigned long long)acc, o, 64);
  return mix64(acc ^ mix64(key_hi ^ (uint64_t)__builtin_bit_cast(uint32_t, value)) ^ (key_lo * 0xd6e8feb86659fd93ull));
}

// One conv layer's K loop for one position and ONE column tile: both row tiles (25 cells), NTAPS x KGT steps as
// straight-line code.  Activations (the MFMA's second operand) come from LDS, this lane's operand row per tap in srow,
// read one step ahead of the MFMAs that use them; the weights (first operand) straight from the packed L2 stream, their
// loads running AHEAD steps in front (an L2 round trip is several steps long).
#ifndef NZ_PERSIST_AHEAD
#define NZ_PERSIST_AHEAD 2
#endif
constexpr int WAVE_AHEAD = NZ_PERSIST_AHEAD;
#ifndef NZ_PERSIST_SYNC_DRAIN
#define NZ_PERSIST_SYNC_DRAIN 0   // 1: a meeting waits for this wavefront's LDS stores before it posts its number
#endif
#ifndef NZ_PERSIST_KPRIO
#define NZ_PERSIST_KPRIO 0        // > 0: the K loops run at this wavefront priority
#endif
#ifndef NZ_PERSIST_SPREAD
#define NZ_PERSIST_SPREAD 1       // 1: a K step's loads placed one per MFMA gap (wave_conv)
#endif
// the first WAVE_AHEAD steps' weights of a column tile's stream (steps are contiguous whatever the layer's K groups)
// (buffer loads: a wave-uniform descriptor of the column tile's stream, ONE vector register with this lane's byte offset,
// the step's offset as a scalar -- plain global loads keep a 64-bit address pair alive per load in flight, forty registers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_weights_rsrc(const uint32_t* wg, int chunks16) {
  // (pinned to scalar registers: taken for lane-dependent, the descriptor makes every load of the stream a loop over
  // the lanes' values -- seen once, a fifth of the round)
  const uint64_t a = (uint64_t)(uintptr_t)wg;
  const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint32_t*>((uintptr_t)u), (short)0,
                                           __builtin_amdgcn_readfirstlane(chunks16 * 16), 0x00020000);
}
// one step's three pieces; the stream position is ONE running vector register (the step's 3 KB added after its loads, the
// pieces as immediate offsets): a scalar offset per (step, piece) was forty scalar registers of constants held across
// the whole pass, and the layer's own scalars went to spill lanes for them
__device__ __forceinline__ void wave_weights_step(u32x4 (&dst)[3], __amdgpu_buffer_rsrc_t rs, int& wv) {
#pragma unroll
  for (int piece = 0; piece < 3; ++piece) {
#ifdef NZ_ABL_PERSIST_NOB      // timing experiment: no weight stream (results wrong)
    dst[piece] = u32x4{(uint32_t)(wv & 0), (uint32_t)(piece & 0), 0u, 0u};       // (zeros: finite activations)
#else
    dst[piece] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, wv + piece * 1024, 0, 0));
#endif
  }
  wv += 3072;
  asm volatile("" : "+v"(wv));
}
__device__ __forceinline__ void wave_weights_prologue(u32x4 (&bq)[WAVE_AHEAD + 1][3], __amdgpu_buffer_rsrc_t rs, int& wv, int lane) {
  wv = lane * 16;
  asm volatile("" : "+v"(wv));
#pragma unroll
  for (int st = 0; st < WAVE_AHEAD; ++st) wave_weights_step(bq[st], rs, wv);
}
// Pieces buffers in a game's LDS block: [row][32-channel group][piece][four 16-byte chunks], a row every `cs` floats
// (48 per group, + 16 when the groups are even in number: rows 256 bytes apart would share their banks), chunk q of row
// r at position q ^ ((r >> 2) & 3).  A K step's operand reads are then ONE address per (row tile, tap) -- made once per
// layer -- plus immediates (group x 192 + piece x 64 bytes): no address arithmetic between the MFMAs, whose gaps hide
// eight cycles of other vector issue each and no more.
constexpr int PIECE_FLOATS = 16, GROUP_FLOATS = 3 * PIECE_FLOATS;
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4;
// this lane's operand rows of a position's conv taps, a byte per tap (four taps to a register): the row index (rows = the
// buffers' row of zeros) in rows[], the byte offset of this lane's chunk within a 32-channel group of that row in swz[]
template <int NTAPS>
struct TapRows {
  static constexpr int WORDS = (NTAPS + 3) / 4;
  uint32_t rows[WORDS], swz[WORDS];
  __device__ __forceinline__ uint32_t row(int tap) const { return (rows[tap >> 2] >> ((tap & 3) * 8)) & 0xffu; }
  __device__ __forceinline__ uint32_t chunk(int tap) const { return (swz[tap >> 2] >> ((tap & 3) * 8)) & 0xffu; }
};
template <bool HEX, int NTAPS>
__device__ __forceinline__ void wave_tap_rows(TapRows<NTAPS>& t, int row, int rows, int H, int Wd, int lane) {
  const bool row_ok = row < rows;
  const int cy = row / Wd, cx = row - cy * Wd;
#pragma unroll
  for (int w = 0; w < TapRows<NTAPS>::WORDS; ++w) { t.rows[w] = 0u; t.swz[w] = 0u; }
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap) {
    const int dy = HEX ? (tap < 3 ? tap - 1 : ((tap - 3) & 1) - 1 + (cx & 1)) : tap / 3 - 1;
    const int dx = HEX ? (tap < 3 ? 0 : (tap < 5 ? -1 : 1)) : tap % 3 - 1;
    const bool on = row_ok && (unsigned)(cy + dy) < (unsigned)H && (unsigned)(cx + dx) < (unsigned)Wd;
    const int r = on ? row + dy * Wd + dx : rows;       // (row index `rows`: every buffer's row of zeros)
    t.rows[tap >> 2] |= (uint32_t)r << ((tap & 3) * 8);
    t.swz[tap >> 2] |= (uint32_t)(((lane >> 4) ^ ((r >> 2) & 3)) << 4) << ((tap & 3) * 8);
  }
}

template <int NTAPS, int KGT, int RT = 2>
__device__ __forceinline__ void wave_conv(f32x4 (&acc)[RT], const TapRows<NTAPS> (&srow)[RT], uint32_t src, uint32_t cs4,
                                          __amdgpu_buffer_rsrc_t wrs, int& wv, u32x4 (&bq)[WAVE_AHEAD + 1][3],
                                          unsigned long long* tkn = nullptr, unsigned long long* tsn = nullptr) {
  constexpr int STEPS = NTAPS * KGT;
  constexpr int AHEAD = WAVE_AHEAD;
  static_assert(STEPS >= AHEAD, "the prologue is always AHEAD steps");
  u32x4 a[2][RT][3];
  // a tap's operand address: source buffer + row x row stride + this lane's chunk (LDS byte address); two taps' worth are
  // alive at a time
  uint32_t ad[2][RT];
  auto address = [&](int tap) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ad[tap & 1][rt] = __umul24(srow[rt].row(tap), cs4) + srow[rt].chunk(tap) + src;
  };
  auto read = [&](int st, int rt, int piece) -> u32x4 {
    const int tap = st / KGT, kg = st - tap * KGT;
#ifdef NZ_ABL_PERSIST_NOA    // timing experiment: no LDS operand reads (results wrong)
    return u32x4{(uint32_t)(ad[tap & 1][rt] & 0), (uint32_t)(piece & 0), (uint32_t)rt, 0u};   // (row tiles stay distinct chains)
#else
    return *reinterpret_cast<lds_u32x4>(ad[tap & 1][rt] + (uint32_t)((kg * GROUP_FLOATS + piece * PIECE_FLOATS) * 4));
#endif
  };
  address(0);
#pragma unroll
  for (int piece = 0; piece < 3; ++piece)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[0][rt][piece] = read(0, rt, piece);
  if (STEPS > 1 && 1 / KGT != 0) address(1 / KGT);
#ifdef NZ_PERSIST_HEADSTAMP   // diagnostic: how long a layer waits for its first operands (slot 3; perturbs the pipeline)
  if (tkn) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    tkn[3] += now - *tsn;
    *tsn = now;
  }
#endif
#if NZ_PERSIST_KPRIO
  __builtin_amdgcn_s_setprio(NZ_PERSIST_KPRIO);     // (matrix work before the SIMD's other wavefront's vector work)
#endif
#if NZ_PERSIST_SPREAD
  if constexpr (RT == 2) {
    // A step as six SLOTS, each one pair of MFMAs (row tile 0, row tile 1; step16's term order) with loads in its shadow
    // (an MFMA's gap hides eight cycles of other vector issue, no more): slots 0..2 the next step's operand reads in the
    // order the next step's pairs want them (pieces 1, 2, 0: every read has a whole step to arrive), slot 3 the weights of
    // AHEAD steps on, slot 4 the address of the tap after next.  Scheduling barriers between the slots: left to itself
    // the scheduler put the reads behind the step's MFMAs (their latency at the head of every step); given load slots
    // only, it ran each row tile's six MFMAs as one dependent chain.
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const bool more = st + 1 < STEPS;
      const u32x4 (&b)[3] = bq[st % (AHEAD + 1)];
      const u32x4 (&a0)[3] = a[st & 1][0];
      const u32x4 (&a1)[3] = a[st & 1][1];
      u32x4 (&n0)[3] = a[(st + 1) & 1][0];
      u32x4 (&n1)[3] = a[(st + 1) & 1][1];
      u32x4 (&nb)[3] = bq[(st + AHEAD) % (AHEAD + 1)];
#define NZ_PAIR(B, A)                         \
  acc[0] = wide_mfma(b[B], a0[A], acc[0]);    \
  acc[1] = wide_mfma(b[B], a1[A], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(1, 1)
      if (more) { n0[1] = read(st + 1, 0, 1); n1[1] = read(st + 1, 1, 1); }
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(0, 2)
      if (more) { n0[2] = read(st + 1, 0, 2); n1[2] = read(st + 1, 1, 2); }
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(2, 0)
      if (more) { n0[0] = read(st + 1, 0, 0); n1[0] = read(st + 1, 1, 0); }
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(0, 1)
      if (st + AHEAD < STEPS) wave_weights_step(nb, wrs, wv);
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(1, 0)
      if (st + 2 < STEPS && (st + 2) / KGT != (st + 1) / KGT) address((st + 2) / KGT);
      __builtin_amdgcn_sched_barrier(0);
      NZ_PAIR(0, 0)
      __builtin_amdgcn_sched_barrier(0);
#undef NZ_PAIR
    }
#if NZ_PERSIST_KPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return;
  }
#endif
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    if (st + 1 < STEPS) {
#pragma unroll
      for (int piece = 0; piece < 3; ++piece)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[(st + 1) & 1][rt][piece] = read(st + 1, rt, piece);
    }
    if (st + AHEAD < STEPS) wave_weights_step(bq[(st + AHEAD) % (AHEAD + 1)], wrs, wv);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) step16(acc[rt], a[st & 1][rt], bq[st % (AHEAD + 1)]);
    if (st + 2 < STEPS && (st + 2) / KGT != (st + 1) / KGT) address((st + 2) / KGT);
    __builtin_amdgcn_sched_barrier(0);          // (the scheduler would hoist every later step's loads up here: spills)
  }
#if NZ_PERSIST_KPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// float offset of (row, channel c0 = 16 ct + 4 (lane >> 4)) within a pieces buffer's row, piece 0
__device__ __forceinline__ int pieces_chunk(int ct, int lane, int row) {
  return (ct >> 1) * GROUP_FLOATS + (((((ct & 1) << 1) | (lane >> 5)) ^ ((row >> 2) & 3)) << 2) + ((lane >> 4) & 1) * 2;
}

// residual, activation, split into pieces (or float32 rows), store: fused16_net_kernel's epilogue for one tile
__device__ __forceinline__ void wave_epilogue(const f32x4& acc, float* __restrict__ net, const Fused16Op& op, int rt, int ct,
                                              int lane, int rows) {
  const int orow = rt * 16 + (lane & 15), c0 = ct * 16 + (lane >> 4) * 4;
  if (orow >= rows) return;
#ifdef NZ_ABL_PERSIST_NOEPI   // timing experiment: no epilogue (results wrong)
  if (acc[0] != 12345.678f) return;
#endif
  const int chunk = pieces_chunk(ct, lane, orow);
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc[r];
  if (op.offr >= 0) {
    const float* rp = net + op.offr + orow * op.csr + chunk;
    const uint2 q0 = *reinterpret_cast<const uint2*>(rp), q1 = *reinterpret_cast<const uint2*>(rp + PIECE_FLOATS),
                q2 = *reinterpret_cast<const uint2*>(rp + 2 * PIECE_FLOATS);
    auto lo = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
    auto hi = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); };
    v[0] += (lo(q0.x) + lo(q1.x)) + lo(q2.x);
    v[1] += (hi(q0.x) + hi(q1.x)) + hi(q2.x);
    v[2] += (lo(q0.y) + lo(q1.y)) + lo(q2.y);
    v[3] += (hi(q0.y) + hi(q1.y)) + hi(q2.y);
  }
  switch (op.act) {
    case 1:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      break;
    case 2:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      break;
    case 3:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : fast_expm1(v[r]);
      break;
    default: break;
  }
  if (op.psd == 0) {
    *reinterpret_cast<f32x4*>(net + op.offd + orow * op.csd + c0) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    uint16_t h3[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) split3_bits(v[r], h3[r]);
    float* dp = net + op.offd + orow * op.csd + chunk;
#pragma unroll
    for (int piece = 0; piece < 3; ++piece)
      *reinterpret_cast<uint2*>(dp + piece * PIECE_FLOATS) =
          uint2{(uint32_t)h3[0][piece] | ((uint32_t)h3[1][piece] << 16), (uint32_t)h3[2][piece] | ((uint32_t)h3[3][piece] << 16)};
  }
}

// wave_epilogue for both row tiles of a column tile at once: the eight values go through every step together (one
// wave-uniform switch, independent chains the scheduler interleaves; v_perm packs two values' upper halves per piece),
// the stores are predicated per row tile.  Same values as wave_epilogue, piece for piece.
__device__ __forceinline__ void wave_epilogue2(const f32x4 (&acc)[2], float* __restrict__ net, const Fused16Op& op, int ct,
                                               int lane, int rows) {
  const int c0 = ct * 16 + (lane >> 4) * 4;
#ifdef NZ_ABL_PERSIST_NOEPI   // timing experiment: no epilogue (results wrong)
  if (acc[0][0] != 12345.678f) return;
#endif
  int orow[2], chunk[2];
  bool ok[2];
  float v[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    orow[rt] = rt * 16 + (lane & 15);
    ok[rt] = orow[rt] < rows;
    chunk[rt] = pieces_chunk(ct, lane, orow[rt]);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[rt][r] = acc[rt][r];
  }
  if (op.offr >= 0) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      if (ok[rt]) {
        const float* rp = net + op.offr + orow[rt] * op.csr + chunk[rt];
        const uint2 q0 = *reinterpret_cast<const uint2*>(rp), q1 = *reinterpret_cast<const uint2*>(rp + PIECE_FLOATS),
                    q2 = *reinterpret_cast<const uint2*>(rp + 2 * PIECE_FLOATS);
        auto lo = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
        auto hi = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); };
        v[rt][0] += (lo(q0.x) + lo(q1.x)) + lo(q2.x);
        v[rt][1] += (hi(q0.x) + hi(q1.x)) + hi(q2.x);
        v[rt][2] += (lo(q0.y) + lo(q1.y)) + lo(q2.y);
        v[rt][3] += (hi(q0.y) + hi(q1.y)) + hi(q2.y);
      }
    }
  }
  switch (op.act) {
    case 1:
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i >> 2][i & 3] = v[i >> 2][i & 3] > 0.f ? v[i >> 2][i & 3] : 0.f;
      break;
    case 2:
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i >> 2][i & 3] = tanhf(v[i >> 2][i & 3]);
      break;
    case 3:
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i >> 2][i & 3] = v[i >> 2][i & 3] > 0.f ? v[i >> 2][i & 3] : fast_expm1(v[i >> 2][i & 3]);
      break;
    default: break;
  }
  if (op.psd == 0) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      if (ok[rt]) *reinterpret_cast<f32x4*>(net + op.offd + orow[rt] * op.csd + c0) = f32x4{v[rt][0], v[rt][1], v[rt][2], v[rt][3]};
    return;
  }
  uint2 q[2][3];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = v[rt][i];
#pragma unroll
    for (int piece = 0; piece < 3; ++piece) {
      q[rt][piece] = uint2{wide_pack_hi16(r[0], r[1]), wide_pack_hi16(r[2], r[3])};
      if (piece < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = r[i] - wide_trunc(r[i]);
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    if (ok[rt]) {
      float* dp = net + op.offd + orow[rt] * op.csd + chunk[rt];
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) *reinterpret_cast<uint2*>(dp + piece * PIECE_FLOATS) = q[rt][piece];
    }
  }
}

template <int NTAPS, int KGT, int RT = 2>
__device__ __forceinline__ void wave_layer_kloop(f32x4 (&acc)[RT], const float* __restrict__ net, const Fused16Op& op,
                                                 const TapRows<NTAPS> (&srow)[RT], int ct, int lane,
                                                 u32x4 (&bq)[WAVE_AHEAD + 1][3], int& wv, unsigned long long* tkn = nullptr,
                                                 unsigned long long* tsn = nullptr) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t src = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(net + op.off0);
  const uint32_t cs4 = (uint32_t)op.cs0 * 4u;
  const __amdgpu_buffer_rsrc_t wrs = wave_weights_rsrc(op.w + (size_t)ct * op.w_chunks * 4, op.w_chunks);
  wave_conv<NTAPS, KGT, RT>(acc, srow, src, cs4, wrs, wv, bq, tkn, tsn);
}

// The two wavefronts of a game meet (the leader runs the tree and half of every layer, the helper the other half): each
// posts the number of the step it has finished and waits for the other's.  LDS stores of one wavefront reach LDS in
// order, so the partner that sees the number sees the activations stored before it.
__device__ __forceinline__ void pair_sync(int* flags, int me, int& seq, int lane) {
  ++seq;
#if NZ_PERSIST_SYNC_DRAIN
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // (LDS only: everything a pair shares is in LDS)
#else
  // (no wait for the stores: the LDS takes one wavefront's operations in the order they were issued, so the number
  // lands behind the activations; the compiler is told not to move them)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif
  if (lane == 0) __hip_atomic_store(&flags[me], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef NZ_PERSIST_SYNC_SLEEP
#define NZ_PERSIST_SYNC_SLEEP 0
#endif
  while (__hip_atomic_load(&flags[me ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) {
    if (NZ_PERSIST_SYNC_SLEEP > 0) __builtin_amdgcn_s_sleep(NZ_PERSIST_SYNC_SLEEP);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// every layer of the network for the position whose input pieces sit in the game's block: this wavefront takes the
// column tiles half, half + 2, ... of each layer
#ifdef NZ_PERSIST_STAMPS
#define WSTAMP(slot) { if (tkn) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tkn[slot] += now - tsn; tsn = now; } }
#else
#define WSTAMP(slot)
#endif
// The per-pass constants of wave_network: the program's header words it needs (read once per move by the caller).
struct WaveNetArgs {
  const Fused16Program* prog;
  int n_ops, rows, H, Wd;
  int solo_at, solo_pol, solo_zero_off, solo_zero_len;
};
// Every layer of the network for the position whose input pieces sit in the game's block.  Trunk layers: this wavefront
// takes column tiles half, half + 2, ... of the layer's PAIRS of column tiles (both row tiles each), then row tile `half`
// of an odd last column tile (whole it was the leader's alone and the helper waited), and the pair meets after every
// layer.  Head layers (from solo_at on): the leader runs the policy head's chain, the helper the value head's, every tile
// of every layer, with no meeting -- the CALLER holds the last one (the leader after its softmax, which needs the
// logits only).
template <bool HEX>
__device__ __forceinline__ void wave_network(const WaveNetArgs& wn, float* __restrict__ net, int lane, int half, int* flags, int& seq,
                                             unsigned long long* tkn = nullptr) {
#ifdef NZ_PERSIST_STAMPS
  unsigned long long tsn = __builtin_amdgcn_s_memtime();
#endif
  constexpr int ntaps = HEX ? 7 : 9;
  const int n_ops = wn.n_ops, rows = wn.rows;
  TapRows<ntaps> srow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) wave_tap_rows<HEX, ntaps>(srow[rt], rt * 16 + (lane & 15), rows, wn.H, wn.Wd, lane);
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int OP_DWORDS = (int)(sizeof(Fused16Op) / 4);
  static_assert(OP_DWORDS <= 64, "one dword per lane");
  const gptr1u ops_words = (gptr1u)reinterpret_cast<const uint32_t*>(wn.prog->ops);
  auto unpack = [](uint32_t v, Fused16Op& op) {
    uint32_t words[OP_DWORDS];
#pragma unroll
    for (int i = 0; i < OP_DWORDS; ++i) words[i] = __builtin_amdgcn_readlane(v, i);
    __builtin_memcpy(&op, words, sizeof(Fused16Op));
  };
  // this wavefront's layers in order: the shared ones, then its own chain
  const int solo_at = wn.solo_at < n_ops ? wn.solo_at : n_ops;
  const int chain_first = half == 0 ? solo_at : solo_at + wn.solo_pol;
  const int chain_len = solo_at < n_ops ? (half == 0 ? wn.solo_pol : n_ops - solo_at - wn.solo_pol) : 0;
  const int n_mine = solo_at + chain_len;
  auto layer_at = [&](int i) { return i < solo_at ? i : chain_first + (i - solo_at); };
  // A layer's descriptor stays in ONE vector register (a dword per lane, fetched a layer ahead) until the layer starts:
  // unpacked a layer early its fields were scalar registers the K loop had no room for (spilled to lanes and back).
  uint32_t cur = (n_mine > 0 && lane < OP_DWORDS) ? ops_words[layer_at(0) * OP_DWORDS + lane] : 0u;
  uint32_t nxt = (n_mine > 1 && lane < OP_DWORDS) ? ops_words[layer_at(1) * OP_DWORDS + lane] : 0u;
  int wv = 0;
  u32x4 bq[WAVE_AHEAD + 1][3];
  auto job_weights = [&](const Fused16Op& x, int ct) {
    wave_weights_prologue(bq, wave_weights_rsrc(x.w + (size_t)ct * x.w_chunks * 4, x.w_chunks), wv, lane);
  };
  for (int i = 0; i < n_mine; ++i) {
    Fused16Op op;
    unpack(cur, op);
    cur = nxt;
    if (i + 2 < n_mine && lane < OP_DWORDS) nxt = ops_words[layer_at(i + 2) * OP_DWORDS + lane];
    const int kgt = op.kg0;
    const bool shared = i < solo_at;
    if (i == solo_at && half == 0 && lane * 4 < wn.solo_zero_len)       // (the policy head's hidden buffer: its row of zeros)
      *reinterpret_cast<f32x4*>(net + wn.solo_zero_off + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pair_tiles = shared ? op.ntiles & ~1 : op.ntiles;
    const bool odd = shared && (op.ntiles & 1);
    for (int ct = shared ? half : 0; ct < pair_tiles; ct += shared ? 2 : 1) {
      f32x4 acc[2];
      WSTAMP(4);                    // (the layer's header: descriptor words, addresses)
      job_weights(op, ct);
#ifdef NZ_PERSIST_STAMPS
      if (kgt == 1) wave_layer_kloop<ntaps, 1>(acc, net, op, srow, ct, lane, bq, wv, tkn, &tsn);
      else if (kgt == 2) wave_layer_kloop<ntaps, 2>(acc, net, op, srow, ct, lane, bq, wv, tkn, &tsn);
      else if (kgt == 3) wave_layer_kloop<ntaps, 3>(acc, net, op, srow, ct, lane, bq, wv, tkn, &tsn);
      else wave_layer_kloop<ntaps, 4>(acc, net, op, srow, ct, lane, bq, wv, tkn, &tsn);
#else
      if (kgt == 1) wave_layer_kloop<ntaps, 1>(acc, net, op, srow, ct, lane, bq, wv);
      else if (kgt == 2) wave_layer_kloop<ntaps, 2>(acc, net, op, srow, ct, lane, bq, wv);
      else if (kgt == 3) wave_layer_kloop<ntaps, 3>(acc, net, op, srow, ct, lane, bq, wv);
      else wave_layer_kloop<ntaps, 4>(acc, net, op, srow, ct, lane, bq, wv);
#endif
      WSTAMP(0);
      wave_epilogue2(acc, net, op, ct, lane, rows);
      WSTAMP(1);
    }
    if (odd) {
      const int ct = op.ntiles - 1;
      TapRows<ntaps> srow1[1];
#pragma unroll
      for (int w = 0; w < TapRows<ntaps>::WORDS; ++w) {
        srow1[0].rows[w] = half ? srow[1].rows[w] : srow[0].rows[w];
        srow1[0].swz[w] = half ? srow[1].swz[w] : srow[0].swz[w];
      }
      f32x4 acc[1];
      WSTAMP(4);
      job_weights(op, ct);
      if (kgt == 1) wave_layer_kloop<ntaps, 1, 1>(acc, net, op, srow1, ct, lane, bq, wv);
      else if (kgt == 2) wave_layer_kloop<ntaps, 2, 1>(acc, net, op, srow1, ct, lane, bq, wv);
      else if (kgt == 3) wave_layer_kloop<ntaps, 3, 1>(acc, net, op, srow1, ct, lane, bq, wv);
      else wave_layer_kloop<ntaps, 4, 1>(acc, net, op, srow1, ct, lane, bq, wv);
      WSTAMP(0);
      wave_epilogue(acc[0], net, op, half, ct, lane, rows);
      WSTAMP(1);
    }
    if (shared) pair_sync(flags, half, seq, lane);     // the layer is whole before either half reads it (or overwrites its source)
    else scs_sync<false>();                            // (this wavefront's own stores, then its own reads: LDS keeps their order)
    WSTAMP(2);
  }
}

// ---- diagnostic: FOUR wavefronts per game, one (row tile, column tile) each (netbench4_kernel; as a search kernel it
// lost to the pair: at 128 registers the leader's tree code spills) -----------------------------------------------------
__device__ __forceinline__ void quad_sync(int* flags, int me, int& seq, int lane) {
  ++seq;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&flags[me], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (int o = 0; o < 4; ++o)
    while (__hip_atomic_load(&flags[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool HEX>
__device__ __forceinline__ void quad_network(const Fused16Program* __restrict__ prog, float* __restrict__ net, int n_ops, int rows,
                                             int H, int Wd, int lane, int quad, int* flags, int& seq) {
  constexpr int ntaps = HEX ? 7 : 9;
  const int rt = quad & 1, half = quad >> 1;
  TapRows<ntaps> srow[1];
  wave_tap_rows<HEX, ntaps>(srow[0], rt * 16 + (lane & 15), rows, H, Wd, lane);
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int OP_DWORDS = (int)(sizeof(Fused16Op) / 4);
  const gptr1u ops_words = (gptr1u)reinterpret_cast<const uint32_t*>(prog->ops);
  uint32_t dvec = lane < OP_DWORDS ? ops_words[lane] : 0u;
  for (int o = 0; o < n_ops; ++o) {
    Fused16Op op;
    {
      uint32_t words[OP_DWORDS];
#pragma unroll
      for (int i = 0; i < OP_DWORDS; ++i) words[i] = __builtin_amdgcn_readlane(dvec, i);
      __builtin_memcpy(&op, words, sizeof(Fused16Op));
    }
    if (o + 1 < n_ops && lane < OP_DWORDS) dvec = ops_words[(o + 1) * OP_DWORDS + lane];
    const int kgt = op.kg0;
    for (int ct = half; ct < op.ntiles; ct += 2) {
      f32x4 acc[1];
      u32x4 bq[WAVE_AHEAD + 1][3];
      int wv = 0;
      wave_weights_prologue(bq, wave_weights_rsrc(op.w + (size_t)ct * op.w_chunks * 4, op.w_chunks), wv, lane);
      if (kgt == 1) wave_layer_kloop<ntaps, 1, 1>(acc, net, op, srow, ct, lane, bq, wv);
      else if (kgt == 2) wave_layer_kloop<ntaps, 2, 1>(acc, net, op, srow, ct, lane, bq, wv);
      else if (kgt == 3) wave_layer_kloop<ntaps, 3, 1>(acc, net, op, srow, ct, lane, bq, wv);
      else wave_layer_kloop<ntaps, 4, 1>(acc, net, op, srow, ct, lane, bq, wv);
      wave_epilogue(acc[0], net, op, rt, ct, lane, rows);
    }
    quad_sync(flags, quad, seq, lane);
  }
}
// The ascending list of a position's legal actions (Explorer.py:163-165): the mask in LDS, then lane w enumerates mask
// word w (and word 64 + w) behind a prefix sum of the words' bit counts.  Returns their number.
__device__ __forceinline__ int legal_list_wave(const ScsRules& R, const ScsState& sc, uint32_t* smask, int* sidx, int lane) {
  scs_legal_mask_wave<MASK_WORDS, false>(R, sc, smask, lane);
  int k = 0;
#pragma unroll
  for (int w = 0; w < (MASK_WORDS + 63) / 64; ++w) {
    uint32_t bits = w * 64 + lane < MASK_WORDS ? smask[w * 64 + lane] : 0u;
    int inc = __popc(bits);
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    int off = k + inc - __popc(bits);
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      if (off < MAXC_LIMIT) sidx[off] = (w * 64 + lane) * 32 + b;
      ++off;
    }
    k += __shfl(inc, 63, 64);
  }
  return k;
}
// This wavefront's half (`half` = 0 leader, 1 helper) of turning the staged float32 planes into the input pieces (every
// row of the input region, its row of zeros included) ...
__device__ __forceinline__ void split_planes_half(float* __restrict__ net, const PersistArgs& q, int hw, int in_off, int in_cs,
                                                  int lane, int half) {
  const float* const stage = net + q.stage_off;
  const int chunks = q.inp >> 3;
  for (int i = lane + 64 * half; i < (hw + 1) * chunks; i += 128) {
    const int r = i / chunks, c8 = i - r * chunks;
    u32x4 q0 = u32x4{0u, 0u, 0u, 0u}, q1 = q0, q2 = q0;
    if (r < hw) {
      f32x4 lo4 = *reinterpret_cast<const f32x4*>(stage + r * q.inp + c8 * 8);
      f32x4 hi4 = *reinterpret_cast<const f32x4*>(stage + r * q.inp + c8 * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c8 * 8 + j >= q.in_channels) lo4[j] = 0.f;
        if (c8 * 8 + 4 + j >= q.in_channels) hi4[j] = 0.f;
      }
      wide_split8(lo4, hi4, q0, q1, q2);
    }
    float* d = net + in_off + r * in_cs + (c8 >> 2) * GROUP_FLOATS + (((c8 & 3) ^ ((r >> 2) & 3)) << 2);
    *reinterpret_cast<u32x4*>(d) = q0;
    *reinterpret_cast<u32x4*>(d + PIECE_FLOATS) = q1;
    *reinterpret_cast<u32x4*>(d + 2 * PIECE_FLOATS) = q2;
  }
}
// ... and of zeroing the staging space again (it lies over the trunk buffers: their rows of zeros, channels no layer writes)
__device__ __forceinline__ void zero_stage_half(float* __restrict__ net, const PersistArgs& q, int lane, int half) {
  float* const stage = net + q.stage_off;
  for (int i = (lane + 64 * half) * 4; i < q.stage_floats; i += 512) *reinterpret_cast<f32x4*>(stage + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// the search of one game's move: the leader wavefront's body (returns when the move's simulations are used up)
template <bool HEX, int CH = MAXC_CHUNKS>
__device__ __forceinline__ void persist_leader(const SearchParams& p, const PersistArgs& q, const ScsRules& R, unsigned char* wb,
                                               int g, const int lane_in, int* flags, int* go) {
  int lane = lane_in;
  ScsState& real_l = *reinterpret_cast<ScsState*>(wb);
  ScsState& sc = *reinterpret_cast<ScsState*>(wb + PERSIST_STATE_BYTES);
  int* const sidx = reinterpret_cast<int*>(wb + 2 * PERSIST_STATE_BYTES + PERSIST_MASK_BYTES);
  float* const net = reinterpret_cast<float*>(wb + PERSIST_GAME_BYTES);
  int seq = 0, pass = 0;
  if (p.real[g].terminal) return;
  int sims_left = p.sims_left[g];
  if (sims_left <= 0) return;
  const int root = p.root[g];
  int base = p.node_count[g];
  int32_t* const path = p.path + (size_t)g * p.max_path;
  SNode* const nodes = arena(p, g);
  const int A = p.num_actions;
  {   // the real game and a clean network block
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&p.real[g]);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&real_l);
    for (int i = lane; i < (int)(sizeof(ScsState) / 4); i += 64) dst[i] = src[i];
    for (int i = lane * 4; i < q.net_floats; i += 256) *reinterpret_cast<f32x4*>(net + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int rec = q.rec_slot ? q.rec_slot[g] : -1;
  int rec_n = rec >= 0 ? q.rec_count[rec] : 0;
  // the program's header, one dword per lane
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int HDR_DWORDS = (int)(offsetof(Fused16Program, ops) / 4);
  static_assert(HDR_DWORDS <= 64, "one dword per lane");
  const uint32_t hdr_v = lane < HDR_DWORDS ? ((gptr1u)reinterpret_cast<const uint32_t*>(q.prog))[lane] : 0u;
#define PHDR(field) ((int)__builtin_amdgcn_readlane(hdr_v, (int)(offsetof(Fused16Program, field) / 4)))
  const int hw = PHDR(hw), H = PHDR(h), Wd = PHDR(wd), n_ops = PHDR(n_ops), planes = PHDR(planes);
  const int in_off = PHDR(in_off), in_cs = PHDR(in_cs);
  const int pol_off = PHDR(pol_off), pp = PHDR(pol_cs), val_off = PHDR(val_off), vp = PHDR(val_cs);
  const WaveNetArgs wna{q.prog, n_ops, hw, H, Wd, PHDR(solo_at), PHDR(solo_pol), PHDR(solo_zero_off), PHDR(solo_zero_len)};
#undef PHDR
  scs_sync<false>();
  int n_sim = 0, n_exp = 0, n_hit = 0, n_miss = 0;      // (of this move: 32 bits are plenty, and half the registers)
  bool failed = false;
  const uint32_t tiles_magic = 0xFFFFFFFFu / (uint32_t)hw + 1u;       // ceil(2^32 / tiles) for scs_step_wave (exact for every action index of boards of <= 100 tiles: checked exhaustively on the host, tests/test_host_logic.py)
#ifdef NZ_PERSIST_STAMPS   // diagnostic build: where a game's time goes (nz_scs_search_persist_ticks)
  unsigned long long tk[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = ts;
#define PSTAMP(slot) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tk[slot] += now - ts; ts = now; }
#else
#define PSTAMP(slot)
#endif

  while (sims_left > 0) {
    // (an opaque lane id per simulation: what is derived from it is a few cheap instructions, and carried around the
    // whole loop instead it is a dozen registers spilled to scratch and reloaded in the middle of the tree phase)
    lane = lane_in;
    asm volatile("" : "+v"(lane));
    {   // scratch_game = game.shallow_clone()
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&real_l);
      uint32_t* dst = reinterpret_cast<uint32_t*>(&sc);
      for (int i = lane; i < (int)(sizeof(ScsState) / 4); i += 64) dst[i] = src[i];
      scs_sync<false>();
    }
    PSTAMP(0);                                  // clone
    int node = root, plen = 1;
    int my_path = root;                         // lane i holds path[i] (levels past 63 go to memory)
    bool bad = false;
    int par_visit, par_base, par_k, par_to_play;
    {
      const SNode r0 = nodes[root];
      par_visit = r0.visit; par_base = r0.child_base; par_k = r0.n_children; par_to_play = r0.to_play;
    }
    // A level's children (its first 64) and its two table entries are fetched BEFORE the rule step that leads to it:
    // the round trip to L2 runs under the step's serial code instead of after it.
    SNode cnext;
    double sq = 0.0, cb = 0.0;
    auto fetch_level = [&]() {
      if (lane < par_k) cnext = nodes[par_base + lane];
      if (par_visit < p.tab_len) { sq = p.sqrt_tab[par_visit]; cb = p.bias_tab[par_visit]; }
    };
    fetch_level();
    while (par_k > 0) {
      if (par_visit >= p.tab_len || plen >= p.max_path) { bad = true; break; }
      const bool negate = par_to_play == p.negate_player;
      double score = -INFINITY;
      int key = -1;
      int b_visit = 0, b_base = 0, b_k = 0, b_to_play = 0;
      for (int j = lane; j < par_k; j += 64) {
        const SNode c = j < 64 ? cnext : nodes[par_base + j];
        const double scv = child_score(p, c, sq, cb, negate);
        const int ky = ((int)c.action << 8) | j;
        if (scv > score || (scv == score && ky > key)) {
          score = scv; key = ky;
          b_visit = c.visit; b_base = c.child_base; b_k = c.n_children; b_to_play = c.to_play;
        }
      }
      const int my_key = key;
      key = wave_argmax_key(score, key);
      const unsigned long long holders = __ballot(my_key == key);
      const int wl = __builtin_amdgcn_readfirstlane(__ffsll((long long)holders) - 1);
      node = par_base + (key & 0xff);
      par_visit = __builtin_amdgcn_readlane(b_visit, wl);
      par_base = __builtin_amdgcn_readlane(b_base, wl);
      par_k = __builtin_amdgcn_readlane(b_k, wl);
      par_to_play = __builtin_amdgcn_readlane(b_to_play, wl);
      if (par_k > 0) fetch_level();
      if (plen < 64) { if (lane == plen) my_path = node; }
      else if (lane == 0) path[plen] = node;
      scs_step_wave<false>(R, sc, key >> 8, lane, tiles_magic);
      ++plen;
    }
    if (bad) {
      if (lane == 0) atomicOr(p.error_flag, 2);
      failed = true;
      break;
    }
    scs_sync<false>();
    PSTAMP(1);                                  // descent: scores, argmax, rule steps
    const int to_play = sc.player, term = sc.terminal;
    double v;
    if (term) {                                 // Explorer.py:140-142: the stored terminal value, no network
      if (lane == 0) {
        nodes[node].to_play = (int8_t)to_play;
        nodes[node].terminal = 1;
      }
      v = (double)sc.terminal_value;
    } else {
      if (lane == 0) nodes[node].to_play = (int8_t)to_play;
      // The helper wavefront starts NOW on the leaf's legal mask and list (it is needed only at the expansion), while this
      // one looks the leaf up in the cache and stages its planes; they meet before the planes are split.
      ++pass;
      if (lane == 0) flags[7] = base;             // (what the helper's own overflow check needs)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // (the helper reads LDS only: no wait for the stores to the tree)
      if (lane == 0) __hip_atomic_store(go, pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef NZ_PERSIST_STAMP_GO    // diagnostic of a diagnostic: the leaf's bookkeeping up to the helper's wake-up, reported in the "slowest" slot
      PSTAMP(9);
#endif
      float* const pol = net + pol_off;
      // (action i = plane * hw + cell; lanes walk i = lane, lane + 64, ... without a division per entry)
      const int cell0 = lane % hw, plane0 = lane / hw, dcell = 64 % hw, dplane = 64 / hw;
      float sum = 1.0f, value = 0.f;
      // The inference cache (Explorer.py:146-155): the table is shared by every game of the engine and, on this route, read
      // and written while the kernel runs -- by games on all eight XCDs, so every access is an agent-scope atomic (past this
      // CU's L1 and this XCD's L2) and there are no fences (agent-scope fences write back / invalidate whole caches: 817
      // games/s with them against 970 without a cache).  Entries validate themselves: next to the state's 128-bit key an
      // entry holds a 64-bit check word over (key, probabilities, value); a reader takes everything in ONE round of loads
      // and accepts it only if the key is its leaf's and the check word fits what it read -- an entry caught half
      // written, or written by two games at once, fails and counts as a miss.  Writers just store (newest wins,
      // KeylessCache.py:60-75).  A hit is the evaluation the network made for the same state: the search cannot tell.
      uint64_t key_hi = 0, key_lo = 0;
      size_t entry = 0;
      bool hit = false;
      if (p.c_bits > 0) {
        state_hash_wave(sc, lane, key_hi, key_lo);
        key_hi = uniform64(key_hi); key_lo = uniform64(key_lo);         // (the same on every lane: scalar registers)
        entry = (size_t)(key_lo & ((1ull << p.c_bits) - 1));
        unsigned long long* const id = reinterpret_cast<unsigned long long*>(p.c_id + 2 * entry);
        const unsigned long long id0 = __hip_atomic_load(id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long id1 = __hip_atomic_load(id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long chk = __hip_atomic_load(reinterpret_cast<unsigned long long*>(p.c_check + entry), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float cval = __hip_atomic_load(p.c_value + entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t acc = 0;
        {   // (all of a block's loads in flight together: one round trip to memory, not one per probability)
          constexpr int PER = 12;
          int cell = cell0, plane = plane0;
          for (int i0 = 0; i0 < A; i0 += 64 * PER) {
            float r[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
              const int i = i0 + j * 64 + lane;
              r[j] = i < A ? __hip_atomic_load(p.c_probs + entry * A + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
              const int i = i0 + j * 64 + lane;
              if (i < A) {
                pol[cell * pp + plane] = r[j];
                acc += mix64((((uint64_t)(uint32_t)i << 32) | __builtin_bit_cast(uint32_t, r[j])) ^ 0x9e3779b97f4a7c15ull);
              }
              cell += dcell; plane += dplane;
              if (cell >= hw) { cell -= hw; ++plane; }
            }
          }
        }
        hit = id0 == key_hi && id1 == key_lo && cache_check_word(acc, cval, key_hi, key_lo) == chk;     // (uniform)
        if (hit) { value = cval; ++n_hit; } else ++n_miss;
        scs_sync<false>();
      }
      uint64_t dig_hi = 0, dig_lo = 0;
      if (hit && rec >= 0) {                    // (the test hook records every evaluation the search consumed: the planes' digest)
        float* const stage = net + q.stage_off;
        scs_state_image_wave<true, true>(R, sc, stage, q.inp >> 4, lane);
        scs_sync<false>();
        image_hash_wave(stage, q.inp, q.in_channels, hw, lane, dig_hi, dig_lo);
        scs_sync<false>();
        for (int i = lane * 4; i < q.stage_floats; i += 256) *reinterpret_cast<f32x4*>(stage + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        scs_sync<false>();
      }
      if (!hit) {
        // the leaf's planes (generate_network_input, SCS_Game.py:1507) as float32 rows over the trunk buffers' space
        float* const stage = net + q.stage_off;
        scs_state_image_wave<true, true>(R, sc, stage, q.inp >> 4, lane);
        scs_sync<false>();
        if (rec >= 0) image_hash_wave(stage, q.inp, q.in_channels, hw, lane, dig_hi, dig_lo);
      }
      PSTAMP(2);                                // cache lookup + planes (the helper: legal mask + list)
      if (lane == 0) flags[5] = hit ? 0 : 1;      // what the helper does after the meeting: nothing more / its half of the pass
      pair_sync(flags, 0, seq, lane);
      const int k = flags[6];                     // the helper's count of legal actions (the list is in sidx)
      {
        const bool overflow = k > p.maxc;
        if (overflow || base + k > p.half_cap) {  // (the helper makes the same check and stops here too)
          if (lane == 0) atomicOr(p.error_flag, overflow ? 16 : 1);
          failed = true;
          break;
        }
      }
      if (!hit) {
      split_planes_half(net, q, hw, in_off, in_cs, lane, 0);
      scs_sync<false>();
      pair_sync(flags, 0, seq, lane);             // (both halves of the pieces are made: the staging space can go)
      zero_stage_half(net, q, lane, 0);
      scs_sync<false>();
      pair_sync(flags, 0, seq, lane);
      PSTAMP(3);                                // split, staging cleared (halves)
#ifdef NZ_PERSIST_STAMPS
      wave_network<HEX>(wna, net, lane, 0, flags, seq, tk + 10);
#else
      wave_network<HEX>(wna, net, lane, 0, flags, seq);
#endif
      PSTAMP(4);                                // network

      // softmax over ALL logits (Explorer.py:158-160) and value = tanh(mean of the value plane) (blocks.py:82-84)
      float mx = -INFINITY;
      for (int i = lane, cell = cell0, plane = plane0; i < A; i += 64) {
        mx = fmaxf(mx, pol[cell * pp + plane]);
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
      for (int w = 32; w; w >>= 1) mx = fmaxf(mx, __shfl_xor(mx, w, 64));
      // the sum in fused16_net_kernel's order for a workgroup of up to four positions (four wavefronts per position, each
      // lane adding every 256th entry, a butterfly per wavefront, the four partial sums added in turn), so that a game
      // plays the same moves on either route
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      {
        int cell = cell0, plane = plane0;
        for (int i0 = 0; i0 < A; i0 += 256) {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            const int i = i0 + sub * 64 + lane;
            if (i < A) {
              const float e = expf(pol[cell * pp + plane] - mx);
              pol[cell * pp + plane] = e;
              part[sub] += e;
            }
            cell += dcell; plane += dplane;
            if (cell >= hw) { cell -= hw; ++plane; }
          }
        }
      }
      sum = 0.f;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        float ps = part[sub];
        for (int w = 32; w; w >>= 1) ps += __shfl_xor(ps, w, 64);
        sum += ps;
      }
      pair_sync(flags, 0, seq, lane);             // (the helper's value head is done)
      float sv = lane < hw ? net[val_off + lane * vp] : 0.f;
      for (int w = 32; w; w >>= 1) sv += __shfl_xor(sv, w, 64);
      value = tanhf(sv / (float)hw);
      (void)planes;
      scs_sync<false>();
      PSTAMP(5);                                // softmax + value
      if (p.c_bits > 0) {
        // KeylessCache.put (KeylessCache.py:60-75): the newest evaluation takes the entry; stores only, nothing to wait for
        unsigned long long* const id = reinterpret_cast<unsigned long long*>(p.c_id + 2 * entry);
        uint64_t acc = 0;
        for (int i = lane, cell = cell0, plane = plane0; i < A; i += 64) {
          const float pr = pol[cell * pp + plane] / sum;
          __hip_atomic_store(p.c_probs + entry * A + i, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc += mix64((((uint64_t)(uint32_t)i << 32) | __builtin_bit_cast(uint32_t, pr)) ^ 0x9e3779b97f4a7c15ull);
          cell += dcell; plane += dplane;
          if (cell >= hw) { cell -= hw; ++plane; }
        }
        const unsigned long long chk = cache_check_word(acc, value, key_hi, key_lo);
        if (lane == 0) {
          __hip_atomic_store(p.c_value + entry, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(id, (unsigned long long)key_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(id + 1, (unsigned long long)key_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.c_check + entry), chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      }   // (!hit)
      v = (double)value;
      if (rec >= 0) {
        if (rec_n < q.rec_cap) {
          const size_t at = (size_t)rec * q.rec_cap + rec_n;
          for (int i = lane; i < A; i += 64) {
            const int plane = i / hw, cell = i - plane * hw;
            q.rec_probs[at * A + i] = pol[cell * pp + plane] / sum;
          }
          if (lane == 0) { q.rec_value[at] = value; q.rec_digest[2 * at] = dig_hi; q.rec_digest[2 * at + 1] = dig_lo; }
        }
        ++rec_n;
      }
      // expansion (Explorer.py:162-181): child j = the j-th legal action, prior = masked prob / their float32 np.sum
      int my_idx[CH];                         // (CH = 1: this game has at most 64 legal actions in a position)
      float my_val[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 64 + lane;
        my_idx[c] = j < k ? sidx[j] : 0x7fffffff;
        float pr = 0.0f;
        if (j < k) {
          const int plane = my_idx[c] / hw, cell = my_idx[c] - plane * hw;
          pr = pol[cell * pp + plane] / sum;
        }
        my_val[c] = pr;
      }
      float total = np_sum_sparse_f32_wave<CH>(p, my_idx, my_val, k);
      if (total == 0.0f) {
#pragma unroll
        for (int c = 0; c < CH; ++c) my_val[c] = my_val[c] + 1.0f;
        total = np_sum_sparse_f32_wave<CH>(p, my_idx, my_val, k);
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 64 + lane;
        if (j < k) {
          SNode n;
          n.prior = (double)(my_val[c] / total);
          n.value_sum = 0.0; n.visit = 0; n.child_base = 0; n.n_children = 0; n.action = (uint16_t)my_idx[c];
          n.to_play = -1; n.prior_f64 = 0; n.terminal = 0; n.pad = 0;
          nodes[base + j] = n;
        }
      }
      if (lane == 0) {
        nodes[node].child_base = base;
        nodes[node].n_children = (uint16_t)k;
      }
      base += k;
      ++n_exp;
      PSTAMP(6);                                // expansion
    }
    // backup (Explorer.py:132-135): the same value to every node of the path, root included
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < plen && lane < 64) {
      SNode& n = nodes[my_path];
      n.visit += 1;
      n.value_sum = n.value_sum + v;
    }
    for (int i = 64 + lane; i < plen; i += 64) {
      SNode& n = nodes[path[i]];
      n.visit += 1;
      n.value_sum = n.value_sum + v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    --sims_left;
    ++n_sim;
    PSTAMP(7);                                  // backup
  }
#ifdef NZ_PERSIST_STAMPS
  if (lane == 0) {
    tk[8] = __builtin_amdgcn_s_memtime() - t_begin;        // the game's whole move
    for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long*)&p.counters[2 + i], tk[i]);
    for (int i = 6; i < 9; ++i) atomicAdd((unsigned long long*)&p.counters[5 + i], tk[i]);
#ifdef NZ_PERSIST_STAMP_GO
    atomicAdd((unsigned long long*)&p.counters[14], tk[9]);
#else
    atomicMax((unsigned long long*)&p.counters[14], tk[8]);  // the slowest (game, move) of the round
#endif
    // inside the network (the leader's half): K loops, epilogues, waiting for the helper -- in the cache counters' words
    // (a diagnostic build runs without the cache)
    for (int i = 0; i < 3; ++i) atomicAdd((unsigned long long*)&p.counters[8 + i], tk[10 + i]);
  }
#endif
  if (lane == 0) {
    p.sims_left[g] = failed ? sims_left : 0;
    p.pending[g] = -1;
    p.node_count[g] = base;
    if (rec >= 0) q.rec_count[rec] = rec_n;
    if (n_sim) atomicAdd((unsigned long long*)&p.counters[0], (unsigned long long)n_sim);
    if (n_exp) atomicAdd((unsigned long long*)&p.counters[1], (unsigned long long)n_exp);
    if (n_hit) atomicAdd((unsigned long long*)&p.counters[8], (unsigned long long)n_hit);
    if (n_miss) atomicAdd((unsigned long long*)&p.counters[9], (unsigned long long)n_miss);
  }
}

template <bool HEX, int CH = MAXC_CHUNKS>
__global__ __launch_bounds__(PERSIST_THREADS) void persist_kernel(SearchParams p, PersistArgs q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // wavefronts 0..3 lead games 0..3; wavefront 4 + i helps game (i + 3) & 3, so that every SIMD hosts one leader and
  // one helper of ANOTHER game (a helper sleeps through its game's tree phases)
  const bool leader = wave < PERSIST_GAMES;
  const int slot = leader ? wave : ((wave - PERSIST_GAMES + 3) & (PERSIST_GAMES - 1));
  const int n_rules = q.rules_per_game ? PERSIST_GAMES : 1;       // descriptions in LDS: one, or one per game slot (its own map)
  const ScsRules& R = *reinterpret_cast<const ScsRules*>(smem + (size_t)(q.rules_per_game ? slot : 0) * PERSIST_RULES_BYTES);
  {   // rules -> LDS, once per workgroup, and the games' flag words; the only workgroup barrier of the kernel
    static_assert(sizeof(ScsRules) % 4 == 0 && sizeof(ScsState) % 4 == 0, "copied as dwords");
    for (int r = 0; r < n_rules; ++r) {
      const int gr = blockIdx.x * PERSIST_GAMES + r;
      if (gr >= p.n_games) break;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&rules_of(p, gr));
      uint32_t* dst = reinterpret_cast<uint32_t*>(smem + (size_t)r * PERSIST_RULES_BYTES);
      for (int i = threadIdx.x; i < (int)(sizeof(ScsRules) / 4); i += PERSIST_THREADS) dst[i] = src[i];
    }
    if (threadIdx.x < PERSIST_GAMES * 8)
      reinterpret_cast<int*>(smem + (size_t)n_rules * PERSIST_RULES_BYTES + (size_t)(threadIdx.x >> 3) * q.wave_bytes + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES)[threadIdx.x & 7] = 0;
  }
  __syncthreads();
  unsigned char* const wb = smem + (size_t)n_rules * PERSIST_RULES_BYTES + (size_t)slot * q.wave_bytes;
  int* const flags = reinterpret_cast<int*>(wb + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES);
  int* const go = flags + 4;
  const int g = blockIdx.x * PERSIST_GAMES + slot;
  if (leader) {
    if (g < p.n_games) persist_leader<HEX, CH>(p, q, R, wb, g, lane, flags, go);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(go, PERSIST_EXIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  // helper: half of every layer of every network pass of its game, asleep in between
  float* const net = reinterpret_cast<float*>(wb + PERSIST_GAME_BYTES);
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int HDR_DWORDS = (int)(offsetof(Fused16Program, ops) / 4);
  const uint32_t hdr_v = lane < HDR_DWORDS ? ((gptr1u)reinterpret_cast<const uint32_t*>(q.prog))[lane] : 0u;
#define PHDR(field) ((int)__builtin_amdgcn_readlane(hdr_v, (int)(offsetof(Fused16Program, field) / 4)))
  const int hw = PHDR(hw), H = PHDR(h), Wd = PHDR(wd), n_ops = PHDR(n_ops);
  const int in_off = PHDR(in_off), in_cs = PHDR(in_cs);
  const WaveNetArgs wna{q.prog, n_ops, hw, H, Wd, PHDR(solo_at), PHDR(solo_pol), PHDR(solo_zero_off), PHDR(solo_zero_len)};
#undef PHDR
  const ScsState& sc = *reinterpret_cast<const ScsState*>(wb + PERSIST_STATE_BYTES);
  uint32_t* const smask = reinterpret_cast<uint32_t*>(wb + 2 * PERSIST_STATE_BYTES);
  int* const sidx = reinterpret_cast<int*>(wb + 2 * PERSIST_STATE_BYTES + PERSIST_MASK_BYTES);
  int seq = 0, pass = 0;
  for (;;) {
    int gv;
#ifndef NZ_PERSIST_GO_SLEEP
#define NZ_PERSIST_GO_SLEEP 2
#endif
    while ((gv = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == pass) __builtin_amdgcn_s_sleep(NZ_PERSIST_GO_SLEEP);
    if (gv == PERSIST_EXIT) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    pass = gv;
    // a leaf that needs an evaluation: its legal mask and list (the leader meanwhile looks it up in the cache and stages
    // its planes), then -- unless the cache had it -- half of the split into pieces and half of every layer
    const int k = legal_list_wave(R, sc, smask, sidx, lane);
    if (lane == 0) flags[6] = k;
    pair_sync(flags, 1, seq, lane);
    const int job = flags[5], base = flags[7];
    if (job == 0 || k > p.maxc || base + k > p.half_cap) continue;     // (a hit, or the leader is raising the overflow flag)
    split_planes_half(net, q, hw, in_off, in_cs, lane, 1);
    scs_sync<false>();
    pair_sync(flags, 1, seq, lane);
    zero_stage_half(net, q, lane, 1);
    scs_sync<false>();
    pair_sync(flags, 1, seq, lane);
    wave_network<HEX>(wna, net, lane, 1, flags, seq);
    pair_sync(flags, 1, seq, lane);               // (the value plane is whole; the leader has done its softmax meanwhile)
  }
}

template <bool HEX>
__global__ __launch_bounds__(PERSIST_GAMES * 4 * 64) void netbench4_kernel(PersistArgs q, int iters, unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = wave >> 2, quad = wave & 3;                // a game's four wavefronts sit on the four SIMDs
  if (threadIdx.x < PERSIST_GAMES * 8)
    reinterpret_cast<int*>(smem + PERSIST_RULES_BYTES + (size_t)(threadIdx.x >> 3) * q.wave_bytes + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES)[threadIdx.x & 7] = 0;
  __syncthreads();
  unsigned char* const wb = smem + PERSIST_RULES_BYTES + (size_t)slot * q.wave_bytes;
  int* const flags = reinterpret_cast<int*>(wb + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES);
  float* const net = reinterpret_cast<float*>(wb + PERSIST_GAME_BYTES);
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int HDR_DWORDS = (int)(offsetof(Fused16Program, ops) / 4);
  const uint32_t hdr_v = lane < HDR_DWORDS ? ((gptr1u)reinterpret_cast<const uint32_t*>(q.prog))[lane] : 0u;
#define PHDR(field) ((int)__builtin_amdgcn_readlane(hdr_v, (int)(offsetof(Fused16Program, field) / 4)))
  const int hw = PHDR(hw), H = PHDR(h), Wd = PHDR(wd), n_ops = PHDR(n_ops);
#undef PHDR
  if (quad == 0)
    for (int i = lane * 4; i < q.net_floats; i += 256) *reinterpret_cast<f32x4*>(net + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  int seq = 0;
  quad_sync(flags, quad, seq, lane);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int active = iters >> 16 ? iters >> 16 : PERSIST_GAMES, n_it = iters & 0xffff;
  if (slot < active)
    for (int it = 0; it < n_it; ++it) quad_network<HEX>(q.prog, net, n_ops, hw, H, Wd, lane, quad, flags, seq);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (quad == 0 && lane == 0) out[blockIdx.x * PERSIST_GAMES + slot] = (t1 - t0) / (unsigned long long)n_it;
}

// Diagnostic: the network part of persist_kernel alone -- every game slot of every workgroup runs `iters` passes on an
// all-zero input (leader + helper exactly as in the search), out[block * PERSIST_GAMES + slot] = shader ticks per pass.
template <bool HEX>
__global__ __launch_bounds__(PERSIST_THREADS) void netbench_kernel(PersistArgs q, int iters, unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool leader = wave < PERSIST_GAMES;
  const int slot = leader ? wave : ((wave - PERSIST_GAMES + 3) & (PERSIST_GAMES - 1));
  if (threadIdx.x < PERSIST_GAMES * 8)
    reinterpret_cast<int*>(smem + PERSIST_RULES_BYTES + (size_t)(threadIdx.x >> 3) * q.wave_bytes + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES)[threadIdx.x & 7] = 0;
  __syncthreads();
  unsigned char* const wb = smem + PERSIST_RULES_BYTES + (size_t)slot * q.wave_bytes;
  int* const flags = reinterpret_cast<int*>(wb + PERSIST_GAME_BYTES - PERSIST_FLAG_BYTES);
  float* const net = reinterpret_cast<float*>(wb + PERSIST_GAME_BYTES);
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int HDR_DWORDS = (int)(offsetof(Fused16Program, ops) / 4);
  const uint32_t hdr_v = lane < HDR_DWORDS ? ((gptr1u)reinterpret_cast<const uint32_t*>(q.prog))[lane] : 0u;
#define PHDR(field) ((int)__builtin_amdgcn_readlane(hdr_v, (int)(offsetof(Fused16Program, field) / 4)))
  const int hw = PHDR(hw), H = PHDR(h), Wd = PHDR(wd), n_ops = PHDR(n_ops);
  const WaveNetArgs wna{q.prog, n_ops, hw, H, Wd, PHDR(solo_at), PHDR(solo_pol), PHDR(solo_zero_off), PHDR(solo_zero_len)};
#undef PHDR
  if (leader)
    for (int i = lane * 4; i < q.net_floats; i += 256) *reinterpret_cast<f32x4*>(net + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  int seq = 0;
  pair_sync(flags, leader ? 0 : 1, seq, lane);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#ifdef NZ_PERSIST_STAMPS    // diagnostic build: iters bits 24.. = 1 + the phase whose ticks are reported instead of the pass
  unsigned long long tkn[5] = {0, 0, 0, 0, 0};   // K loops, epilogues, meetings, first operands (HEADSTAMP), layer headers
  const int phase = (iters >> 24) - 1;
  const int active = (iters >> 16) & 0xff ? (iters >> 16) & 0xff : PERSIST_GAMES, n_it = iters & 0xffff;
  if (slot < active)
    for (int it = 0; it < n_it; ++it) {
      wave_network<HEX>(wna, net, lane, leader ? 0 : 1, flags, seq, tkn);
      pair_sync(flags, leader ? 0 : 1, seq, lane);
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long rep = t1 - t0;
  for (int k = 0; k < 5; ++k) if (phase == k) rep = tkn[k];
  if (phase >= 8 && phase < 13) {              // phases 8..12: the HELPER's
    if (!leader && lane == 0) out[blockIdx.x * PERSIST_GAMES + slot] = tkn[phase - 8] / (unsigned long long)n_it;
  } else if (leader && lane == 0) out[blockIdx.x * PERSIST_GAMES + slot] = rep / (unsigned long long)n_it;
#else
  const int active = (iters >> 16) & 0xff ? (iters >> 16) & 0xff : PERSIST_GAMES, n_it = iters & 0xffff;
  if (slot < active)
    for (int it = 0; it < n_it; ++it) {
      wave_network<HEX>(wna, net, lane, leader ? 0 : 1, flags, seq);
      pair_sync(flags, leader ? 0 : 1, seq, lane);
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (leader && lane == 0) out[blockIdx.x * PERSIST_GAMES + slot] = (t1 - t0) / (unsigned long long)n_it;
#endif
}

__global__ void search_status_kernel(SearchParams p, int32_t* out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  const ScsState& s = p.real[g];
  int32_t* o = out + g * 7;
  o[0] = s.player; o[1] = s.sub_phase; o[2] = s.stage; o[3] = s.turn; o[4] = s.terminal; o[5] = s.terminal_value;
  o[6] = s.length;
}

// select_action, records, step and re-rooting (Explorer.py:70-97,183-199; Gamer.py:71-79)
// `forced` (may be null): per game an action to play instead of the search's own choice, -1 = own choice -- the
// opponent's move in MctsAgent.update_subtree (Testing/Agents/Generic/MctsAgent.py:35-39)
__global__ void end_move_kernel(SearchParams p, const double* __restrict__ uniforms, const int32_t* __restrict__ forced) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  ScsState& real = p.real[g];
  if (real.terminal) return;
  const ScsRules& R = rules_of(p, g);
  SNode* nodes = arena(p, g);
  const SNode root = nodes[p.root[g]];
  const int k = root.n_children, move = real.length;
  if (k == 0 || p.sims_left[g] != 0 || p.pending[g] >= 0 || move >= p.max_moves) {
    atomicOr(p.error_flag, move >= p.max_moves ? 32 : 4);
    return;
  }
  const size_t gm = (size_t)g * p.max_moves + move;
  const int MAXC = p.maxc;
  int best = 0;
  for (int j = 0; j < k; ++j) {
    const SNode& c = nodes[root.child_base + j];
    p.rec_child_action[gm * MAXC + j] = c.action;
    p.rec_child_visit[gm * MAXC + j] = c.visit;
    p.rec_child_prior[gm * MAXC + j] = c.prior;
    p.rec_child_value_sum[gm * MAXC + j] = c.value_sum;
    if (c.visit > nodes[root.child_base + best].visit) best = j;      // max_action: first maximum
  }
  p.rec_tree_size[gm] = root.visit;
  p.rec_children[gm] = k;
  p.rec_bias[gm] = root.visit < p.tab_len ? p.bias_tab[root.visit] : 0.0;
  p.rec_root_value_sum[gm] = root.value_sum;

  int mode = 0;
  double u3 = 0.0;
  const int forced_action = forced ? forced[g] : -1;
  if (forced_action >= 0) {
    mode = 3;
  } else if (p.training) {
    const double u1 = uniforms[g * 3], u2 = uniforms[g * 3 + 1];
    u3 = uniforms[g * 3 + 2];
    if (move < p.softmax_moves) mode = 1;
    else if (u1 < p.eps_softmax) mode = 1;
    else if (u2 < p.eps_random) mode = 2;
  }
  int chosen_child = best;
  if (mode == 1) {                         // softmax_action (Explorer.py:187-199)
    double e[MAXC_LIMIT];
    int mx = 0;
    for (int j = 0; j < k; ++j) mx = max(mx, nodes[root.child_base + j].visit);
    for (int j = 0; j < k; ++j) e[j] = exp((double)(nodes[root.child_base + j].visit - mx));
    const double s = np_sum_f64(e, k);
    for (int j = 0; j < k; ++j) e[j] = e[j] / s;
    const double s2 = np_sum_f64(e, k);
    double run = 0.0, last;
    for (int j = 0; j < k; ++j) { e[j] = e[j] / s2; }
    for (int j = 0; j < k; ++j) { run = j == 0 ? e[0] : run + e[j]; e[j] = run; }
    last = e[k - 1];
    chosen_child = k - 1;
    for (int j = 0; j < k; ++j) if (e[j] / last > u3) { chosen_child = j; break; }
  } else if (mode == 2) {                  // uniform over the legal actions (Explorer.py:86-89)
    // p = mask / n_valid over all actions; cdf = cumsum(p) / cdf[-1]; searchsorted(u, 'right').
    // The root's children are exactly the legal actions, in ascending order.
    const double pv = 1.0 / (double)k;     // 1 / np.sum(mask): int8 sum -> exact integer
    double run = 0.0;
    double cdf[MAXC_LIMIT];
    for (int j = 0; j < k; ++j) { run = j == 0 ? pv : run + pv; cdf[j] = run; }
    chosen_child = k - 1;
    for (int j = 0; j < k; ++j) if (cdf[j] / cdf[k - 1] > u3) { chosen_child = j; break; }
  }
  if (mode == 3) {
    chosen_child = -1;
    for (int j = 0; j < k; ++j)
      if (nodes[root.child_base + j].action == forced_action) chosen_child = j;
    if (chosen_child < 0) {                  // not a legal move of this position
      atomicOr(p.error_flag, 8);
      return;
    }
  }
  const int action = nodes[root.child_base + chosen_child].action;
  p.rec_action[gm] = action;
  Scs(R, real).step(action);
  p.root[g] = root.child_base + chosen_child;
}

// Re-rooting (Gamer.py:78-79) with the dead part of the tree dropped: the subtree of the new root is copied
// breadth-first into the other half of the game's arena (a node's children stay one contiguous block in the same
// order, so nothing observable changes), the old half is free for the move after next.  One wavefront per game:
// lane i takes the i-th node of the current BFS window and copies its block of children.
__global__ __launch_bounds__(64) void compact_kernel(SearchParams p) {
  const int g = blockIdx.x;
  const int lane = lane_id();
  if (p.real[g].terminal) return;
  const SNode* src = arena(p, g);
  SNode* dst = p.nodes + (size_t)g * p.cap + (size_t)(p.half[g] ^ 1) * p.half_cap;
  if (lane == 0) dst[0] = src[p.root[g]];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  int count = 1, head = 0;
  while (head < count) {
    const int window = count - head < 64 ? count - head : 64;     // nodes whose children are copied this round
    const int i = head + lane;
    int k = 0, cb = 0;
    if (lane < window) { k = dst[i].n_children; cb = dst[i].child_base; }
    int inc = k;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    const int total = __shfl(inc, 63, 64);
    if (count + total > p.half_cap) {                  // uniform
      if (lane == 0) atomicOr(p.error_flag, 1);
      return;
    }
    if (k > 0) {
      const int at = count + inc - k;
      for (int j = 0; j < k; ++j) dst[at + j] = src[cb + j];
      dst[i].child_base = at;
    }
    count += total;
    head += window;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (lane == 0) {
    p.half[g] ^= 1;
    p.root[g] = 0;
    p.node_count[g] = count;
  }
}

// the persistent kernel for a network (square / hexagonal taps), wavefronts per game and the game's children bound
// (at most 64 legal actions per position: the expansion handles one chunk of children instead of four)
typedef void (*persist_fn_t)(SearchParams, PersistArgs);
persist_fn_t persist_pick(bool hex, bool one_chunk) {
  if (hex) return one_chunk ? persist_kernel<true, 1> : persist_kernel<true, MAXC_CHUNKS>;
  return one_chunk ? persist_kernel<false, 1> : persist_kernel<false, MAXC_CHUNKS>;
}

}  // namespace

struct nz_scs_search {
  int device = 0, n_games = 0;
  nz_search_cfg cfg;
  ScsRules host_rules;
  SearchParams p;
  std::vector<void*> allocs;
  std::string error;
  // nz_scs_search_play: leaf batch, evaluations and the host-drawn randomness of one move
  float *images = nullptr, *probs = nullptr, *value = nullptr;
  int32_t *leaf_game = nullptr, *nchild = nullptr, *status = nullptr;
  double *noise = nullptr, *uniforms = nullptr;
  int64_t waves = 0;
  int32_t* active_pinned = nullptr;          // pinned host word + event: the early-finish poll of the move loop
  hipEvent_t ev_poll = nullptr;
  // nz_scs_search_play_round: the round's store and the slot bookkeeping of one move
  RoundStore round{};
  int64_t round_capacity = 0, round_games = 0;
  int32_t *to_record = nullptr, *restart = nullptr;
  int32_t* counters_base = nullptr;          // 6 ints: two (leaf, active, cache hit) triples
  // inference cache (nz_scs_search_cache)
  int cache_bits = 0;
  int64_t cache_entries = 0;
  int32_t cache_wave = 0;
  int cache_route = 0;                        // which route filled the table (1 wave by wave, 2 persistent); they do not mix
  // persistent route (persist_kernel): -1 follow the default (on where the network has a per-wavefront form), 0 off, 1 on
  int persist_mode = -1;
  int persist_used = 0;                       // the last play ran on it
  void (*persist_fn)(SearchParams, PersistArgs) = nullptr;     // the kernel variant of the last play
  bool persist_profile = false;               // HIP events around every persist_kernel launch (nz_scs_search_persist_profile)
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;
  double persist_ms = 0.0;
  int64_t persist_launches = 0, persist_mfmas = 0, persist_flops = 0;
  std::string persist_why;                    // why not
  PersistArgs pq{};
  int32_t* rec_slot_dev = nullptr;
  int32_t rec_slots = 0;
  // per-game maps and streams of the next plays (nz_scs_search_set_games)
  ScsRules* base_rules_dev = nullptr;         // the description of nz_scs_search_create (one row)
  ScsRules* game_rules_dev = nullptr;         // [n_game_rows]
  int32_t* rules_row_dev = nullptr;           // [n_games]
  int64_t n_game_rows = 0;
  std::vector<nz_rng*> game_streams;          // [n_game_rows] or empty: streams from the seeds
};

namespace {
thread_local std::string g_err;
nz_status sfail(nz_scs_search* h, nz_status code, const char* fmt, ...) {
  char buf[384];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->error = buf; else g_err = buf;
  return code;
}
#define S_HIP(h, call)                                                                            \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) return sfail((h), NZ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)
template <typename T>
bool dalloc(nz_scs_search* h, T** out, size_t n) {
  void* q = nullptr;
  if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) return false;
  h->allocs.push_back(q);
  *out = static_cast<T*>(q);
  return true;
}
nz_status check_flag(nz_scs_search* h, hipStream_t s) {
  int32_t f = 0;
  S_HIP(h, hipMemcpyAsync(&f, h->p.error_flag, sizeof(f), hipMemcpyDeviceToHost, s));
  S_HIP(h, hipStreamSynchronize(s));
  if (f) return sfail(h, NZ_ERR_OVERFLOW, "device check failed (flag %d: 1 arena full, 2 visit table/path too short, "
                                          "4 move ended before its search, 8 forced action is not legal, 16 more legal actions than the bound computed at create, 32 game longer than that bound)", f);
  return NZ_OK;
}
}  // namespace

extern "C" {

const char* nz_scs_search_last_error(const nz_scs_search* h) { return h ? h->error.c_str() : g_err.c_str(); }

nz_status nz_scs_search_create(nz_scs_search** out, const nz_scs_desc* d, const nz_search_cfg* cfg, int32_t n_games,
                               int32_t nodes_per_game, int32_t device) {
  if (!out || !d || !cfg) return sfail(nullptr, NZ_ERR_ARG, "null argument");
  *out = nullptr;
  if (n_games <= 0 || cfg->mcts_simulations <= 0 || (nodes_per_game > 0 && nodes_per_game < 4))
    return sfail(nullptr, NZ_ERR_ARG, "bad sizes");
  if (!cfg->keep_subtree) return sfail(nullptr, NZ_ERR_ARG, "keep_subtree = False is not supported (Gamer.py:78-79)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return sfail(nullptr, NZ_ERR_HIP, "no HIP device %d (no CPU fallback)", device);
  nz_scs_search* h = new nz_scs_search;
  h->device = device;
  h->n_games = n_games;
  h->cfg = *cfg;
  std::string err;
  if (!scs_fill_rules(d, &h->host_rules, &err)) { delete h; return sfail(nullptr, NZ_ERR_ARG, "%s", err.c_str()); }
  if (h->host_rules.planes * h->host_rules.tiles > MAX_ACTIONS) { delete h; return sfail(nullptr, NZ_ERR_ARG, "too many actions"); }
  (void)hipSetDevice(device);
  SearchParams& p = h->p;
  memset(&p, 0, sizeof(p));
  const size_t G = n_games;
  p.n_games = n_games;

  p.sims = cfg->mcts_simulations;
  p.training = cfg->training;
  p.softmax_moves = cfg->number_of_softmax_moves;
  p.negate_player = 2;                       // Explorer.py:124; SCS players are 0 and 1
  {   // what the records must hold, from the game description (the reference has no limits at all):
      //   decisions per game: one placement per unit, then per turn and unit at most ceil(movement / cheapest tile)
      //     steps + 1 "end movement", 1 "end fighting" or 1 target choice, 1 selection as attacker, 1 confirmation
      //   legal actions per position: the arrival tiles of a unit (placement), 7 per unit of the player to move
      //     (6 directions + end movement; end fighting + <= 6 adjacent enemy tiles), 6 * stacking + 1 attackers
    const ScsRules& R = h->host_rules;
    int min_cost = 1 << 30, units[2] = {0, 0}, arrive = 0;
    for (int t = 0; t < R.tiles; ++t) min_cost = std::min(min_cost, (int)R.cost[t]);
    if (min_cost < 1) { delete h; return sfail(nullptr, NZ_ERR_ARG, "a terrain with movement cost < 1 makes a game's length unbounded"); }
    long long moves = R.n_units;
    for (int u = 0; u < R.n_units; ++u) {
      units[R.u_player[u] ? 1 : 0] += 1;
      moves += (long long)R.turns * ((R.u_mov[u] + min_cost - 1) / min_cost + 4);
      int a = 0;
      for (int t = 0; t < R.tiles; ++t) a += R.arrival[u][t] ? 1 : 0;
      arrive = std::max(arrive, a);
    }
    const int children = std::max(std::max(arrive, 7 * std::max(units[0], units[1])), 6 * (int)R.stacking + 1);
    if (children > MAXC_LIMIT) {
      delete h;
      return sfail(nullptr, NZ_ERR_ARG, "this game can have %d legal actions in one position; the search kernels hold %d "
                                        "children per node", children, MAXC_LIMIT);
    }
    if (moves > 32000) { delete h; return sfail(nullptr, NZ_ERR_ARG, "a game of this description can last %lld decisions (limit 32000)", moves); }
    p.maxc = (children + 63) / 64 * 64;
    p.max_moves = (int32_t)moves + 8;
  }
  const int MAX_MOVES = p.max_moves, MAXC = p.maxc;
  // Tree arena: two halves per game (compact_kernel copies the kept subtree into the other half at every re-root), so a
  // half holds the kept subtree plus one move's expansions.  Default: 2.5 x the children bound per simulation (160 nodes
  // for up to 64 children: what every configuration measured so far needed several times over); a full half is
  // reported (NZ_ERR_OVERFLOW), never silent.
  if (nodes_per_game <= 0) nodes_per_game = 2 * (1 + cfg->mcts_simulations * (5 * MAXC / 2));
  p.cap = nodes_per_game & ~1;
  p.half_cap = p.cap / 2;
  p.tab_len = cfg->mcts_simulations * MAX_MOVES + 2;
  p.max_path = std::max(MAX_MOVES + 8, 64);   // one tree level per game decision (at least a lane each: wave_kernel reads path[lane])
  {   // numpy's pairwise_sum over num_actions float32 entries (np.sum in Explorer.py:169) as a block program
    const int A = h->host_rules.planes * h->host_rules.tiles;
    p.num_actions = A;
    p.pw_blocks = 0;
    bool fits = true;
    struct Rec {
      static void build(SearchParams& q, int lo, int hi, bool& ok) {
        const int n = hi - lo;
        if (n <= 128) {
          if (q.pw_blocks >= 48) { ok = false; return; }
          q.pw_hi[q.pw_blocks] = (uint16_t)hi;
          q.pw_be[q.pw_blocks] = (uint16_t)(n < 8 ? lo : lo + (n - n % 8));
          q.pw_merge[q.pw_blocks] = 0;
          ++q.pw_blocks;
          return;
        }
        int n2 = n / 2;
        n2 -= n2 % 8;
        build(q, lo, lo + n2, ok);
        build(q, lo + n2, hi, ok);
        if (ok) q.pw_merge[q.pw_blocks - 1] += 1;
      }
    };
    Rec::build(p, 0, A, fits);
    if (!fits) { nz_scs_search_destroy(h); return sfail(nullptr, NZ_ERR_ARG, "too many actions for the pairwise-sum program"); }
  }
  p.terminal_budget = 1 << 30;
  p.frac = cfg->root_exploration_fraction;
  p.one_minus_frac = 1.0 - cfg->root_exploration_fraction;
  p.value_factor = cfg->value_factor;
  p.eps_softmax = cfg->epsilon_softmax_exploration;
  p.eps_random = cfg->epsilon_random_exploration;
  ScsRules* rules = nullptr;
  double *bias = nullptr, *sq = nullptr;
  const size_t GM = G * MAX_MOVES;
  bool ok = dalloc(h, &rules, 1) && dalloc(h, &p.real, G) && dalloc(h, &p.scratch, G) &&
            dalloc(h, &p.nodes, G * (size_t)p.cap) && dalloc(h, &p.half, G) && dalloc(h, &p.node_count, G) && dalloc(h, &p.root, G) &&
            dalloc(h, &p.sims_left, G) && dalloc(h, &p.pending, G) && dalloc(h, &p.path, G * (size_t)p.max_path) &&
            dalloc(h, &p.path_len, G) && dalloc(h, &p.leaf_mask, G * MASK_WORDS) && dalloc(h, &p.leaf_count, 6) &&
            dalloc(h, &p.error_flag, 1) && dalloc(h, &p.counters, 16) && dalloc(h, &bias, (size_t)p.tab_len) &&
            dalloc(h, &sq, (size_t)p.tab_len) && dalloc(h, &p.rec_action, GM) && dalloc(h, &p.rec_tree_size, GM) &&
            dalloc(h, &p.rec_children, GM) && dalloc(h, &p.rec_bias, GM) && dalloc(h, &p.rec_root_value_sum, GM) &&
            dalloc(h, &p.rec_child_action, GM * MAXC) && dalloc(h, &p.rec_child_visit, GM * MAXC) &&
            dalloc(h, &p.rec_child_prior, GM * MAXC) && dalloc(h, &p.rec_child_value_sum, GM * MAXC);
  if (!ok) { nz_scs_search_destroy(h); return sfail(nullptr, NZ_ERR_HIP, "device allocation failed"); }
  std::vector<double> hb(p.tab_len), hs(p.tab_len);
  for (int n = 0; n < p.tab_len; ++n) {      // Explorer.py:103-112 with the host libm
    hb[n] = std::log(((double)n + cfg->pb_c_base + 1.0) / cfg->pb_c_base) + cfg->pb_c_init;
    hs[n] = std::sqrt((double)n);
  }
  if (hipMemcpy(rules, &h->host_rules, sizeof(ScsRules), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(bias, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(sq, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    nz_scs_search_destroy(h);
    return sfail(nullptr, NZ_ERR_HIP, "upload failed");
  }
  p.active_count = p.leaf_count + 1;
  p.hit_count = p.leaf_count + 2;
  p.clear_counters = nullptr;
  h->counters_base = p.leaf_count;
  p.rules = rules;
  p.rules_row = nullptr;
  h->base_rules_dev = rules;
  p.bias_tab = bias;
  p.sqrt_tab = sq;
  hipLaunchKernelGGL(search_reset_kernel, dim3((n_games + 127) / 128), dim3(128), 0, nullptr, p);
  if (hipDeviceSynchronize() != hipSuccess) { nz_scs_search_destroy(h); return sfail(nullptr, NZ_ERR_HIP, "reset failed"); }
  *out = h;
  return NZ_OK;
}

void nz_scs_search_destroy(nz_scs_search* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* q : h->allocs) (void)hipFree(q);
  if (h->active_pinned) (void)hipHostFree(h->active_pinned);
  if (h->ev_poll) (void)hipEventDestroy(h->ev_poll);
  if (h->ev_p0) { (void)hipEventDestroy(h->ev_p0); (void)hipEventDestroy(h->ev_p1); }
  {
    const RoundStore& r = h->round;
    void* store[] = {r.action, r.tree_size, r.children, r.child_action, r.child_visit, r.status, r.bias, r.root_value_sum,
                     r.child_prior, r.child_value_sum};
    for (void* q : store)
      if (q) (void)hipFree(q);
  }
  if (h->p.c_id) { (void)hipFree(h->p.c_id); (void)hipFree(h->p.c_probs); (void)hipFree(h->p.c_value); (void)hipFree(h->p.c_writer); (void)hipFree(h->p.c_check); }
  if (h->rec_slot_dev) {
    (void)hipFree(h->rec_slot_dev); (void)hipFree(h->pq.rec_count); (void)hipFree(h->pq.rec_digest);
    (void)hipFree(h->pq.rec_probs); (void)hipFree(h->pq.rec_value);
  }
  if (h->game_rules_dev) (void)hipFree(h->game_rules_dev);
  if (h->rules_row_dev) (void)hipFree(h->rules_row_dev);
  for (nz_rng* r : h->game_streams) nz_rng_destroy(r);
  delete h;
}

nz_status nz_scs_search_reset(nz_scs_search* h, void* stream) {
  if (!h) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(search_reset_kernel, dim3((h->n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream, h->p);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_search_root_children(nz_scs_search* h, int32_t* out_dev, void* stream) {
  if (!h || !out_dev) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(root_children_kernel, dim3((h->n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream, h->p, out_dev);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_search_begin_move(nz_scs_search* h, const double* noise_dev, void* stream) {
  if (!h) return NZ_ERR_ARG;
  if (h->cfg.training && !noise_dev) return sfail(h, NZ_ERR_ARG, "training search needs noise");
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(begin_move_kernel, dim3(h->n_games), dim3(64), 0, (hipStream_t)stream, h->p, noise_dev);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_search_select(nz_scs_search* h, float* images_dev, int32_t* leaf_game_dev, int32_t* n_leaves_host,
                               void* stream) {
  if (!h || !images_dev || !leaf_game_dev || !n_leaves_host) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  S_HIP(h, hipMemsetAsync(h->p.leaf_count, 0, 3 * sizeof(int32_t), s));
  h->p.c_bits = 0;                            // the cache lives in the library's own move loop (nz_scs_search_play)
  h->p.terminal_budget = 1 << 30;             // run on until a leaf needs an evaluation, as this API promises
  h->p.image_row_stride = 0;                  // NCHW images for the caller
  h->p.clear_counters = nullptr;
  hipLaunchKernelGGL(wave_kernel, dim3(h->n_games), dim3(64), 0, s, h->p, 2, (const float*)nullptr, (const float*)nullptr,
                     images_dev, leaf_game_dev);
  S_HIP(h, hipGetLastError());
  S_HIP(h, hipMemcpyAsync(n_leaves_host, h->p.leaf_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  S_HIP(h, hipStreamSynchronize(s));
  return check_flag(h, s);
}

nz_status nz_scs_search_expand(nz_scs_search* h, const float* probs_dev, const float* value_dev, void* stream) {
  if (!h || !probs_dev || !value_dev) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(wave_kernel, dim3(h->n_games), dim3(64), 0, (hipStream_t)stream, h->p, 1, probs_dev, value_dev,
                     (float*)nullptr, (int32_t*)nullptr);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_search_end_move(nz_scs_search* h, const double* uniforms_dev, void* stream) {
  if (!h) return NZ_ERR_ARG;
  if (h->cfg.training && !uniforms_dev) return sfail(h, NZ_ERR_ARG, "training search needs uniforms");
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(end_move_kernel, dim3((h->n_games + 63) / 64), dim3(64), 0, s, h->p, uniforms_dev,
                     (const int32_t*)nullptr);
  hipLaunchKernelGGL(compact_kernel, dim3(h->n_games), dim3(64), 0, s, h->p);
  S_HIP(h, hipGetLastError());
  return check_flag(h, s);
}

nz_status nz_scs_search_apply(nz_scs_search* h, const int32_t* actions_dev, const double* uniforms_dev, void* stream) {
  if (!h) return NZ_ERR_ARG;
  if (!actions_dev && h->cfg.training && !uniforms_dev) return sfail(h, NZ_ERR_ARG, "a training search choosing its own action needs uniforms");
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(end_move_kernel, dim3((h->n_games + 63) / 64), dim3(64), 0, s, h->p, uniforms_dev, actions_dev);
  hipLaunchKernelGGL(compact_kernel, dim3(h->n_games), dim3(64), 0, s, h->p);
  S_HIP(h, hipGetLastError());
  return check_flag(h, s);
}

namespace {
__global__ void scs_last_actions_kernel(SearchParams p, int32_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  const int n = p.real[g].length;
  out[g] = n > 0 && n <= p.max_moves ? p.rec_action[(size_t)g * p.max_moves + n - 1] : -1;
}
}  // namespace

nz_status nz_scs_search_last_actions(nz_scs_search* h, int32_t* actions_dev, void* stream) {
  if (!h || !actions_dev) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_last_actions_kernel, dim3((h->n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream, h->p, actions_dev);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_search_status(nz_scs_search* h, int32_t* status_dev, void* stream);

// Gamer.play_game for all games to the end with the network on the device too: the move loop of
// Training/Gamer.py:52-92 and Explorer.run_mcts, one simulation wave = [expand + select] kernel
// -> nz_boardnet_forward on the wave's leaves (batch size read from device memory).  The host
// only draws each move's random numbers (numpy RandomState streams, one per game, in the
// reference's order) and checks every few waves whether the move's searches are complete.
nz_status nz_scs_search_play(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, void* stream) {
  return nz_scs_search_play_moves(h, net, seeds_host, 0, stream);
}

}  // extern "C"

namespace {
// page-locked host array (the move loop's read-backs and uploads: from pageable memory every copy is staged)
template <typename T>
struct Pinned {
  T* p = nullptr;
  size_t n = 0;
  explicit Pinned(size_t count) : n(count) {
    if (hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault) != hipSuccess) p = nullptr;
  }
  ~Pinned() { if (p) (void)hipHostFree(p); }
  Pinned(const Pinned&) = delete;
  Pinned& operator=(const Pinned&) = delete;
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
  size_t size() const { return n; }
};
// a few host threads that live for one play call: every move hands them the same job (the games' random draws, a slice
// of the games each) -- starting and joining eight threads per move was a quarter of a millisecond of every move
class WorkerPool {
 public:
  explicit WorkerPool(int n) {
    for (int t = 0; t < n; ++t) threads_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  int size() const { return (int)threads_.size(); }
  void run(const std::function<void(int)>& job) {            // job(thread index) on every thread; returns when all are done
    { std::lock_guard<std::mutex> lk(m_); job_ = &job; pending_ = (int)threads_.size(); ++generation_; }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
  }
 private:
  void loop(int t) {
    long seen = 0;
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        job = job_;
      }
      (*job)(t);
      { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  long generation_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

// (re)allocate the round's store for `n` games
nz_status round_store(nz_scs_search* h, int64_t n) {
  if (n <= h->round_capacity) return NZ_OK;
  RoundStore& r = h->round;
  void* old[] = {r.action, r.tree_size, r.children, r.child_action, r.child_visit, r.status, r.bias, r.root_value_sum,
                 r.child_prior, r.child_value_sum};
  for (void* q : old)
    if (q) (void)hipFree(q);
  r = RoundStore{};
  h->round_capacity = 0;
  const size_t NM = (size_t)n * h->p.max_moves, NMC = NM * h->p.maxc;
  const bool ok = hipMalloc((void**)&r.action, NM * 4) == hipSuccess && hipMalloc((void**)&r.tree_size, NM * 4) == hipSuccess &&
                  hipMalloc((void**)&r.children, NM * 4) == hipSuccess && hipMalloc((void**)&r.bias, NM * 8) == hipSuccess &&
                  hipMalloc((void**)&r.root_value_sum, NM * 8) == hipSuccess &&
                  hipMalloc((void**)&r.child_action, NMC * 4) == hipSuccess && hipMalloc((void**)&r.child_visit, NMC * 4) == hipSuccess &&
                  hipMalloc((void**)&r.child_prior, NMC * 8) == hipSuccess && hipMalloc((void**)&r.child_value_sum, NMC * 8) == hipSuccess &&
                  hipMalloc((void**)&r.status, (size_t)n * 2 * 4) == hipSuccess;
  if (!ok) return sfail(h, NZ_ERR_HIP, "device allocation failed (round store for %lld games)", (long long)n);
  // unplayed moves' child rows are never written by archive_kernel: keep them defined
  S_HIP(h, hipMemset(r.child_action, 0, NMC * 4)); S_HIP(h, hipMemset(r.child_visit, 0, NMC * 4));
  S_HIP(h, hipMemset(r.child_prior, 0, NMC * 8)); S_HIP(h, hipMemset(r.child_value_sum, 0, NMC * 8));
  h->round_capacity = n;
  return NZ_OK;
}

// The library's move loop.  n_round == n_games: one game per slot (nz_scs_search_play_moves).  n_round > n_games: a round
// of n_round games over the engine's slots -- a slot whose game has ended hands its records to the round's store and
// starts the next game of the round (nz_scs_search_play_round); game i of the round is seeded with seeds_host[i]
// wherever and whenever it runs, so the round's games do not depend on the number of slots.
nz_status play_impl(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, int64_t n_round, int32_t max_moves,
                    void* stream) {
  if (!h || !net || (h->cfg.training && !seeds_host && h->game_streams.empty())) return sfail(h, NZ_ERR_ARG, "null argument");
  const bool refill = n_round > h->n_games;
  if (n_round < h->n_games) return sfail(h, NZ_ERR_ARG, "a round has at least one game per slot (%d)", h->n_games);
  if (refill && max_moves > 0) return sfail(h, NZ_ERR_ARG, "max_moves applies to one game per slot only");
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int G = h->n_games;
  const int MAX_MOVES = h->p.max_moves, MAXC = h->p.maxc;
  const ScsRules& R = h->host_rules;
  const int A = R.planes * R.tiles;
  int32_t nin = 0, npol = 0, nrows = 0, ncols = 0, nmax = 0;
  if (nz_boardnet_dims(net, &nin, &npol, &nrows, &ncols, &nmax) != NZ_OK) return sfail(h, NZ_ERR_ARG, "bad network handle");
  if (nin != R.channels || npol != R.planes || nrows != R.rows || ncols != R.cols)
    return sfail(h, NZ_ERR_ARG, "network is %d planes -> %d planes on %dx%d, the game needs %d -> %d on %dx%d", nin, npol,
                 nrows, ncols, R.channels, R.planes, R.rows, R.cols);
  if (nmax < G) return sfail(h, NZ_ERR_ARG, "network max_batch %d < %d games", nmax, G);
  int32_t* const counters = h->counters_base;
  float* net_rows = nullptr;                   // the leaf images go straight into the network's input rows
  int32_t row_stride = 0;
  if (nz_boardnet_input_rows(net, &net_rows, &row_stride) != NZ_OK) return sfail(h, NZ_ERR_ARG, "bad network handle");
  if (!h->images) {
    // evaluations: the network's in slots [0, G), cache hits in [G, 2 G) / [2 G, 3 G) by wave parity
    const bool ok = dalloc(h, &h->images, (size_t)G * R.channels * R.tiles) && dalloc(h, &h->probs, (size_t)3 * G * A) &&
                    dalloc(h, &h->value, (size_t)3 * G) && dalloc(h, &h->leaf_game, (size_t)G) &&
                    dalloc(h, &h->p.leaf_key, (size_t)2 * G) &&
                    dalloc(h, &h->nchild, (size_t)G) && dalloc(h, &h->status, (size_t)G * 7) &&
                    dalloc(h, &h->noise, (size_t)G * MAXC) && dalloc(h, &h->uniforms, (size_t)G * 3);
    if (!ok) return sfail(h, NZ_ERR_HIP, "device allocation failed");
  }
  // the persistent route: the network must have a per-wavefront form
  bool persist = false;
  size_t persist_lds = 0;
  h->persist_used = 0;
  {
    static const int env_mode = getenv("NZ_SCS_PERSIST") ? atoi(getenv("NZ_SCS_PERSIST")) : -1;       // A/B experiments
    const int mode = h->persist_mode >= 0 ? h->persist_mode : env_mode;
    nz::WaveNet wn{};
    std::string why;
    if (mode == 0) h->persist_why = "switched off";
    else if (!nz::boardnet_wave_program(net, &wn, &why)) h->persist_why = why;
    else {
      PersistArgs& q = h->pq;
      q.prog = wn.prog; q.net_floats = wn.lds_floats; q.stage_off = wn.stage_off; q.stage_floats = wn.stage_floats;
      q.inp = wn.inp; q.in_channels = wn.in_channels;
      h->persist_mfmas = wn.mfmas; h->persist_flops = wn.flops;
      q.wave_bytes = PERSIST_GAME_BYTES + (wn.lds_floats * 4 + 15) / 16 * 16;
      q.rules_per_game = h->n_game_rows > 0 ? 1 : 0;
      persist_lds = (size_t)(q.rules_per_game ? PERSIST_GAMES : 1) * PERSIST_RULES_BYTES + (size_t)PERSIST_GAMES * q.wave_bytes;
      if (persist_lds > 160 * 1024) h->persist_why = "four games' blocks do not fit in LDS";
      else {
        h->persist_fn = persist_pick(wn.hex != 0, h->p.maxc <= 64);
        const hipError_t e = hipFuncSetAttribute((const void*)h->persist_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)persist_lds);
        if (e != hipSuccess) { (void)hipGetLastError(); h->persist_why = "hipFuncSetAttribute failed"; }
        else { persist = true; h->persist_why.clear(); }
      }
      if (persist) h->persist_used = wn.hex ? 2 : 1;
    }
    if (mode == 1 && !persist) return sfail(h, NZ_ERR_STATE, "persistent route requested but not available: %s", h->persist_why.c_str());
    if (h->rec_slot_dev && !persist) return sfail(h, NZ_ERR_STATE, "leaf recording needs the persistent route: %s", h->persist_why.c_str());
  }
  h->pq.rec_slot = h->rec_slot_dev;
  if (h->cache_bits > 0 && h->cache_route != (persist ? 2 : 1)) {
    // the two routes keep their entries differently (check words / writer serial numbers): a table filled by the other
    // one starts empty
    if (h->cache_route != 0) {
      const size_t n = (size_t)h->cache_entries;
      S_HIP(h, hipMemsetAsync(h->p.c_id, 0, n * 2 * sizeof(uint64_t), (hipStream_t)stream));
      S_HIP(h, hipMemsetAsync(h->p.c_writer, 0, n * sizeof(int32_t), (hipStream_t)stream));
      S_HIP(h, hipMemsetAsync(h->p.c_check, 0, n * sizeof(uint64_t), (hipStream_t)stream));
    }
    h->cache_route = persist ? 2 : 1;
  }
  std::vector<nz_rng*> rngs;
  struct RngGuard {
    std::vector<nz_rng*>& v;
    ~RngGuard() { for (nz_rng* r : v) nz_rng_destroy(r); }
  } guard{rngs};
  const bool own_games = h->n_game_rows > 0;       // per-game maps (and streams) from nz_scs_search_set_games
  if (own_games && n_round > h->n_game_rows)
    return sfail(h, NZ_ERR_ARG, "%lld games in the round, %lld set with nz_scs_search_set_games", (long long)n_round, (long long)h->n_game_rows);
  // the stream of game i of the round: where its map's draws left it (set_games), else RandomState(seeds_host[i])
  auto new_stream = [&](int64_t i) -> nz_rng* {
    if (!h->game_streams.empty()) return nz_rng_clone(h->game_streams[(size_t)i]);
    return nz_rng_create(seeds_host[i]);
  };
  if (h->cfg.training)
    for (int g = 0; g < G; ++g) rngs.push_back(new_stream(g));
  std::vector<int32_t> rules_rows(G);
  if (own_games) {
    for (int g = 0; g < G; ++g) rules_rows[g] = g;
    S_HIP(h, hipMemcpyAsync(h->rules_row_dev, rules_rows.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
    S_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  }
  Pinned<int32_t> status((size_t)G * 7), nchild(G);
  Pinned<double> noise((size_t)G * MAXC), uni((size_t)G * 3);
  if (!status.p || !nchild.p || !noise.p || !uni.p) return sfail(h, NZ_ERR_HIP, "host allocation failed");
  const int n_thr = h->cfg.training && G >= 256 ? std::min<int>(8, std::max<unsigned>(1u, std::thread::hardware_concurrency())) : 0;
  WorkerPool pool(n_thr);
  nz_status st = nz_scs_search_reset(h, stream);
  if (st != NZ_OK) return st;
  h->waves = 0;
  h->round_games = 0;
  std::vector<int32_t> slot_game(G), to_record(G), restart(G);     // round bookkeeping: the game each slot is playing
  int64_t next_game = G;
  for (int g = 0; g < G; ++g) slot_game[g] = g;
  if (refill) {
    st = round_store(h, n_round);
    if (st != NZ_OK) return st;
    if (!h->to_record && !(dalloc(h, &h->to_record, (size_t)G) && dalloc(h, &h->restart, (size_t)G)))
      return sfail(h, NZ_ERR_HIP, "device allocation failed");
  }
  const dim3 grid1((G + 127) / 128), block1(128);
  const int64_t move_limit = refill ? (int64_t)MAX_MOVES * ((n_round + G - 1) / G + 1)
                                    : (max_moves > 0 && max_moves < MAX_MOVES ? max_moves : MAX_MOVES);
  for (int64_t move = 0; move < move_limit; ++move) {
    hipLaunchKernelGGL(search_status_kernel, grid1, block1, 0, s, h->p, h->status);
    if (h->cfg.training) hipLaunchKernelGGL(root_children_kernel, grid1, block1, 0, s, h->p, h->nchild);
    S_HIP(h, hipMemcpyAsync(status.data(), h->status, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (h->cfg.training)
      S_HIP(h, hipMemcpyAsync(nchild.data(), h->nchild, nchild.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    S_HIP(h, hipStreamSynchronize(s));
    if (refill) {
      bool archive = false, restarted = false;
      for (int g = 0; g < G; ++g) {
        to_record[g] = -1;
        restart[g] = 0;
        if (!status[(size_t)g * 7 + 4] || slot_game[g] < 0) continue;
        to_record[g] = slot_game[g];                 // the slot's game is over: its records go to the round's store
        archive = true;
        if (next_game < n_round) {                    // and the slot starts the round's next game
          slot_game[g] = (int32_t)next_game;
          if (h->cfg.training) {
            nz_rng_destroy(rngs[g]);
            rngs[g] = new_stream(next_game);
          }
          rules_rows[g] = (int32_t)next_game;          // (its own map, where the games have them)
          ++next_game;
          restart[g] = 1;
          restarted = true;
        } else {
          slot_game[g] = -1;
        }
      }
      if (archive) {
        S_HIP(h, hipMemcpyAsync(h->to_record, to_record.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(archive_kernel, dim3(G), dim3(64), 0, s, h->p, h->round, h->to_record);
      }
      if (restarted) {
        if (own_games)
          S_HIP(h, hipMemcpyAsync(h->rules_row_dev, rules_rows.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, s));
        S_HIP(h, hipMemcpyAsync(h->restart, restart.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(restart_kernel, grid1, block1, 0, s, h->p, h->restart);
        hipLaunchKernelGGL(search_status_kernel, grid1, block1, 0, s, h->p, h->status);
        if (h->cfg.training) hipLaunchKernelGGL(root_children_kernel, grid1, block1, 0, s, h->p, h->nchild);
        S_HIP(h, hipMemcpyAsync(status.data(), h->status, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (h->cfg.training)
          S_HIP(h, hipMemcpyAsync(nchild.data(), h->nchild, nchild.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      }
      if (archive) S_HIP(h, hipStreamSynchronize(s));     // the bookkeeping vectors are reused next move
    }
    bool any = false;
    for (int g = 0; g < G; ++g) any |= status[(size_t)g * 7 + 4] == 0;
    if (!any) break;
    if (h->cfg.training) {
      std::fill(noise.data(), noise.data() + noise.size(), 0.0);
      std::fill(uni.data(), uni.data() + uni.size(), 0.0);
      // (every game has its own stream: the draws of different games run on host threads side by side)
      auto draw = [&](int g0, int g1) {
        for (int g = g0; g < g1; ++g) {
          if (status[(size_t)g * 7 + 4]) continue;
          nz_rng* r = rngs[g];
          nz_rng_gamma(r, h->cfg.root_dist_alpha, h->cfg.root_dist_beta, nchild[g], &noise[(size_t)g * MAXC]);
          double* u = &uni[(size_t)g * 3];
          if (status[(size_t)g * 7 + 6] < h->cfg.number_of_softmax_moves) {
            u[2] = nz_rng_double(r);
          } else {
            u[0] = nz_rng_double(r);
            u[1] = nz_rng_double(r);
            if (u[0] < h->cfg.epsilon_softmax_exploration || u[1] < h->cfg.epsilon_random_exploration) u[2] = nz_rng_double(r);
          }
        }
      };
      if (n_thr <= 1) draw(0, G);
      else pool.run([&](int t) { draw((int)((int64_t)G * t / n_thr), (int)((int64_t)G * (t + 1) / n_thr)); });
      S_HIP(h, hipMemcpyAsync(h->noise, noise.data(), noise.size() * sizeof(double), hipMemcpyHostToDevice, s));
      S_HIP(h, hipMemcpyAsync(h->uniforms, uni.data(), uni.size() * sizeof(double), hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(begin_move_kernel, dim3(G), dim3(64), 0, s, h->p, h->noise);
    h->p.image_row_stride = row_stride;
    S_HIP(h, hipMemsetAsync(counters, 0, 6 * sizeof(int32_t), s));
    h->p.eval_probs = h->probs;
    h->p.eval_value = h->value;
    h->p.c_bits = h->cache_bits;
    h->p.terminal_budget = 1;                  // measured best (bench_scs.py: 1 -> 308 games/s, 16 -> 259, unbounded -> 226)
    if (const char* e = getenv("NZ_SCS_TERMINAL_BUDGET")) h->p.terminal_budget = std::max(1, atoi(e));   // tuning experiments
    if (persist) {                               // the whole move's search of every game: one launch
      if (h->persist_profile) {
        if (!h->ev_p0) { S_HIP(h, hipEventCreate(&h->ev_p0)); S_HIP(h, hipEventCreate(&h->ev_p1)); }
        S_HIP(h, hipEventRecord(h->ev_p0, s));
      }
      const dim3 pgrid((G + PERSIST_GAMES - 1) / PERSIST_GAMES);
      hipLaunchKernelGGL(h->persist_fn, pgrid, dim3(PERSIST_THREADS), persist_lds, s, h->p, h->pq);
      S_HIP(h, hipGetLastError());
      ++h->waves;
      static const bool move_times = getenv("NZ_SCS_MOVE_TIMES") != nullptr;     // experiment: the duration of every move's launch
      if (move_times) {
        int live = 0;
        for (int g = 0; g < G; ++g) live += status[(size_t)g * 7 + 4] == 0;
        const auto t0 = std::chrono::steady_clock::now();
        S_HIP(h, hipStreamSynchronize(s));
        fprintf(stderr, "move %lld: %d live games, %.3f ms\n", (long long)move, live,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      if (h->persist_profile) S_HIP(h, hipEventRecord(h->ev_p1, s));
      st = nz_scs_search_end_move(h, h->uniforms, stream);     // (synchronises)
      if (st != NZ_OK) return st;
      if (h->persist_profile) {
        float ms = 0.f;
        S_HIP(h, hipEventElapsedTime(&ms, h->ev_p0, h->ev_p1));
        h->persist_ms += ms;
        ++h->persist_launches;
      }
      continue;
    }
    const int sims = h->cfg.mcts_simulations;
    bool poll_pending = false;
    if (!h->active_pinned) {
      S_HIP(h, hipHostMalloc((void**)&h->active_pinned, sizeof(int32_t), hipHostMallocDefault));
      S_HIP(h, hipEventCreateWithFlags(&h->ev_poll, hipEventDisableTiming));
    }
    for (int w = 0;; ++w) {
      // two (leaf, active) counter pairs: wave w counts into pair w & 1 and zeroes the other one for wave w + 1
      h->p.leaf_count = counters + 3 * (w & 1);
      h->p.active_count = h->p.leaf_count + 1;
      h->p.hit_count = h->p.leaf_count + 2;
      h->p.clear_counters = counters + 3 * ((w + 1) & 1);
      h->p.c_wave = ++h->cache_wave;
      h->p.c_hit_base = G * (1 + (w & 1));
      hipLaunchKernelGGL(wave_kernel, dim3(G), dim3(64), 0, s, h->p, w ? 3 : 2, h->probs, h->value, net_rows, h->leaf_game);
      ++h->waves;
      if (w >= sims - 1) {                       // the move's last waves: wait for the count of games still searching
        int32_t active = 0;
        S_HIP(h, hipMemcpyAsync(&active, h->p.active_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        S_HIP(h, hipStreamSynchronize(s));
        if (active == 0) break;
        if (w > sims + 2) return sfail(h, NZ_ERR_STATE, "internal: a move's searches did not finish in %d waves", w);
      } else if ((w & 7) == 7) {
        // every game may have finished early (simulations that end in terminal leaves do not wait for a wave): the count
        // is copied to pinned memory and looked at eight waves later -- no wait, the GPU is never left without work; the
        // waves launched in between find nothing to do
        if (poll_pending && hipEventQuery(h->ev_poll) == hipSuccess) {
          poll_pending = false;
          if (*h->active_pinned == 0) break;
        }
        if (!poll_pending) {
          S_HIP(h, hipMemcpyAsync(h->active_pinned, h->p.active_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
          S_HIP(h, hipEventRecord(h->ev_poll, s));
          poll_pending = true;
        }
      }
      if (nz_boardnet_forward_rows(net, G, h->p.leaf_count, nullptr, h->probs, h->value, stream) != NZ_OK)
        return sfail(h, NZ_ERR_HIP, "network: %s", nz_boardnet_last_error(net));
      if (h->cache_bits > 0)                   // the leaves the network just evaluated enter the table (KeylessCache.put)
        hipLaunchKernelGGL(cache_put_kernel, dim3(G), dim3(64), 0, s, h->p, h->p.leaf_count);
    }
    if (poll_pending) S_HIP(h, hipEventSynchronize(h->ev_poll));     // (long done: the move's last waves were waited for)
    st = nz_scs_search_end_move(h, h->uniforms, stream);
    if (st != NZ_OK) return st;
  }
  h->p.leaf_count = counters;
  h->p.active_count = counters + 1;
  h->p.hit_count = counters + 2;
  h->p.clear_counters = nullptr;
  if (refill) {
    for (int g = 0; g < G; ++g)
      if (slot_game[g] >= 0) return sfail(h, NZ_ERR_STATE, "internal: the round did not finish in %lld moves", (long long)move_limit);
    h->round_games = n_round;
  }
  return NZ_OK;
}
}  // namespace

extern "C" {

nz_status nz_scs_search_play_moves(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, int32_t max_moves,
                                   void* stream) {
  return play_impl(h, net, seeds_host, h ? h->n_games : 0, max_moves, stream);
}

// A round of n_round >= n_games games over the engine's slots (see play_impl); read it with nz_scs_search_export_round.
nz_status nz_scs_search_play_round(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, int64_t n_round,
                                   void* stream) {
  return play_impl(h, net, seeds_host, n_round, 0, stream);
}

// nz_scs_search_export for the last nz_scs_search_play_round with more games than slots: arrays of n_round rows, and
// status [n_round][2] = (length, terminal value) of every game.  (A round of exactly n_games games is read with
// nz_scs_search_export / nz_scs_search_status.)
nz_status nz_scs_search_export_round(nz_scs_search* h, int32_t* actions, int32_t* tree_size, int32_t* n_children,
                                     double* bias, double* root_value_sum, int32_t* child_action, int32_t* child_visit,
                                     double* child_prior, double* child_value_sum, int32_t* status2,
                                     int64_t* counters_host, void* stream) {
  if (!h) return NZ_ERR_ARG;
  if (h->round_games <= 0) return sfail(h, NZ_ERR_STATE, "no finished round of more games than slots");
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const RoundStore& r = h->round;
  const size_t NM = (size_t)h->round_games * h->p.max_moves;
  const int MAXC = h->p.maxc;
#define CP(dst, src, n)                                                                          \
  if (dst) S_HIP(h, hipMemcpyAsync(dst, src, (n) * sizeof(*src), hipMemcpyDeviceToDevice, s))
  CP(actions, r.action, NM); CP(tree_size, r.tree_size, NM); CP(n_children, r.children, NM);
  CP(bias, r.bias, NM); CP(root_value_sum, r.root_value_sum, NM);
  CP(child_action, r.child_action, NM * MAXC); CP(child_visit, r.child_visit, NM * MAXC);
  CP(child_prior, r.child_prior, NM * MAXC); CP(child_value_sum, r.child_value_sum, NM * MAXC);
  CP(status2, r.status, (size_t)h->round_games * 2);
#undef CP
  if (counters_host) {
    S_HIP(h, hipMemcpyAsync(counters_host, h->p.counters, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    S_HIP(h, hipStreamSynchronize(s));
  }
  return NZ_OK;
}

// The reference's inference cache for the library's move loop (cache_choice "keyless" / "dict" of Gamer,
// Training/Gamer.py:20,53-55; KeylessCache.py:27-38: the table size is the largest power of two <= max_size).
// max_entries > 0: (re)allocate an empty table; 0: switch the cache off; < 0: empty the table, keep it.
nz_status nz_scs_search_cache(nz_scs_search* h, int64_t max_entries) {
  if (!h) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  SearchParams& p = h->p;
  const size_t A = (size_t)p.num_actions;
  if (max_entries >= 0) {
    if (p.c_id) { (void)hipFree(p.c_id); (void)hipFree(p.c_probs); (void)hipFree(p.c_value); (void)hipFree(p.c_writer); (void)hipFree(p.c_check); }
    p.c_id = nullptr; p.c_probs = nullptr; p.c_value = nullptr; p.c_writer = nullptr; p.c_check = nullptr;
    h->cache_bits = 0; h->cache_entries = 0;
    if (max_entries == 0) return NZ_OK;
    int bits = 0;
    while ((2ll << bits) <= max_entries && bits < 30) ++bits;          // closest power of two under max_entries
    if (bits == 0) bits = 1;
    const size_t n = (size_t)1 << bits;
    if (hipMalloc((void**)&p.c_id, n * 2 * sizeof(uint64_t)) != hipSuccess || hipMalloc((void**)&p.c_probs, n * A * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&p.c_value, n * sizeof(float)) != hipSuccess || hipMalloc((void**)&p.c_writer, n * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&p.c_check, n * sizeof(uint64_t)) != hipSuccess)
      return sfail(h, NZ_ERR_HIP, "cache allocation failed (%zu entries of %zu actions)", n, A);
    h->cache_bits = bits;
    h->cache_entries = (int64_t)n;
  }
  if (h->cache_bits > 0) {
    const size_t n = (size_t)h->cache_entries;
    S_HIP(h, hipMemset(p.c_id, 0, n * 2 * sizeof(uint64_t)));
    S_HIP(h, hipMemset(p.c_writer, 0, n * sizeof(int32_t)));
    S_HIP(h, hipMemset(p.c_check, 0, n * sizeof(uint64_t)));
    S_HIP(h, hipMemset(p.counters + 8, 0, 3 * sizeof(int64_t)));
    h->cache_wave = 0;
  }
  return NZ_OK;
}

// out4: hits, misses, entries in use, table size (Cache.get_hit_ratio / length / get_fill_ratio)
nz_status nz_scs_search_cache_stats(nz_scs_search* h, int64_t* out4_host) {
  if (!h || !out4_host) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  S_HIP(h, hipMemcpy(out4_host, h->p.counters + 8, 3 * sizeof(int64_t), hipMemcpyDeviceToHost));
  out4_host[3] = h->cache_entries;
  if (h->cache_bits > 0) {      // entries in use: counted on the table itself (the persistent route's writers keep no count)
    S_HIP(h, hipMemset(h->p.counters + 10, 0, sizeof(int64_t)));
    hipLaunchKernelGGL(cache_count_kernel, dim3(256), dim3(256), 0, nullptr, h->p.c_id, h->cache_entries, h->p.counters + 10);
    S_HIP(h, hipDeviceSynchronize());
    S_HIP(h, hipMemcpy(out4_host + 2, h->p.counters + 10, sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  return NZ_OK;
}

// Diagnostic (library built with -DNZ_SCS_STAMPS; zeros otherwise): shader ticks summed over all games and waves
// since the last reset -- out6: expansion, rules copy, scratch clone, descent, leaf mask + image, terminal simulations.
nz_status nz_scs_search_phase_ticks(nz_scs_search* h, int64_t* out6_host) {
  if (!h || !out6_host) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  S_HIP(h, hipMemcpy(out6_host, h->p.counters + 2, 6 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return NZ_OK;
}

// The persistent route of the library's move loop (persist_kernel).  enable: 1 require it (a play then fails where it is
// not available), 0 never use it, -1 the default (use it where it is available).  *used: whether the last play ran on it.
nz_status nz_scs_search_persistent(nz_scs_search* h, int32_t enable, int32_t* used) {
  if (!h) return NZ_ERR_ARG;
  if (enable >= -1 && enable <= 1) h->persist_mode = enable;
  if (used) *used = h->persist_used ? 1 : 0;
  return NZ_OK;
}

// Every game of the next plays on its OWN map, as the reference builds a new game object -- and with a "Randomized"
// config a new map -- per game (Training/Gamer.py:52, SCS_Game.py:1678-1738).  Host arrays for n games (n >= the games of
// a round): terrain float32 [n][tiles][3] (attack modifier, defense modifier, cost), vp int32 [n][n_vp0 + n_vp1][2] (row,
// column); and, optionally, the random stream each game goes on with after its map's draws (numpy RandomState state:
// mt_keys uint32 [n][624], mt_pos int32 [n]; NULL: the plays' seeds make the streams).  n = 0: back to the description's
// one map.  Game i of a round reads row i whichever slot plays it.
nz_status nz_scs_search_set_games(nz_scs_search* h, int64_t n, const float* terrain_host, const int32_t* vp_host,
                                  const uint32_t* mt_keys_host, const int32_t* mt_pos_host) {
  if (!h || n < 0 || (n > 0 && (!terrain_host || !vp_host)) || ((mt_keys_host == nullptr) != (mt_pos_host == nullptr)))
    return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  for (nz_rng* r : h->game_streams) nz_rng_destroy(r);
  h->game_streams.clear();
  if (h->game_rules_dev) { (void)hipFree(h->game_rules_dev); h->game_rules_dev = nullptr; }
  h->n_game_rows = 0;
  h->p.rules = h->base_rules_dev;
  h->p.rules_row = nullptr;
  if (n == 0) return nz_scs_search_reset(h, nullptr);
  if (n < h->n_games) return sfail(h, NZ_ERR_ARG, "%lld games set, the engine plays %d at a time", (long long)n, h->n_games);
  const ScsRules& b = h->host_rules;
  const int T = b.tiles, nv = b.n_vp[0] + b.n_vp[1];
  std::vector<ScsRules> rows((size_t)n, b);
  for (int64_t i = 0; i < n; ++i) {
    nz::scs_apply_map(&rows[(size_t)i], terrain_host + (size_t)i * T * 3, vp_host + (size_t)i * nv * 2);
    for (int t = 0; t < T; ++t)
      if (rows[(size_t)i].cost[t] < 1) return sfail(h, NZ_ERR_ARG, "game %lld: a terrain with movement cost < 1", (long long)i);
  }
  if (hipMalloc((void**)&h->game_rules_dev, rows.size() * sizeof(ScsRules)) != hipSuccess)
    return sfail(h, NZ_ERR_HIP, "device allocation failed (%lld game descriptions)", (long long)n);
  S_HIP(h, hipMemcpy(h->game_rules_dev, rows.data(), rows.size() * sizeof(ScsRules), hipMemcpyHostToDevice));
  if (!h->rules_row_dev && hipMalloc((void**)&h->rules_row_dev, (size_t)h->n_games * sizeof(int32_t)) != hipSuccess)
    return sfail(h, NZ_ERR_HIP, "device allocation failed");
  std::vector<int32_t> ident(h->n_games);
  for (int g = 0; g < h->n_games; ++g) ident[g] = g;
  S_HIP(h, hipMemcpy(h->rules_row_dev, ident.data(), ident.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (mt_keys_host)
    for (int64_t i = 0; i < n; ++i) {
      nz_rng* r = nz_rng_create_state(mt_keys_host + (size_t)i * 624, mt_pos_host[i], 0, 0.0);
      if (!r) return sfail(h, NZ_ERR_ARG, "game %lld: bad stream position %d", (long long)i, mt_pos_host[i]);
      h->game_streams.push_back(r);
    }
  h->n_game_rows = n;
  h->p.rules = h->game_rules_dev;
  h->p.rules_row = h->rules_row_dev;
  return nz_scs_search_reset(h, nullptr);      // the slots' games start on their own maps (row g for slot g)
}

// HIP-event timing of the persistent kernel on the stream it runs on.  enable >= 0: switch (1 also zeroes the sums).
// out4 (may be NULL): milliseconds summed over launches, launches, v_mfma_f32_16x16x32_bf16 per position, algorithmic
// float32 FLOPs per position.
nz_status nz_scs_search_persist_profile(nz_scs_search* h, int32_t enable, double* out4_host) {
  if (!h) return NZ_ERR_ARG;
  if (enable >= 0) {
    h->persist_profile = enable != 0;
    if (enable) { h->persist_ms = 0.0; h->persist_launches = 0; }
  }
  if (out4_host) {
    out4_host[0] = h->persist_ms; out4_host[1] = (double)h->persist_launches;
    out4_host[2] = (double)h->persist_mfmas; out4_host[3] = (double)h->persist_flops;
  }
  return NZ_OK;
}

// Test hook of the persistent route: keep the leaf evaluations (digest of the planes, post-softmax probabilities, value)
// of `n` games, up to `capacity` each, in the order the game's search consumed them.  n = 0: stop recording.
nz_status nz_scs_search_record(nz_scs_search* h, const int32_t* games_host, int32_t n, int32_t capacity) {
  if (!h || n < 0 || (n > 0 && (!games_host || capacity <= 0))) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  if (h->rec_slot_dev) {
    (void)hipFree(h->rec_slot_dev); (void)hipFree(h->pq.rec_count); (void)hipFree(h->pq.rec_digest);
    (void)hipFree(h->pq.rec_probs); (void)hipFree(h->pq.rec_value);
    h->rec_slot_dev = nullptr; h->pq.rec_count = nullptr; h->pq.rec_digest = nullptr; h->pq.rec_probs = nullptr; h->pq.rec_value = nullptr;
    h->rec_slots = 0;
  }
  h->pq.rec_slot = nullptr;
  if (n == 0) return NZ_OK;
  std::vector<int32_t> slot(h->n_games, -1);
  for (int i = 0; i < n; ++i) {
    if (games_host[i] < 0 || games_host[i] >= h->n_games) return sfail(h, NZ_ERR_ARG, "no game %d", games_host[i]);
    slot[games_host[i]] = i;
  }
  const size_t A = (size_t)h->p.num_actions, N = (size_t)n * capacity;
  if (hipMalloc((void**)&h->rec_slot_dev, slot.size() * 4) != hipSuccess || hipMalloc((void**)&h->pq.rec_count, (size_t)n * 4) != hipSuccess ||
      hipMalloc((void**)&h->pq.rec_digest, N * 16) != hipSuccess || hipMalloc((void**)&h->pq.rec_probs, N * A * 4) != hipSuccess ||
      hipMalloc((void**)&h->pq.rec_value, N * 4) != hipSuccess)
    return sfail(h, NZ_ERR_HIP, "device allocation failed (recording %d games x %d evaluations)", n, capacity);
  S_HIP(h, hipMemcpy(h->rec_slot_dev, slot.data(), slot.size() * 4, hipMemcpyHostToDevice));
  S_HIP(h, hipMemset(h->pq.rec_count, 0, (size_t)n * 4));
  h->pq.rec_cap = capacity;
  h->rec_slots = n;
  return NZ_OK;
}

// one recorded game: *count evaluations were consumed (those past the capacity are not kept); the arrays take
// min(*count, capacity) rows.  Host pointers; synchronises.
nz_status nz_scs_search_record_read(nz_scs_search* h, int32_t slot, int32_t* count, uint64_t* digests_host, float* probs_host,
                                    float* values_host) {
  if (!h || !count || slot < 0 || slot >= h->rec_slots) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  S_HIP(h, hipMemcpy(count, h->pq.rec_count + slot, 4, hipMemcpyDeviceToHost));
  const size_t n = (size_t)std::min(*count, h->pq.rec_cap), A = (size_t)h->p.num_actions, at = (size_t)slot * h->pq.rec_cap;
  if (digests_host) S_HIP(h, hipMemcpy(digests_host, h->pq.rec_digest + 2 * at, n * 16, hipMemcpyDeviceToHost));
  if (probs_host) S_HIP(h, hipMemcpy(probs_host, h->pq.rec_probs + at * A, n * A * 4, hipMemcpyDeviceToHost));
  if (values_host) S_HIP(h, hipMemcpy(values_host, h->pq.rec_value + at, n * 4, hipMemcpyDeviceToHost));
  return NZ_OK;
}

// Diagnostic (library built with -DNZ_PERSIST_STAMPS; zeros otherwise): shader ticks of the persistent kernel summed
// over all games and moves since the last reset -- out10: clone, descent, legal mask + list, planes + split, network,
// softmax + value, expansion, backup, whole moves (sum over games), the slowest single (game, move).
nz_status nz_scs_search_persist_ticks(nz_scs_search* h, int64_t* out10_host) {   // (13 values)
  if (!h || !out10_host) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  S_HIP(h, hipDeviceSynchronize());
  int64_t c[16];
  S_HIP(h, hipMemcpy(c, h->p.counters, sizeof(c), hipMemcpyDeviceToHost));
  for (int i = 0; i < 6; ++i) out10_host[i] = c[2 + i];
  for (int i = 6; i < 9; ++i) out10_host[i] = c[5 + i];
  out10_host[9] = c[14];
  for (int i = 0; i < 3; ++i) out10_host[10 + i] = c[8 + i];
  return NZ_OK;
}

// Diagnostic: shader ticks of one per-wavefront-pair network pass (netbench_kernel), `blocks` workgroups of four game
// slots each running `iters` passes; ticks_host[blocks * 4].
nz_status nz_scs_netbench(nz_boardnet* net, int32_t blocks, int32_t iters, uint64_t* ticks_host) {
  if (!net || blocks <= 0 || iters <= 0 || !ticks_host) return NZ_ERR_ARG;
  const bool quad = getenv("NZ_NETBENCH_QUAD") != nullptr;        // prototype: four wavefronts per game
  nz::WaveNet wn{};
  std::string why;
  if (!nz::boardnet_wave_program(net, &wn, &why)) return sfail(nullptr, NZ_ERR_STATE, "%s", why.c_str());
  PersistArgs q{};
  q.prog = wn.prog; q.net_floats = wn.lds_floats; q.stage_off = wn.stage_off; q.stage_floats = wn.stage_floats;
  q.inp = wn.inp; q.in_channels = wn.in_channels;
  q.wave_bytes = PERSIST_GAME_BYTES + (wn.lds_floats * 4 + 15) / 16 * 16;
  const size_t lds = (size_t)PERSIST_RULES_BYTES + (size_t)PERSIST_GAMES * q.wave_bytes;
  unsigned long long* out = nullptr;
  S_HIP(nullptr, hipMalloc((void**)&out, (size_t)blocks * PERSIST_GAMES * 8));
  hipError_t e = wn.hex ? hipFuncSetAttribute((const void*)netbench_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                        : hipFuncSetAttribute((const void*)netbench_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess && quad && !wn.hex) {
    e = hipFuncSetAttribute((const void*)netbench4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(netbench4_kernel<false>, dim3(blocks), dim3(PERSIST_GAMES * 4 * 64), lds, nullptr, q, iters, out);
      e = hipDeviceSynchronize();
    }
  } else if (e == hipSuccess) {
    if (wn.hex) hipLaunchKernelGGL(netbench_kernel<true>, dim3(blocks), dim3(PERSIST_THREADS), lds, nullptr, q, iters, out);
    else hipLaunchKernelGGL(netbench_kernel<false>, dim3(blocks), dim3(PERSIST_THREADS), lds, nullptr, q, iters, out);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(ticks_host, out, (size_t)blocks * PERSIST_GAMES * 8, hipMemcpyDeviceToHost);
  (void)hipFree(out);
  if (e != hipSuccess) return sfail(nullptr, NZ_ERR_HIP, "netbench: %s", hipGetErrorString(e));
  return NZ_OK;
}

nz_status nz_scs_search_limits(const nz_scs_search* h, int32_t* max_moves, int32_t* max_children) {
  if (!h) return NZ_ERR_ARG;
  if (max_moves) *max_moves = h->p.max_moves;
  if (max_children) *max_children = h->p.maxc;
  return NZ_OK;
}

nz_status nz_scs_search_waves(const nz_scs_search* h, int64_t* waves) {
  if (!h || !waves) return NZ_ERR_ARG;
  *waves = h->waves;
  return NZ_OK;
}

nz_status nz_scs_search_export(nz_scs_search* h, int32_t* actions, int32_t* tree_size, int32_t* n_children,
                               double* bias, double* root_value_sum, int32_t* child_action, int32_t* child_visit,
                               double* child_prior, double* child_value_sum, int64_t* counters_host, void* stream) {
  if (!h) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const SearchParams& p = h->p;
  const size_t GM = (size_t)h->n_games * p.max_moves;
  const int MAXC = p.maxc;
#define CP(dst, src, n)                                                                          \
  if (dst) S_HIP(h, hipMemcpyAsync(dst, src, (n) * sizeof(*src), hipMemcpyDeviceToDevice, s))
  CP(actions, p.rec_action, GM); CP(tree_size, p.rec_tree_size, GM); CP(n_children, p.rec_children, GM);
  CP(bias, p.rec_bias, GM); CP(root_value_sum, p.rec_root_value_sum, GM);
  CP(child_action, p.rec_child_action, GM * MAXC); CP(child_visit, p.rec_child_visit, GM * MAXC);
  CP(child_prior, p.rec_child_prior, GM * MAXC); CP(child_value_sum, p.rec_child_value_sum, GM * MAXC);
#undef CP
  if (counters_host) {
    S_HIP(h, hipMemcpyAsync(counters_host, p.counters, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    S_HIP(h, hipStreamSynchronize(s));
  }
  return NZ_OK;
}

}  // extern "C"


extern "C" nz_status nz_scs_search_status(nz_scs_search* h, int32_t* status_dev, void* stream) {
  if (!h || !status_dev) return NZ_ERR_ARG;
  S_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(search_status_kernel, dim3((h->n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream, h->p, status_dev);
  S_HIP(h, hipGetLastError());
  return NZ_OK;
}
