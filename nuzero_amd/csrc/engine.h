// Internal declarations shared by the engine's translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nuzero_amd.h"

namespace nz {

constexpr int TTT_ACTIONS = 9;
constexpr int TTT_MAX_MOVES = 9;
constexpr int TTT_TABLE_ROWS = 19683;   // 3^9
constexpr int MAX_PATH = 16;            // TTT depth <= 9 plus the root
constexpr int LANES_PER_GAME = 16;      // one 16-lane DPP row per game tree

// node link word y: n_children[0:12) | action[12:24) | to_play[24:28) | term[28:32)
constexpr uint32_t TO_PLAY_UNSET = 15u;
__host__ __device__ inline uint32_t pack_meta(uint32_t n_children, uint32_t action, uint32_t to_play,
                                              uint32_t term) {
  return n_children | (action << 12) | (to_play << 24) | (term << 28);
}
__host__ __device__ inline uint32_t meta_children(uint32_t m) { return m & 0xfffu; }
__host__ __device__ inline uint32_t meta_action(uint32_t m) { return (m >> 12) & 0xfffu; }
__host__ __device__ inline uint32_t meta_to_play(uint32_t m) { return (m >> 24) & 0xfu; }

// One tree node (Search/Node.py:3-32), 48 bytes = three 16-byte loads; a node's children are
// contiguous and in ascending action order.
struct __attribute__((aligned(16))) TNode {
  double value_sum;
  double q;               // value_sum / visit, 0.0 while unvisited (Node.value()); rewritten by every backup
  double prior;
  int32_t visit;
  uint32_t first;         // index of the first child
  uint32_t meta;          // packed: see pack_meta
  uint32_t pad[3];
};
static_assert(sizeof(TNode) == 48, "three 16-byte loads per node");

// Everything the tree kernels need, passed by value.
struct TreeParams {
  TNode* nodes;           // [n_slots][cap]
  int32_t cap;
  int32_t n_games;        // games per self-play round (records, random tables)
  int32_t n_slots;        // games in flight at once (tree arenas); lock-step needs n_slots == n_games
  int32_t* next_game;     // work queue head of the persistent kernel
  // per-game state
  uint32_t* board;        // live game: player-one stones | player-two stones << 16
  int32_t* length;
  int32_t* alive;
  int32_t* outcome;
  int32_t* root;
  int32_t* node_count;
  int32_t* sims_left;
  int32_t* pending;       // leaf slot awaiting a network result, or -1
  uint32_t* leaf_board;   // scratch position at the pending leaf
  int32_t* path;          // [G][MAX_PATH]
  int32_t* path_len;
  int32_t* sim_count;
  int32_t* exp_count;
  int32_t* sel_nodes;     // descent levels (internal nodes scored)
  int32_t* sel_children;  // children scored
  int32_t* new_nodes;     // nodes created by the game's expansions (sum of the children counts)
  int32_t* n_root_children;
  int32_t* desync;        // 1: the host's pre-drawn randomness did not fit this game (selfplay.hip)
  // leaf queue feeding the network kernel
  int32_t* leaf_count;    // [2], ping-pong by iteration parity
  uint32_t* leaf_boards;  // [G]
  const float* leaf_logits;  // [G][A]
  const float* leaf_value;   // [G]
  // table evaluator (test hook) or nullptr
  const float* table;
  // search constants
  const double* bias_tab;  // log((N + base + 1) / base) + init, N = 0..tab_len-1
  const double* sqrt_tab;  // sqrt(N)
  int32_t tab_len;
  int32_t sims;
  int32_t negate_player;
  double value_factor;
  double frac, one_minus_frac;
  int32_t training;
  int32_t softmax_moves;
  double eps_softmax, eps_random;
  int32_t sims_per_cycle;  // persistent kernel: simulations a game may run between two network passes
  int32_t slots_per_wg;    // persistent kernel: game slots a workgroup plays (1..16 of its 16 network rows): fewer slots
                           // than 16 x CUs are spread over all CUs rather than packed into a quarter of them
  int32_t max_cycles;      // persistent kernel: tree/network cycles a workgroup may run before it gives up
                           // (error flag 64) -- a bound every wave reaches, whatever goes wrong
  int32_t* error_flag;
  // per-move records, [G][T]...
  uint32_t* hist_board;
  int32_t* hist_action;
  int32_t* hist_visits;    // [G][T][A]
  int32_t* hist_tree_size;
  int32_t* hist_children;
  double* hist_bias;
  double* hist_prior;      // [G][T][A]
  double* hist_value_sum;  // [G][T][A]
  double* hist_root_value_sum;
};

void launch_reset(const TreeParams& p, hipStream_t s);
void launch_noise(const TreeParams& p, const double* noise, hipStream_t s);
void launch_advance(const TreeParams& p, int iteration, hipStream_t s);
void launch_finish_move(const TreeParams& p, const double* uniforms, const int32_t* forced, hipStream_t s);
void launch_last_actions(const TreeParams& p, int32_t* actions, hipStream_t s);
void launch_export_states(const TreeParams& p, float* states, hipStream_t s);
void launch_export_visits(const TreeParams& p, int32_t* visits, int32_t* actions, int32_t* tree_size,
                          int32_t* n_children, double* bias, hipStream_t s);

int selfplay_blocks(int n_slots, int slots_per_wg);
// stamps != nullptr selects the diagnostic build ([blocks][4] phase ticks, see selfplay.hip)
void launch_selfplay(const TreeParams& p, const struct NetProgram* prog_dev, int n_layers, const float* weights,
                     const double* noise, const double* uniforms, unsigned long long* stamps, hipStream_t s);

// ---- network ----------------------------------------------------------------
// Arithmetic of the fused network: net_dev.hpp.  K groups are 32 channels; the packed weights of
// one K group of one 16-channel output tile are [tap][piece][lane][8 bf16].
constexpr int NET_KG_CHANNELS = 32;
constexpr int NET_KG_DWORDS = 9 * 3 * 64 * 4;
constexpr int NET_ACT_BUFFERS = 2;                 // activation buffers in LDS (ping-pong)
constexpr int NET_DST_POLICY = NET_ACT_BUFFERS;    // NetJob::dst: policy logits out
constexpr int NET_DST_VALUE = NET_ACT_BUFFERS + 1; // NetJob::dst: value out

// The network is compiled on the host into one job list per wave (net_dev.hpp).
// A job is one (conv layer, 16-channel output tile, output-cell group) unit; the
// jobs of a stage are independent, a workgroup barrier separates stages.
struct NetJob {
  int32_t w_off;       // dword offset of this (layer, n-tile)'s packed main weights [kgroup][tap][piece][lane][8 bf16]
  int32_t wx_off;      // float offset of its packed input-plane weights [tap][lane]
  int32_t next_w_off;  // w_off of the next job of this wave that reads weights, or -1 (prefetch target)
  int16_t kgroups;     // 32-channel K groups read from the source activation buffer
  int16_t nt;          // output tile: channels 16 nt .. 16 nt + 15
  int8_t extra;        // 1: also read the (<= 4) input planes as one extra K step
  int8_t og;           // output-cell group (net_dev.hpp OG_MASK); OG_NONE = no work, barrier only
  int8_t src, dst;     // activation buffers 0..1; dst NET_DST_POLICY / NET_DST_VALUE = network outputs
  int8_t res;          // residual buffer or -1
  int8_t act;          // 0 none, 1 relu, 2 tanh, 3 elu
  int8_t stage_end;    // 1: workgroup barrier after this job
  int8_t pad;
};
constexpr int NET_WAVES_HOST = 8;
constexpr int NET_MAX_JOBS = 192;
constexpr int OG_NONE = 7;
struct NetProgram {
  int32_t n_jobs[NET_WAVES_HOST];
  int32_t first_w_off[NET_WAVES_HOST];   // w_off of each wave's first weight-reading job, or -1
  NetJob jobs[NET_WAVES_HOST][NET_MAX_JOBS];
};

void launch_net(const NetProgram* prog_dev, int n_layers, const float* packed_weights,
                const uint32_t* boards, const float* states, const int32_t* count_dev, int max_positions,
                float* logits, float* value, float* probs, unsigned long long* stamps, hipStream_t s);

}  // namespace nz
