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

// Everything the tree kernels need, passed by value.
struct TreeParams {
  // structure-of-arrays node storage, [G][cap]; a node's children are
  // contiguous and in ascending action order
  int32_t* visit;
  double* value_sum;
  double* prior;
  uint2* link;            // x = index of first child, y = packed meta
  int32_t cap;
  int32_t n_games;
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
void launch_finish_move(const TreeParams& p, const double* uniforms, hipStream_t s);
void launch_export_states(const TreeParams& p, float* states, hipStream_t s);
void launch_export_visits(const TreeParams& p, int32_t* visits, int32_t* actions, int32_t* tree_size,
                          int32_t* n_children, double* bias, hipStream_t s);

void launch_selfplay(const TreeParams& p, const struct NetProgram* prog_dev, int n_layers, const float* weights,
                     const double* noise, const double* uniforms, hipStream_t s);

// ---- network ----------------------------------------------------------------
struct NetLayer {
  int32_t cin_main;    // channels read from the activation buffer (multiple of 16 after padding)
  int32_t kgroups;     // cin_main / 16
  int32_t extra;       // 1: also read the (<=4) input planes as one extra K step
  int32_t cout;        // real output channels
  int32_t ntiles;      // ceil(cout / 16)
  int32_t src, dst;    // activation buffer ids (0/1); dst 2 = policy out, 3 = value out
  int32_t res;         // residual buffer id or -1
  int32_t act;         // 0 none, 1 relu, 2 tanh
  int32_t w_off;       // float offset of the packed main weights
  int32_t wx_off;      // float offset of the packed extra-plane weights
};
constexpr int NET_MAX_LAYERS = 128;
struct NetProgram {
  int32_t n_layers;
  NetLayer layers[NET_MAX_LAYERS];
};

void launch_net(const NetProgram* prog_dev, int n_layers, const float* packed_weights,
                const uint32_t* boards, const float* states, const int32_t* count_dev, int max_positions,
                float* logits, float* value, float* probs, hipStream_t s);

}  // namespace nz
