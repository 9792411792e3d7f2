// Lock-step tree kernels: every game of the batch advances through the same
// move together, the network runs as its own launch between two `advance`
// launches.  This is the path nz_engine_move drives (randomness injected by the
// caller) and the fallback of the persistent self-play kernel (selfplay.hip).
// Semantics and exactness notes: tree_dev.hpp.
#include "tree_dev.hpp"

namespace nz {
namespace {

constexpr int BLOCK = 256;
constexpr int GAMES_PER_BLOCK = BLOCK / LANES_PER_GAME;

__global__ void reset_kernel(TreeParams p) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0) {
    p.leaf_count[0] = 0;
    p.leaf_count[1] = 0;
    *p.error_flag = 0;
  }
  if (g == 0) *p.next_game = p.n_slots < p.n_games ? p.n_slots : p.n_games;
  if (g < p.n_games) p.alive[g] = g < p.n_slots ? 1 : 0;
  if (g < p.n_slots) {          // per-slot state
    arena_reset(arena_of(p, g));
    p.board[g] = 0u;
    p.root[g] = 0;
    p.node_count[g] = 1;
    p.sims_left[g] = p.sims;
    p.pending[g] = -1;
    p.leaf_board[g] = 0u;
    p.path_len[g] = 0;
    p.n_root_children[g] = 0;
  }
  if (g < p.n_games) {          // per-game records
    p.length[g] = 0;
    p.outcome[g] = 0;
    p.sim_count[g] = 0;
    p.exp_count[g] = 0;
    p.sel_nodes[g] = 0;
    p.sel_children[g] = 0;
    p.new_nodes[g] = 0;
    p.desync[g] = 0;
    hist_clear(p, g);
  }
}

__global__ __launch_bounds__(BLOCK) void noise_kernel(TreeParams p, const double* __restrict__ noise) {
  const int g = (blockIdx.x * BLOCK + threadIdx.x) / LANES_PER_GAME;
  const int sub = threadIdx.x & (LANES_PER_GAME - 1);
  if (g >= p.n_games || !p.alive[g]) return;
  noise_row(p, arena_of(p, g), p.root[g], noise + (size_t)g * TTT_ACTIONS, sub);
}

// Finish the pending expansion, then simulate until the next leaf that needs
// the network (or until this move's simulations are used up).
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

    if (pend >= 0) {
      // network result for the leaf queued by the previous launch
      const int path_len = p.path_len[g];
      my_node = p.path[g * MAX_PATH + sub];
      const int leaf = p.path[g * MAX_PATH + path_len - 1];
      const uint32_t sb = p.leaf_board[g];
      const float logit = sub < 9 ? p.leaf_logits[(size_t)pend * TTT_ACTIONS + sub] : 0.0f;
      const float prob = row_softmax9(logit, sub);
      const double value = (double)p.leaf_value[pend];
      node_count = expand_row(p, t, leaf, t[leaf].meta, sb, prob, sub, node_count);
      backup_row(t, my_node, path_len, value, sub);
      row_memory_fence();
      --sims_left;
      ++n_sim;
      ++n_exp;
    }

    while (sims_left > 0) {
      const Descent d = descend_row(p, t, root, board, sub, my_node);
      n_lvl += d.levels;
      n_kid += d.children;
      // evaluate (Explorer.py:137-181)
      const int term = ttt_terminal(d.sb);
      if (term != 0) {
        if (sub == 0)
          t[d.node].meta = pack_meta(0u, meta_action(d.lk.y), (uint32_t)ttt_player(d.sb), (uint32_t)term);
        backup_row(t, my_node, d.path_len, (double)term_value(term), sub);
        row_memory_fence();
        --sims_left;
        ++n_sim;
        continue;
      }
      if (p.table != nullptr) {
        const float* row = p.table + (size_t)ttt_code(d.sb) * 10;
        const float prob = sub < 9 ? row[sub] : 0.0f;
        const double value = (double)row[9];
        node_count = expand_row(p, t, d.node, d.lk.y, d.sb, prob, sub, node_count);
        backup_row(t, my_node, d.path_len, value, sub);
        row_memory_fence();
        --sims_left;
        ++n_sim;
        ++n_exp;
        continue;
      }
      // leaf needs the network: park the simulation until the next launch
      want_slot = true;
      slot_board = d.sb;
      p.path[g * MAX_PATH + sub] = my_node;
      if (sub == 0) {
        p.path_len[g] = d.path_len;
        p.leaf_board[g] = d.sb;
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
      p.new_nodes[g] = node_count - 1;
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

__global__ void finish_move_kernel(TreeParams p, const double* __restrict__ uniforms,
                                   const int32_t* __restrict__ forced) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  if (!p.alive[g]) {
    p.n_root_children[g] = 0;
    return;
  }
  if (p.sims_left[g] != 0 || p.pending[g] >= 0) {
    atomicOr(p.error_flag, 4);      // search did not complete
    return;
  }
  const int move = p.length[g];
  const MoveResult r = finish_move_one(p, arena_of(p, g), g, p.root[g], p.board[g], move,
                                       uniforms ? uniforms + (size_t)g * 3 : nullptr, forced ? forced[g] : -1);
  if (r.chosen < 0) return;
  p.board[g] = r.new_board;
  p.length[g] = move + 1;
  p.root[g] = r.new_root;
  p.sims_left[g] = p.sims;
  p.pending[g] = -1;
  if (r.term != 0) {
    p.alive[g] = 0;
    p.outcome[g] = term_value(r.term);
    p.n_root_children[g] = 0;
  } else {
    p.n_root_children[g] = r.new_children;
  }
}

// action taken at each game's most recent move, -1 before the first
__global__ void last_actions_kernel(TreeParams p, int32_t* __restrict__ actions) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.n_games) return;
  const int n = p.length[g];
  actions[g] = n > 0 ? p.hist_action[g * TTT_MAX_MOVES + n - 1] : -1;
}

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
  const int n = p.n_games > p.n_slots ? p.n_games : p.n_slots;
  hipLaunchKernelGGL(reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
}
void launch_noise(const TreeParams& p, const double* noise, hipStream_t s) {
  const int blocks = (p.n_games + GAMES_PER_BLOCK - 1) / GAMES_PER_BLOCK;
  hipLaunchKernelGGL(noise_kernel, dim3(blocks), dim3(BLOCK), 0, s, p, noise);
}
void launch_advance(const TreeParams& p, int iteration, hipStream_t s) {
  const int blocks = (p.n_games + GAMES_PER_BLOCK - 1) / GAMES_PER_BLOCK;
  hipLaunchKernelGGL(advance_kernel, dim3(blocks), dim3(BLOCK), 0, s, p, iteration);
}
void launch_finish_move(const TreeParams& p, const double* uniforms, const int32_t* forced, hipStream_t s) {
  hipLaunchKernelGGL(finish_move_kernel, dim3((p.n_games + 255) / 256), dim3(256), 0, s, p, uniforms, forced);
}
void launch_last_actions(const TreeParams& p, int32_t* actions, hipStream_t s) {
  hipLaunchKernelGGL(last_actions_kernel, dim3((p.n_games + 255) / 256), dim3(256), 0, s, p, actions);
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
