/*
 * nuzero_amd.h -- C ABI of the MI355X self-play engine.
 *
 * The reference (guilherme439/NuZero) is pure Python and has no FFI; the
 * boundary this library replaces is the Python contract between the trainer and
 * the self-play worker.  Each entry point names the
 * reference code whose work it takes over.  All paths are relative to the
 * reference repository root.
 *
 * Conventions
 *   - plain C symbols, opaque handle, int status (0 = NZ_OK), no exceptions;
 *   - one engine per GPU, not thread-safe (one Gamer = one single-threaded
 *     actor in the reference, Training/Gamer.py:17-37);
 *   - "dev" pointers are device pointers owned by the caller (PyTorch-ROCm
 *     tensors); "host" pointers are ordinary host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     work is enqueued on it and is NOT synchronised unless stated.
 */
#ifndef NUZERO_AMD_H
#define NUZERO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nz_engine nz_engine;
typedef struct nz_rng nz_rng;
typedef int nz_status;

enum {
  NZ_OK = 0,
  NZ_ERR_ARG = 1,        /* bad argument / unsupported configuration */
  NZ_ERR_HIP = 2,        /* a HIP call failed; see nz_last_error */
  NZ_ERR_STATE = 3,      /* call out of order (e.g. search before weights) */
  NZ_ERR_OVERFLOW = 4    /* a device-side capacity check failed */
};

enum { NZ_GAME_TIC_TAC_TOE = 0 };
enum { NZ_ACT_TANH = 0, NZ_ACT_RELU = 1 };
enum { NZ_ARCH_RECURRENT = 0, NZ_ARCH_RESNET = 1, NZ_ARCH_CONVNET = 2 };

/* Search hyper-parameters: exactly the keys the reference's Explorer reads
 * from its search config (Configs/Search/Examples/documentation_search_config.yaml:1-47;
 * Search/Explorer.py:48,74,79-80,105-106,122,202-205). */
typedef struct nz_search_cfg {
  int32_t mcts_simulations;
  int32_t keep_subtree;                 /* must be 1 (Gamer.py:78-79) */
  double pb_c_base;
  double pb_c_init;
  int32_t number_of_softmax_moves;
  int32_t training;                     /* Explorer(search_config, training) */
  double epsilon_softmax_exploration;
  double epsilon_random_exploration;
  double value_factor;
  double root_exploration_fraction;
  double root_dist_alpha;
  double root_dist_beta;
} nz_search_cfg;

/* Game description (the duck-typed Game surface, Games/Game.py). */
typedef struct nz_game_desc {
  int32_t game;                         /* NZ_GAME_* */
  int32_t negate_player;                /* Q is negated iff parent.to_play == this (Explorer.py:124) */
} nz_game_desc;

/* The policy/value networks of the reference (square convs; hexagonal ones for nz_boardnet_*), all with the
 * "conv" policy head and the "reduce" value head (blocks.py:46-92,130-170):
 *   NZ_ARCH_RECURRENT  RecurrentNet(in_channels, policy_channels, num_filters=width, num_blocks,
 *                      recall, value_activation)   (Architectures/RecurrentNet.py:18-99)
 *   NZ_ARCH_RESNET     ResNet(in_channels, policy_channels, num_filters=width, num_blocks,
 *                      batch_norm=False, value_activation)     (Architectures/ResNet.py:13-70)
 *   NZ_ARCH_CONVNET    ConvNet(in_channels, policy_channels, kernel_size, num_filters=width,
 *                      num_layers=num_blocks)  -- ELU trunk          (Architectures/ConvNet.py:12-57)
 * Weights are passed in the model's state_dict order. */
typedef struct nz_net_desc {
  int32_t in_channels;
  int32_t policy_channels;
  int32_t width;
  int32_t num_blocks;                   /* residual blocks; ConvNet: num_layers */
  int32_t recall;                       /* RecurrentNet only */
  int32_t value_activation;             /* NZ_ACT_* */
  int32_t arch;                         /* NZ_ARCH_* */
  int32_t kernel_size;                  /* ConvNet trunk: 1 or 3; others: 3.  With hex: the hexagdly size, 1 */
  int32_t hex;                          /* 1: every conv is hexagdly.Conv2d(kernel_size=1): centre + 6 neighbours on a
                                           grid whose odd columns sit half a cell lower; each conv then has TWO weight
                                           tensors, kernel0 [out][in][3][1] and kernel1 [out][in][2][2].  nz_boardnet_*
                                           only; restated from the hexagdly documentation, parity unpinned (the package
                                           is not installed here).  nz_engine_set_weights rejects it. */
} nz_net_desc;

/* Fixed per-engine sizes, for sizing the caller's export buffers. */
typedef struct nz_dims {
  int32_t n_games;                      /* games per self-play round */
  int32_t n_slots;                      /* games in flight at once */
  int32_t num_actions;                  /* A */
  int32_t max_moves;                    /* T */
  int32_t state_channels, rows, cols;   /* C, H, W */
  int32_t node_capacity;                /* tree arena nodes per game */
} nz_dims;

const char* nz_version(void);
/* Message of the last failure on `e` (or of the last failed nz_engine_create when e == NULL). */
const char* nz_last_error(const nz_engine* e);

/* ---- engine life cycle ---------------------------------------------------
 * Replaces: Gamer.__init__ (Training/Gamer.py:20-37), one per actor, and the
 * per-game `game_class(*game_args)` / `Node(0)` set-up (Gamer.py:52,59). */
nz_status nz_engine_create(nz_engine** out, const nz_search_cfg* cfg, const nz_game_desc* game,
                           int32_t n_games, int32_t device);
/* As nz_engine_create, with the concurrency separate from the round size: a
 * round of n_games games is played on n_slots (<= n_games) concurrent trees; a
 * slot whose game ends takes the next unplayed game, as the reference's
 * ActorPool of num_actors Gamers does for num_games_per_step games
 * (Training/AlphaZero.py:525-577).  nz_engine_create is n_slots == n_games. */
nz_status nz_engine_create_ex(nz_engine** out, const nz_search_cfg* cfg, const nz_game_desc* game,
                              int32_t n_slots, int32_t n_games, int32_t device);
void nz_engine_destroy(nz_engine* e);
nz_status nz_engine_dims(const nz_engine* e, nz_dims* out);

/* Hand the network to the engine.  Replaces `shared_storage.get` +
 * `network_copy.check_devices()` (Gamer.py:40,61-62) and selects what
 * Network_Manager.inference (Neural_Networks/Network_Manager.py:46-64) computes.
 * `weights[i]` are float32 tensors in the reference's state_dict order and
 * [C_out][C_in][3][3] layout (host or device memory); they are repacked for the
 * MFMA kernel and copied, so the caller may free them afterwards. */
nz_status nz_engine_set_weights(nz_engine* e, const nz_net_desc* net, const float* const* weights,
                                int32_t n_tensors, int32_t recurrent_iterations);

/* Test hook: evaluate leaves by table look-up instead of the network.
 * table[code][0..8] = post-softmax probabilities, table[code][9] = value,
 * code = sum(cell[a] * 3^a); 19683 rows, host or device memory.  Same meaning
 * as a cache hit in Explorer.evaluate (Explorer.py:146-149). */
nz_status nz_engine_set_table(nz_engine* e, const float* table, int32_t n_rows);

/* New batch of games: every game back to the initial position with an
 * unexpanded root (Gamer.py:52-59). */
nz_status nz_engine_reset(nz_engine* e, void* stream);

/* Children of each game's current root, 0 for an unexpanded root or a finished
 * game: the number of gamma draws add_exploration_noise will take
 * (Explorer.py:201-210).  dev int32[G]. */
nz_status nz_engine_root_children(nz_engine* e, int32_t* n_children_dev, void* stream);

/* One move for every live game: Explorer.run_mcts (Explorer.py:40-67:
 * root noise, mcts_simulations x {select, evaluate/expand, backup},
 * select_action) followed by game.step, store_search_statistics and re-rooting
 * (Gamer.py:65-79).
 *   noise_dev    double[G][A]: gamma draws for the root's children in child
 *                order (first n_children entries of a row are used); may be
 *                NULL when training == 0.
 *   uniforms_dev double[G][3]: epsilon_softmax, epsilon_random and the uniform
 *                np.random.choice consumes (Explorer.py:77-78,89,199); may be
 *                NULL when training == 0. */
nz_status nz_engine_move(nz_engine* e, const double* noise_dev, const double* uniforms_dev,
                         void* stream);

/* The two halves of nz_engine_move, for evaluation matches (Testing/Tester.py:46-121 with
 * Testing/Agents/Generic/MctsAgent.py:28-39):
 *   nz_engine_search  root noise (training only) + the simulations, no action yet;
 *   nz_engine_apply   select the action -- the search's own choice when actions_dev is NULL
 *                     (MctsAgent.choose_action), else actions_dev[g] (the opponent's move in
 *                     MctsAgent.update_subtree) -- then step, record and re-root;
 *   nz_engine_last_actions  the action each game took at its latest move (dev int32[G], -1 if none).
 * An agent that keeps its subtree searches on the opponent's turn too, so a match between two
 * MCTS agents is two engines that both search every ply and both apply the mover's action. */
nz_status nz_engine_search(nz_engine* e, const double* noise_dev, void* stream);
nz_status nz_engine_apply(nz_engine* e, const int32_t* actions_dev, const double* uniforms_dev, void* stream);
nz_status nz_engine_last_actions(nz_engine* e, int32_t* actions_dev, void* stream);

/* 1 for every game that is not yet terminal, else 0.  dev int32[G]. */
nz_status nz_engine_alive(nz_engine* e, int32_t* alive_dev, void* stream);

/* Number of games not yet terminal.  Synchronises `stream`. */
nz_status nz_engine_live_games(nz_engine* e, int32_t* n_live_host, void* stream);

/* Whole games with the engine's own host random streams: game g uses a private
 * legacy MT19937 stream seeded base_seed + g, consumed in the reference's order
 * (SURVEY.md appendix A rules 13-17).  Resets, then plays until every game is
 * terminal (Gamer.play_game, Gamer.py:39-97, for G games).  Synchronises. */
nz_status nz_engine_play(nz_engine* e, uint64_t base_seed, void* stream);
/* As nz_engine_play; with have_next the random numbers of the round that will be played next (seeds
 * next_base_seed + g) are drawn on the host threads while this round's kernel runs, and the next call with that
 * base seed uses them (any other call draws afresh: results never depend on the hint). */
nz_status nz_engine_play_next(nz_engine* e, uint64_t base_seed, int32_t have_next, uint64_t next_base_seed, void* stream);

/* Same games, same results, by the lock-step route: one kernel sequence per move
 * with a host round trip in between (the route nz_engine_move exposes).
 * nz_engine_play runs the persistent self-play kernel instead and falls back to
 * this route for the rare game whose pre-drawn randomness does not fit. */
nz_status nz_engine_play_lockstep(nz_engine* e, uint64_t base_seed, void* stream);
/* Games nz_engine_play had to replay by the lock-step route since creation. */
nz_status nz_engine_desync_count(const nz_engine* e, int64_t* count_host);

/* Results of the batch, the data ReplayBuffer.save_game reads from a game
 * (Training/ReplayBuffer.py:24-36) plus the per-move statistics Gamer reports
 * (Gamer.py:42-50,81-92).  Any pointer may be NULL.  T = max_moves.
 *   states    float [G][T][C][H][W]  game.state_history (zeros past the end)
 *   visits    int32 [G][T][A]        root child visit counts by action
 *                                    (child_policy = visits / sum, tic_tac_toe.py:177-182)
 *   actions   int32 [G][T]           chosen action, -1 past the end
 *   lengths   int32 [G]              game.length
 *   outcomes  int32 [G]              game.terminal_value
 *   tree_size int32 [G][T]           root.visit_count after the search (Gamer.py:71)
 *   n_children int32[G][T]           root.num_children()             (Gamer.py:72)
 *   bias      double[G][T]           final_root_bias                 (Explorer.py:63)
 */
nz_status nz_engine_export(nz_engine* e, float* states, int32_t* visits, int32_t* actions,
                           int32_t* lengths, int32_t* outcomes, int32_t* tree_size,
                           int32_t* n_children, double* bias, void* stream);

/* Parity trace of the root's children after each search, by action index:
 * prior (after noise), value_sum; plus root value_sum.  dev double[G][T][A],
 * double[G][T][A], double[G][T].  Any pointer may be NULL. */
nz_status nz_engine_export_trace(nz_engine* e, double* child_prior, double* child_value_sum,
                                 double* root_value_sum, void* stream);

/* Work counters since the last reset, summed over games: simulations and
 * node expansions (= network evaluations, Explorer.py:144-181).  Synchronises. */
nz_status nz_engine_counters(nz_engine* e, int64_t* simulations_host, int64_t* expansions_host,
                             void* stream);
/* As above plus the select work done: out[0] simulations, out[1] expansions,
 * out[2] internal nodes whose children were scored (descent levels), out[3]
 * children scored.  Writes exactly FOUR values. */
nz_status nz_engine_counters_ex(nz_engine* e, int64_t* out4_host, void* stream);
/* The first n_out (1..5) of: the four above, then out[4] nodes created by
 * expansions.  Used to price the tree phase's algorithmic bytes (SURVEY.md
 * section 8d).  The caller states its array's length, so later additions never
 * write past an older caller's array. */
nz_status nz_engine_counters_n(nz_engine* e, int64_t* out_host, int32_t n_out, void* stream);

/* Algorithmic FLOPs of one network evaluation (one position): sum over convs of
 * 2 * C_out * C_in * 49, i.e. only the taps that fall inside the 3x3 board. */
nz_status nz_engine_net_flops(const nz_engine* e, double* flops_host);
/* What the matrix cores execute per position for it: the fused kernel forms each float32
 * product from six bf16 MFMA terms on exact three-way splits (bf16_flops, including the
 * zero-padded channels of 32-channel K groups) and uses float32 MFMAs for the raw input
 * planes (f32_flops). */
nz_status nz_engine_net_matrix_flops(const nz_engine* e, double* bf16_flops_host, double* f32_flops_host);

/* ---- stand-alone operators (same kernels, callable by themselves) ---------
 * Network_Manager.inference for a batch (Network_Manager.py:46-64):
 * states float[B][C][3][3] -> logits float[B][P*9], value float[B]; and the
 * softmax Explorer.evaluate applies (Explorer.py:159) -> probs float[B][P*9]
 * (may be NULL). */
nz_status nz_net_forward(nz_engine* e, const float* states_dev, int32_t batch, float* logits_dev,
                         float* value_dev, float* probs_dev, void* stream);

/* Diagnostic build of the network kernel: mean shader-clock ticks wave 0 of a
 * workgroup spends in [0] the K-group loops (LDS reads + MFMA), [1] the
 * input-plane step, [2] the epilogues, [3] at the stage barriers.  Synchronises. */
nz_status nz_net_forward_stamps(nz_engine* e, const float* states_dev, int32_t batch, float* logits_dev,
                                float* value_dev, double* ticks4_host);

/* Timing of the engine's own kernels, measured with HIP events on the stream
 * the kernels run on.  Enable, run, then read: total milliseconds and launch
 * count per kernel class (0 = search: the persistent self-play kernel or the lock-step
 * advance kernel, 1 = stand-alone network kernel, 2 = move/noise/reset/export). */
nz_status nz_engine_profile(nz_engine* e, int32_t enable);
nz_status nz_engine_profile_read(nz_engine* e, double* ms_host /*[3]*/, int64_t* launches_host /*[3]*/,
                                 int64_t* net_positions_host);

/* Diagnostic build of the persistent kernel: with `enable` the next
 * nz_engine_play runs the stamped variant (shader-clock ticks per phase).  If
 * out4_host is not NULL it first receives the last stamped run's figures:
 * [0] mean tree/net cycles per workgroup, [1] share of ticks in the tree phase,
 * [2] share in the network phase, [3] mean / max workgroup lifetime (how evenly
 * the workgroups finish), [4] shader-clock ticks per network phase, [5] per tree
 * phase, [6] ticks of the longest-lived workgroup, [7] workgroups, [8] wave 0's
 * ticks per cycle in end-of-move bookkeeping, [9] in the pending expansion.  Run
 * times of the stamped build are not quoted. */
nz_status nz_engine_phase_stamps(nz_engine* e, int32_t enable, double* out10_host);

/* ---- SCS rules as batch operators ------------------------------------------
 * Device-side state transition, legal-move mask and state image of the SCS hex
 * war-game (Games/SCS/SCS_Game.py: step :375-391, possible_actions :395-484,
 * update_game_env :687-831, resolve_combat :997-1044, generate_state :1348-1505),
 * one independent game per batch row.  The search is not built on them yet; they
 * exist so that the rules can be checked against the oracle step by step.
 * A game is described by plain arrays (what load_game_from_config, :1570-1779,
 * reads from the YAML file, "Detailed" map / victory points only):
 *   terrain  float[rows*cols][3]  attack modifier, defense modifier, movement cost
 *   vp       int32[n_vp[0]+n_vp[1]][2]  (row, col), player one's points first
 *   units    int32[n_units][5]  player (0/1), arrival turn, attack, defense, movement;
 *            in schedule order: player one's turns 0..T, then player two's
 *   arrival  uint8[n_units][rows*cols]  1 where the unit may be placed            */
typedef struct nz_scs nz_scs;
typedef struct nz_scs_desc {
  int32_t rows, cols, turns, stacking;
  const float* terrain;
  int32_t n_vp[2];
  const int32_t* vp;
  int32_t n_units;
  const int32_t* units;
  const uint8_t* arrival;
} nz_scs_desc;
nz_status nz_scs_create(nz_scs** out, const nz_scs_desc* desc, int32_t n_games, int32_t device);
void nz_scs_destroy(nz_scs* h);
const char* nz_scs_last_error(const nz_scs* h);
nz_status nz_scs_dims(const nz_scs* h, int32_t* planes, int32_t* rows, int32_t* cols, int32_t* channels);
nz_status nz_scs_reset(nz_scs* h, void* stream);
/* actions_dev int32[G]: flat action index (plane, row, col) or -1 to leave the game as it is */
nz_status nz_scs_step(nz_scs* h, const int32_t* actions_dev, void* stream);
/* mask_dev int8[G][planes*rows*cols]; all zero for a finished game */
nz_status nz_scs_legal_mask(nz_scs* h, int8_t* mask_dev, void* stream);
/* image_dev float[G][channels][rows][cols] = generate_state() */
nz_status nz_scs_state_image(nz_scs* h, float* image_dev, void* stream);
/* status_dev int32[G][7]: player, sub_phase, stage, turn, terminal, terminal_value, length */
nz_status nz_scs_status(nz_scs* h, int32_t* status_dev, void* stream);

/* ---- MCTS self-play on SCS, lock-step, evaluations supplied by the caller -----
 * Explorer.run_mcts / Gamer.play_game for G SCS games with the tree, the rules and the
 * legal-move masks on the device; the network stays outside: nz_scs_search_select hands
 * out the state images of the leaves that need an evaluation (what Explorer.evaluate
 * passes to Network_Manager.inference, Explorer.py:145-158) and nz_scs_search_expand
 * takes (softmax probabilities, value) back (Explorer.py:159-181).  Any PyTorch model
 * can therefore drive the search (hex or square convs) until the fused kernel covers
 * SCS boards.  One move for all live games:
 *   root_children -> host draws gamma noise -> begin_move(noise [G][C], C = nz_scs_search_limits' max_children)
 *   repeat: select(images [G][C][R][Cc], leaf_game [G], &n) ; if n == 0 break ;
 *           evaluate images[0..n) ; expand(probs [n][A], value [n])
 *   end_move(uniforms [G][3])
 * nodes_per_game sizes each game's tree arena: two halves; at every re-root the new root's subtree is copied
 * into the other half (the rest of the tree is dropped), so a half must hold the kept subtree plus one move's
 * expansions (32 bytes per node; a full half is reported as NZ_ERR_OVERFLOW).  nodes_per_game <= 0: the default,
 * 2 x (1 + simulations x 2.5 x the children bound). */
typedef struct nz_scs_search nz_scs_search;
nz_status nz_scs_search_create(nz_scs_search** out, const nz_scs_desc* desc, const nz_search_cfg* cfg,
                               int32_t n_games, int32_t nodes_per_game, int32_t device);
void nz_scs_search_destroy(nz_scs_search* h);
const char* nz_scs_search_last_error(const nz_scs_search* h);
nz_status nz_scs_search_reset(nz_scs_search* h, void* stream);
nz_status nz_scs_search_root_children(nz_scs_search* h, int32_t* n_children_dev, void* stream);
nz_status nz_scs_search_begin_move(nz_scs_search* h, const double* noise_dev, void* stream);
/* Synchronises; *n_leaves_host = number of leaves queued (0: every game's search is complete). */
nz_status nz_scs_search_select(nz_scs_search* h, float* images_dev, int32_t* leaf_game_dev,
                               int32_t* n_leaves_host, void* stream);
nz_status nz_scs_search_expand(nz_scs_search* h, const float* probs_dev, const float* value_dev, void* stream);
nz_status nz_scs_search_end_move(nz_scs_search* h, const double* uniforms_dev, void* stream);
/* status_dev int32[G][7] as nz_scs_status */
nz_status nz_scs_search_status(nz_scs_search* h, int32_t* status_dev, void* stream);
/* Per move m < M of game g (dev pointers, any may be NULL; M = max_moves, C = max_children of
 * nz_scs_search_limits): actions int32[G][M] (-1 past the end), tree_size, n_children int32[G][M], bias,
 * root_value_sum double[G][M]; the root's children in child order: child_action, child_visit int32[G][M][C],
 * child_prior, child_value_sum double[G][M][C].  counters_host int64[2]: simulations, expansions. */
nz_status nz_scs_search_export(nz_scs_search* h, int32_t* actions, int32_t* tree_size, int32_t* n_children,
                               double* bias, double* root_value_sum, int32_t* child_action,
                               int32_t* child_visit, double* child_prior, double* child_value_sum,
                               int64_t* counters_host, void* stream);

/* ---- policy/value network on boards of any size (SCS maps) -------------------
 * The square-conv (hex=False) RecurrentNet / ResNet / ConvNet of the reference
 * (Neural_Networks/Architectures/RecurrentNet.py:18-99, ResNet.py:13-70, ConvNet.py:12-57,
 * blocks.py) on rows x cols boards, evaluated on a batch of state images: what
 * Network_Manager.inference (Network_Manager.py:46-64) plus the softmax of
 * Explorer.evaluate (Explorer.py:158-162) compute for one position, for n positions.
 * Every convolution is an implicit-GEMM FP32 MFMA kernel; activations stay on the device.
 *   images_dev  float[n][in_channels][rows][cols]   (Game.generate_network_input)
 *   logits_dev  float[n][policy_channels*rows*cols] raw policy logits, may be NULL
 *   probs_dev   float[n][policy_channels*rows*cols] softmax over ALL logits, may be NULL
 *   value_dev   float[n]
 *   n_dev       optional device int32: the live batch size (<= n) when the host does not
 *               know it yet (the leaf count of a simulation wave); NULL = n. */
typedef struct nz_boardnet nz_boardnet;
nz_status nz_boardnet_create(nz_boardnet** out, const nz_net_desc* net, int32_t rows, int32_t cols,
                             int32_t max_batch, int32_t device);
void nz_boardnet_destroy(nz_boardnet* h);
const char* nz_boardnet_last_error(const nz_boardnet* h);
/* weights: the state_dict tensors in order (float32, device or host), as nz_engine_set_weights */
nz_status nz_boardnet_set_weights(nz_boardnet* h, const float* const* weights, int32_t n_weights,
                                  int32_t recurrent_iterations);
nz_status nz_boardnet_forward(nz_boardnet* h, const float* images_dev, int32_t n, const int32_t* n_dev,
                              float* logits_dev, float* probs_dev, float* value_dev, void* stream);
/* The same without the layout conversion: the network's own input buffer is rows of `row_stride` floats, channels
 * contiguous (zero beyond in_channels: never write there), row = ((position / 16) * rows*cols + cell) * 16 +
 * position % 16.  A producer (the SCS search kernel) writes the positions' planes there directly, then calls
 * nz_boardnet_forward_rows. */
nz_status nz_boardnet_input_rows(nz_boardnet* h, float** rows_dev, int32_t* row_stride);
nz_status nz_boardnet_forward_rows(nz_boardnet* h, int32_t n, const int32_t* n_dev, float* logits_dev, float* probs_dev,
                                   float* value_dev, void* stream);
/* algorithmic FLOPs of one position (taps that fall off the board are not counted) */
int64_t nz_boardnet_flops(const nz_boardnet* h);
/* Narrow nets on small boards run all layers, the softmax and the value in ONE launch with the activations in LDS
 * (when a workgroup's share of max_batch fits; same floats as the per-layer kernels).  enable: 1 / 0 switch it on /
 * off (the A/B of the parity test), -1 leaves it; *available (may be NULL): whether the one-launch form was built. */
nz_status nz_boardnet_fused(nz_boardnet* h, int32_t enable, int32_t* available);
nz_status nz_boardnet_dims(const nz_boardnet* h, int32_t* in_channels, int32_t* policy_channels, int32_t* rows,
                           int32_t* cols, int32_t* max_batch);

/* Gamer.play_game (Training/Gamer.py:52-92) for all G SCS games of `h` to the end, tree, rules,
 * masks AND network on the device: one simulation wave = one [expand + select] kernel followed
 * by nz_boardnet_forward on the wave's leaves, whose count stays in device memory.  The host
 * draws each move's random numbers (game g uses numpy RandomState(seeds_host[g]) in the
 * reference's order) and polls for the end of the move's searches every few waves.
 * Synchronises.  Read the games with nz_scs_search_export / nz_scs_search_status. */
nz_status nz_scs_search_play(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, void* stream);
/* As nz_scs_search_play, stopping after `max_moves` decisions of every game (<= 0: play to the end): the
 * first moves of a game are exactly those of the full game (Gamer.py:64-79 has no look-ahead), so this
 * bounds a check on a large board / a heavy network without changing what is checked. */
nz_status nz_scs_search_play_moves(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, int32_t max_moves,
                                   void* stream);

/* A round of n_round >= n_games games over the engine's n_games slots: a slot whose game has ended hands its records to
 * the round's store and starts the round's next game, as a Gamer actor plays its games back to back
 * (Training/Gamer.py:45-98) -- the round ends with its longest game instead of idling the slots of short ones.  Game i
 * is seeded with seeds_host[i] (numpy RandomState(seed), the reference's draw order) whichever slot plays it, so a
 * round's games do not depend on the number of slots.  n_round == n_games is nz_scs_search_play. */
nz_status nz_scs_search_play_round(nz_scs_search* h, nz_boardnet* net, const uint32_t* seeds_host, int64_t n_round,
                                   void* stream);
/* nz_scs_search_export for the last round of more games than slots: every array has n_round rows;
 * status2 [n_round][2] = (length, terminal value).  Device pointers, any may be null. */
nz_status nz_scs_search_export_round(nz_scs_search* h, int32_t* actions, int32_t* tree_size, int32_t* n_children,
                                     double* bias, double* root_value_sum, int32_t* child_action, int32_t* child_visit,
                                     double* child_prior, double* child_value_sum, int32_t* status2,
                                     int64_t* counters_host, void* stream);
/* Diagnostic (library built with -DNZ_SCS_STAMPS, zeros otherwise): shader ticks summed over games and waves since
 * the last reset; out6: expansion, rules copy, scratch clone, descent, leaf mask + image, terminal simulations. */
nz_status nz_scs_search_phase_ticks(nz_scs_search* h, int64_t* out6_host);
/* Evaluation matches on SCS (Testing/Agents/Generic/MctsAgent.py:28-39, Testing/Tester.py:46-121), as
 * nz_engine_search / _apply / _last_actions for Tic-Tac-Toe: after a move's search (begin_move + select / expand until
 * no leaf is left) nz_scs_search_apply plays actions_dev[g] (int32 [G]; NULL or -1: the search's own choice,
 * like nz_scs_search_end_move), steps and re-roots; nz_scs_search_last_actions gives each game's latest action. */
nz_status nz_scs_search_apply(nz_scs_search* h, const int32_t* actions_dev, const double* uniforms_dev, void* stream);
nz_status nz_scs_search_last_actions(nz_scs_search* h, int32_t* actions_dev, void* stream);

/* The reference's optional inference cache (Explorer.py:146-155; Utils/Caches/KeylessCache.py:24-160, DictCache.py:4-85;
 * handed to Gamer.play_game by AlphaZero.py:560-577) for nz_scs_search_play: one keyless hash table in HBM shared by the
 * games of the engine -- a leaf whose state was evaluated before takes (probs, value) from the table instead of the
 * network.  Results-neutral: a position's evaluation does not depend on the batch it was computed in.
 *   max_entries > 0: (re)allocate an empty table of the largest power of two <= max_entries (KeylessCache.py:27-38);
 *   0: no cache; < 0: empty the table (a new self-play round: the reference builds new Gamers, hence new caches).
 * stats out4: hits, misses, entries in use, table size. */
nz_status nz_scs_search_cache(nz_scs_search* h, int64_t max_entries);
nz_status nz_scs_search_cache_stats(nz_scs_search* h, int64_t* out4_host);
/* What the per-move records hold, bounded from the game description at create (the reference itself has no limits,
 * SCS_Game.py:395-484): decisions per game (= the M of nz_scs_search_export's [G, M] / [G, M, C] arrays) and children
 * per node (= C, a multiple of 64, at most 256; a description that allows more is rejected by nz_scs_search_create). */
nz_status nz_scs_search_limits(const nz_scs_search* h, int32_t* max_moves, int32_t* max_children);
/* simulation waves (kernel rounds) the last nz_scs_search_play took (persistent route: one launch per move) */
nz_status nz_scs_search_waves(const nz_scs_search* h, int64_t* waves);
/* The persistent route of nz_scs_search_play / _play_moves / _play_round: one wavefront per game runs the whole
 * search of a move -- descent, rules, the network for its own leaf, expansion, backup (Explorer.run_mcts,
 * Search/Explorer.py:40-67, with one simulation in flight per tree, :49-61) -- in ONE launch per move, no game waiting
 * for another.  Available for ConvNet / ResNet board nets on boards of up to 32 cells whose layers are at most 64
 * channels wide, with the inference cache off; other configurations keep the wave-by-wave route.
 * enable: 1 require it (a play fails where it is not available), 0 never, -1 the default (use it where available).
 * *used (may be NULL): whether the last play ran on it. */
nz_status nz_scs_search_persistent(nz_scs_search* h, int32_t enable, int32_t* used);
/* Every game on its OWN map: the reference builds a new game object per game (Training/Gamer.py:52) and a "Randomized"
 * config draws terrain and victory points from numpy's global stream when that object is built (SCS_Game.py:1678-1738;
 * what most of the reference's SCS presets train on, Run.py:115).  Host arrays for n games (n >= the games of a round;
 * game i of a round reads row i whichever slot plays it): terrain float32 [n][tiles][3] (attack modifier, defense
 * modifier, cost), vp int32 [n][n_vp0 + n_vp1][2] (row, column), the counts as in the description passed to
 * nz_scs_search_create (its own map is then only a template: bound the limits with the cheapest terrain).  mt_keys
 * uint32 [n][624] + mt_pos int32 [n] (both or neither): the numpy RandomState state each game's stream goes on from --
 * where the map's draws left it, so that one stream serves map and search as in `np.random.seed(s); SCS_Game(cfg); play`;
 * the `seeds` of the play calls are then ignored (may be NULL).  n = 0: back to the description's one map.  Resets. */
nz_status nz_scs_search_set_games(nz_scs_search* h, int64_t n, const float* terrain_host, const int32_t* vp_host,
                                  const uint32_t* mt_keys_host, const int32_t* mt_pos_host);
/* As above for the rule operators: every game of the batch on its own map (NULL terrain: the description's).  Resets. */
nz_status nz_scs_set_maps(nz_scs* h, const float* terrain_host, const int32_t* vp_host, void* stream);
/* HIP-event timing of the persistent kernel, on the stream it is launched on.  enable: 1 on (sums zeroed), 0 off, -1 leave.
 * out4_host (may be NULL): milliseconds summed over launches, launches, v_mfma_f32_16x16x32_bf16 instructions issued
 * per evaluated position, algorithmic float32 FLOPs per position. */
nz_status nz_scs_search_persist_profile(nz_scs_search* h, int32_t enable, double* out4_host);
/* Test hook of the persistent route: keep the leaf evaluations of n games (games_host: indices), up to `capacity`
 * each, in the order the game's search consumed them; n = 0 stops.  nz_scs_search_record_read (host pointers, any of the
 * three arrays may be NULL): *count evaluations consumed, digests uint64 [.][2] (a 128-bit mix of the leaf's float32
 * planes), probs float32 [.][A] post-softmax, values float32 [.]; rows = min(*count, capacity). */
/* Diagnostic (library built with -DNZ_PERSIST_STAMPS; zeros otherwise): shader ticks of the persistent route summed over
 * games and moves since the last reset: clone, descent, legal mask + list, planes + split, network, softmax + value,
 * expansion, backup, whole moves, the slowest single (game, move); inside the network (leader's half): K loops,
 * epilogues, waits for the helper.  13 values. */
nz_status nz_scs_search_persist_ticks(nz_scs_search* h, int64_t* out13_host);
/* Diagnostic: the persistent route's network alone -- `blocks` workgroups of four (leader, helper) wavefront pairs each
 * run `iters` passes on an all-zero input; ticks_host[blocks * 4] = shader ticks per pass. */
nz_status nz_scs_netbench(nz_boardnet* net, int32_t blocks, int32_t iters, uint64_t* ticks_host);
/* Diagnostic (-DNZ_WIDE_STAMPS builds of the library only, NZ_ERR_STATE otherwise): phase ticks of conv_wide_kernel's K
 * steps summed over its launches so far -- [0] loads issued, [1] first fragments read, [2] MFMAs + staging, [3] barrier,
 * [4] loop overhead, [5] K steps, [6] launches, [7] 0.  No reference counterpart. */
nz_status nz_boardnet_wide_stamps(uint64_t* out8);
nz_status nz_scs_search_record(nz_scs_search* h, const int32_t* games_host, int32_t n, int32_t capacity);
nz_status nz_scs_search_record_read(nz_scs_search* h, int32_t slot, int32_t* count, uint64_t* digests_host,
                                    float* probs_host, float* values_host);

/* ---- host random streams (numpy legacy RandomState, MT19937) --------------
 * Replaces the reference's use of the global np.random stream
 * (Explorer.py:77-78,89,199,208). */
nz_rng* nz_rng_create(uint32_t seed);
/* a stream that goes on where a numpy RandomState stands (RandomState.get_state(): key[624], pos, has_gauss, cached_gaussian) */
nz_rng* nz_rng_create_state(const uint32_t* key624, int32_t pos, int32_t has_gauss, double cached_gaussian);
nz_rng* nz_rng_clone(const nz_rng* r);
void nz_rng_destroy(nz_rng* r);
void nz_rng_seed(nz_rng* r, uint32_t seed);
uint32_t nz_rng_u32(nz_rng* r);
double nz_rng_double(nz_rng* r);                                  /* random_sample() */
void nz_rng_gamma(nz_rng* r, double shape, double scale, int32_t n, double* out);  /* gamma(shape, scale, n) */

/* ---- replay buffer on the device ------------------------------------------
 * The positions of finished games stay in HBM; which physical slot a position takes and which slots a batch reads
 * is decided by the caller (host logic of Training/ReplayBuffer.py:24-53: window in games with per-position
 * eviction, shuffle, slice, sample), the library moves the data:
 *   nz_replay_append  replaces the tuple building of ReplayBuffer.save_game (ReplayBuffer.py:31-36) together with
 *                     Game.store_search_statistics / make_target (tic_tac_toe.py:177-190, SCS_Game.py:1517-1528):
 *                     row r of the source (state + the root's visit counts, or ready-made policies, or per-child
 *                     (action, visit) lists) becomes the position in slot dst_slot[r] (-1: not stored); policy =
 *                     visit / sum(visits) in double, rounded to float32 as torch.tensor() does (AlphaZero.py:901);
 *                     value = game_value[r / rows_per_game].
 *   nz_replay_gather  replaces batch assembly (AlphaZero.py:846-852,892-903: torch.cat of the states, one
 *                     torch.tensor per target): states [B, state_floats], policies [B, A], values [B], game_index [B]
 *                     for the physical slots of a batch, in the order given (the caller groups by game index).
 * All pointers are device pointers; work is enqueued on `stream`. */
typedef struct nz_replay nz_replay;
nz_status nz_replay_create(nz_replay** out, int64_t capacity_positions, int32_t state_floats, int32_t num_actions,
                           int32_t device);
void nz_replay_destroy(nz_replay* h);
const char* nz_replay_last_error(const nz_replay* h);
nz_status nz_replay_dims(const nz_replay* h, int64_t* capacity, int32_t* state_floats, int32_t* num_actions);
/* exactly one of visits_dev [N, A] int32 / policies_dev [N, A] float32 / (child_action_dev, child_visit_dev
 * [N, max_children] int32 + n_children_dev [N]) */
nz_status nz_replay_append(nz_replay* h, const float* states_dev, const int32_t* visits_dev, const float* policies_dev,
                           const int32_t* child_action_dev, const int32_t* child_visit_dev, const int32_t* n_children_dev,
                           int32_t max_children, const int32_t* game_value_dev, int32_t rows_per_game,
                           const int64_t* dst_slot_dev, int64_t n_rows, int32_t game_index, void* stream);
nz_status nz_replay_gather(nz_replay* h, const int64_t* slots_dev, int64_t batch, float* states_out, float* policies_out,
                           float* values_out, int32_t* game_index_out, void* stream);
/* synchronises; NZ_ERR_OVERFLOW if a kernel saw a slot or an action out of range */
nz_status nz_replay_check(nz_replay* h, void* stream);

/* ---- batched loss ---------------------------------------------------------------
 * AlphaZero.calculate_loss (Training/AlphaZero.py:891-921) for a whole batch in one launch, with the gradients of the
 * combined loss: losses3 = (value_loss, policy_loss, combined_loss); dlogits [B, A], dvalues [B] may be NULL.
 * policy_loss: cross entropy with label smoothing 0.02 (AlphaZero.py:327), KL divergence, masked MSE
 * (Utils/Functions/loss_functions.py:7-26); value_loss: squared / absolute error (loss_functions.py:28-33);
 * normalize_policy: divide the policy loss by log(batch) as the reference does (AlphaZero.py:912-915).
 * workspace_dev: 2 * batch floats.  float32 arithmetic; results agree with the reference's to ~1e-6 relative.
 * Masked MSE with a target row of zeros only (the reference raises ZeroDivisionError there): losses3[1] and [2] come out
 * NaN, that sample's dlogits row is zeros. */
enum { NZ_LOSS_CE = 0, NZ_LOSS_KLD = 1, NZ_LOSS_MSE = 2 };
enum { NZ_LOSS_SE = 0, NZ_LOSS_AE = 1 };
nz_status nz_loss_forward_backward(const float* logits_dev, const float* values_dev, const float* target_policies_dev,
                                   const float* target_values_dev, int32_t batch, int32_t actions, int32_t policy_loss,
                                   int32_t value_loss, int32_t normalize_policy, float* losses3_dev, float* dlogits_dev,
                                   float* dvalues_dev, float* workspace_dev, void* stream);
const char* nz_loss_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NUZERO_AMD_H */
