// Host side of the C ABI (include/nuzero_amd.h): buffer ownership, the per-move
// launch sequence, weight packing for the MFMA kernel and the host random
// streams' call order.  No compute happens here.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

using namespace nz;

namespace {
thread_local std::string g_create_error;

struct ProfileSpan {
  hipEvent_t a, b;
  int cls;
};
}  // namespace

struct nz_engine {
  nz_search_cfg cfg;
  nz_game_desc game;
  int device = 0;
  int n_games = 0;   // games per round
  int n_slots = 0;   // games in flight
  int cap = 0;
  int tab_len = 0;
  TreeParams tp;
  std::vector<void*> allocs;
  std::string error;
  // network
  bool have_net = false, have_table = false;
  nz_net_desc net;
  int iters = 0;
  NetProgram prog_host;
  NetProgram* prog_dev = nullptr;
  float* weights_dev = nullptr;
  float* table_dev = nullptr;
  float* leaf_logits = nullptr;
  float* leaf_value = nullptr;
  double executed_bf16_flops_per_position = 0.0, executed_f32_flops_per_position = 0.0;
  double algorithmic_flops_per_position = 0.0;
  // host staging for nz_engine_play
  int32_t* h_children = nullptr;   // pinned [G]
  int32_t* h_alive = nullptr;      // pinned [G]
  double* h_noise = nullptr;       // pinned [G][A]
  double* h_uniforms = nullptr;    // pinned [G][3]
  double* d_noise = nullptr;
  double* d_uniforms = nullptr;
  std::vector<nz_rng*> rngs;
  // persistent-kernel path: whole-game randomness, [G][T][A] and [G][T][3]
  double* h_game_noise = nullptr;     // pinned
  double* h_game_uniforms = nullptr;  // pinned
  double* d_game_noise = nullptr;
  double* d_game_uniforms = nullptr;
  int32_t* h_desync = nullptr;        // pinned [G]
  // randomness of the NEXT round, drawn on host threads while this round's kernel runs (nz_engine_play_next)
  double* h_next_noise = nullptr;     // pinned [G][T][A]
  double* h_next_uniforms = nullptr;  // pinned [G][T][3]
  bool next_ready = false;
  uint64_t next_seed = 0;
  bool borrowed_net = false;          // fallback engine: network buffers belong to the parent
  nz_engine* fallback = nullptr;
  int64_t desync_total = 0;
  unsigned long long* d_stamps = nullptr;   // [blocks][4], diagnostic build only
  bool stamps = false;
  // profiling
  bool profile = false;
  std::vector<ProfileSpan> spans;
  int64_t net_positions = 0;       // upper bound: positions offered to the network kernel
};

namespace {

nz_status fail(nz_engine* e, nz_status code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (e) e->error = buf;
  else g_create_error = buf;
  return code;
}

#define NZ_HIP(e, call)                                                                          \
  do {                                                                                           \
    hipError_t err__ = (call);                                                                   \
    if (err__ != hipSuccess)                                                                     \
      return fail((e), NZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
  } while (0)

template <typename T>
nz_status dev_alloc(nz_engine* e, T** out, size_t count) {
  void* p = nullptr;
  NZ_HIP(e, hipMalloc(&p, count * sizeof(T)));
  e->allocs.push_back(p);
  *out = static_cast<T*>(p);
  return NZ_OK;
}

struct Span {
  nz_engine* e;
  hipStream_t s;
  ProfileSpan sp{};
  bool on;
  Span(nz_engine* e_, hipStream_t s_, int cls) : e(e_), s(s_), on(e_->profile) {
    if (!on) return;
    sp.cls = cls;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(sp.a, s);
  }
  ~Span() {
    if (!on) return;
    (void)hipEventRecord(sp.b, s);
    e->spans.push_back(sp);
  }
};

int head_channel(int width, int out, int n_layers, int idx) {   // blocks.py:56-66,144-153
  const double step = (double)(out - width) / n_layers;
  double prev = width;
  int c = width;
  for (int i = 0; i < idx; ++i) {
    prev += step;
    c = (int)prev;
  }
  return c;
}

// One conv tensor repacked for the MFMA kernel (net_dev.hpp): per 16-channel output
// tile nt a contiguous weight stream main[nt][kgroup of 32 ch][tap][piece][lane][8 bf16] (the
// three exact bf16 pieces of every weight) and, for the two convs that also read the raw
// input planes, extra[nt][tap][lane] in float32.
struct PackedConv {
  int cout, cin, cin_main, kgroups, ntiles, extra;
  int32_t w_off, wx_off;
};
// a = p[0] + p[1] + p[2] exactly, each piece a bf16 number (the next 8 significant bits, truncated)
static void split3(float a, uint16_t p[3]) {
  float r = a;
  for (int i = 0; i < 3; ++i) {
    uint32_t bits;
    memcpy(&bits, &r, 4);
    bits &= 0xFFFF0000u;
    p[i] = (uint16_t)(bits >> 16);
    float t;
    memcpy(&t, &bits, 4);
    r = r - t;
  }
}
PackedConv pack_conv(const float* w, int cout, int cin, int cin_main, std::vector<float>& out) {
  PackedConv pc{};
  pc.cout = cout; pc.cin = cin; pc.cin_main = cin_main;
  pc.ntiles = (cout + 15) / 16;
  pc.kgroups = (cin_main + NET_KG_CHANNELS - 1) / NET_KG_CHANNELS;
  pc.extra = cin > cin_main ? 1 : 0;
  while (out.size() % 4) out.push_back(0.f);
  pc.w_off = (int32_t)out.size();
  auto weight = [&](int co, int ci, int t) { return co < cout && ci < cin_main ? w[((size_t)co * cin + ci) * 9 + t] : 0.f; };
  for (int nt = 0; nt < pc.ntiles; ++nt)
    for (int kg = 0; kg < pc.kgroups; ++kg)
      for (int t = 0; t < 9; ++t)
        for (int piece = 0; piece < 3; ++piece)
          for (int lane = 0; lane < 64; ++lane)
            for (int pr = 0; pr < 4; ++pr) {       // lane holds channels 32 kg + 8 (lane >> 4) + 0..7 as 4 bf16 pairs
              uint16_t lo[3], hi[3];
              const int co = nt * 16 + (lane & 15), ci = kg * 32 + (lane >> 4) * 8 + 2 * pr;
              split3(weight(co, ci, t), lo);
              split3(weight(co, ci + 1, t), hi);
              const uint32_t word = (uint32_t)lo[piece] | ((uint32_t)hi[piece] << 16);
              float f;
              memcpy(&f, &word, 4);
              out.push_back(f);
            }
  pc.wx_off = (int32_t)out.size();
  if (pc.extra)
    for (int nt = 0; nt < pc.ntiles; ++nt)
      for (int t = 0; t < 9; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = nt * 16 + (lane & 15);
          const int ci = cin_main + (lane >> 4);
          out.push_back(co < cout && ci < cin ? w[((size_t)co * cin + ci) * 9 + t] : 0.f);
        }
  return pc;
}

// A stage of the network = convs that may run side by side.  Its units (conv, output tile,
// output-cell group) are dealt to the eight waves longest-first; every wave's last job of the
// stage carries the barrier.
struct StageConv {
  int tensor, src, dst, res, act;
  bool split;   // each output tile may be cut into output-cell groups (halves or quarters) to give every wave a unit
};
const int og_taps[7] = {49, 13, 12, 12, 12, 26, 23};   // (input cell, tap) pairs per output-cell group (net_dev.hpp og_mask)
bool add_stage(NetProgram& pg, const std::vector<PackedConv>& convs, const std::vector<StageConv>& stage) {
  struct Unit { NetJob job; int cost; };
  std::vector<Unit> units;
  for (const StageConv& sc : stage) {
    const PackedConv& pc = convs[sc.tensor];
    for (int nt = 0; nt < pc.ntiles; ++nt) {
      if (!sc.split || pc.ntiles >= NET_WAVES_HOST) return false;   // whole-tile jobs (group 0) are not compiled in
      const int pieces = 2 * pc.ntiles >= NET_WAVES_HOST ? 2 : 4;
      const int og_first = pieces == 1 ? 0 : pieces == 2 ? 5 : 1, og_last = pieces == 1 ? 0 : pieces == 2 ? 6 : 4;
      for (int og = og_first; og <= og_last; ++og) {
        NetJob j{};
        j.w_off = pc.w_off + nt * pc.kgroups * NET_KG_DWORDS;
        j.wx_off = pc.wx_off + nt * 9 * 64;
        j.kgroups = (int16_t)pc.kgroups;
        j.nt = (int16_t)nt;
        j.extra = (int8_t)pc.extra;
        j.og = (int8_t)og;
        j.src = (int8_t)sc.src; j.dst = (int8_t)sc.dst; j.res = (int8_t)sc.res; j.act = (int8_t)sc.act;
        units.push_back({j, og_taps[og] * (3 * pc.kgroups + pc.extra)});   // ~MFMA time in units of 32 cycles
      }
    }
  }
  std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.cost > b.cost; });
  int load[NET_WAVES_HOST] = {};
  int first[NET_WAVES_HOST];
  for (int w = 0; w < NET_WAVES_HOST; ++w) first[w] = pg.n_jobs[w];
  for (const Unit& u : units) {
    int w = 0;
    for (int i = 1; i < NET_WAVES_HOST; ++i)
      if (load[i] < load[w]) w = i;
    if (pg.n_jobs[w] >= NET_MAX_JOBS) return false;
    pg.jobs[w][pg.n_jobs[w]++] = u.job;
    load[w] += u.cost;
  }
  for (int w = 0; w < NET_WAVES_HOST; ++w) {
    if (pg.n_jobs[w] == first[w]) {            // nothing to do in this stage: barrier only
      if (pg.n_jobs[w] >= NET_MAX_JOBS) return false;
      NetJob j{};
      j.og = OG_NONE;
      pg.jobs[w][pg.n_jobs[w]++] = j;
    }
    pg.jobs[w][pg.n_jobs[w] - 1].stage_end = 1;
  }
  return true;
}

hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

nz_status check_device_flag(nz_engine* e, hipStream_t s) {
  int32_t flag = 0;
  NZ_HIP(e, hipMemcpyAsync(&flag, e->tp.error_flag, sizeof(flag), hipMemcpyDeviceToHost, s));
  NZ_HIP(e, hipStreamSynchronize(s));
  if (flag != 0)
    return fail(e, NZ_ERR_OVERFLOW, "device check failed (flag %d: 1 = tree arena full, 2 = visit table too short, "
                                    "4 = move finished before its search, 8 = forced action is not legal, "
                                    "64 = persistent kernel gave up after its cycle bound)", flag);
  return NZ_OK;
}

}  // namespace

extern "C" {

const char* nz_version(void) { return "nuzero_amd 0.1 (gfx950)"; }

const char* nz_last_error(const nz_engine* e) { return e ? e->error.c_str() : g_create_error.c_str(); }

nz_status nz_engine_create(nz_engine** out, const nz_search_cfg* cfg, const nz_game_desc* game, int32_t n_games,
                           int32_t device) {
  return nz_engine_create_ex(out, cfg, game, n_games, n_games, device);
}

nz_status nz_engine_create_ex(nz_engine** out, const nz_search_cfg* cfg, const nz_game_desc* game, int32_t n_slots,
                              int32_t n_games, int32_t device) {
  if (!out || !cfg || !game) return fail(nullptr, NZ_ERR_ARG, "null argument");
  if (n_slots <= 0) return fail(nullptr, NZ_ERR_ARG, "n_slots must be positive");
  if (n_slots > n_games) n_slots = n_games > 0 ? n_games : n_slots;
  *out = nullptr;
  if (game->game != NZ_GAME_TIC_TAC_TOE) return fail(nullptr, NZ_ERR_ARG, "unsupported game %d", game->game);
  if (n_games <= 0) return fail(nullptr, NZ_ERR_ARG, "n_games must be positive");
  if (cfg->mcts_simulations <= 0) return fail(nullptr, NZ_ERR_ARG, "mcts_simulations must be positive");
  if (!cfg->keep_subtree)
    return fail(nullptr, NZ_ERR_ARG, "keep_subtree = False is not supported: the reference never resets the root "
                                     "in that mode (Training/Gamer.py:78-79) and every shipped config sets True");
  if (!(cfg->pb_c_base > 0)) return fail(nullptr, NZ_ERR_ARG, "pb_c_base must be positive");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, NZ_ERR_HIP, "no HIP device available (the engine has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(nullptr, NZ_ERR_ARG, "device %d out of range", device);

  nz_engine* e = new nz_engine;
  e->cfg = *cfg;
  e->game = *game;
  e->device = device;
  e->n_games = n_games;
  e->n_slots = n_slots;
  // every expansion at move m adds at most 9 - m children: 1 + sims * (9 + 8 + ... + 1)
  e->cap = 1 + cfg->mcts_simulations * 45;
  e->tab_len = cfg->mcts_simulations * TTT_MAX_MOVES + 2;
  nz_status st = NZ_OK;
  auto bail = [&](nz_status s) {
    g_create_error = e->error;
    nz_engine_destroy(e);
    return s;
  };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(e, NZ_ERR_HIP, "hipSetDevice(%d) failed", device));

  TreeParams& p = e->tp;
  memset(&p, 0, sizeof(p));
  const size_t G = n_games, N = (size_t)n_slots * (size_t)e->cap, GT = G * TTT_MAX_MOVES, GTA = GT * TTT_ACTIONS;
#define A(ptr, n)                                        \
  if ((st = dev_alloc(e, &(ptr), (n))) != NZ_OK) return bail(st)
  A(p.nodes, N);
  A(p.board, G); A(p.length, G); A(p.alive, G); A(p.outcome, G); A(p.root, G); A(p.node_count, G);
  A(p.sims_left, G); A(p.pending, G); A(p.leaf_board, G); A(p.path, G * MAX_PATH); A(p.path_len, G);
  A(p.sim_count, G); A(p.exp_count, G); A(p.sel_nodes, G); A(p.sel_children, G); A(p.new_nodes, G); A(p.desync, G); A(p.n_root_children, G);
  A(p.leaf_count, 2); A(p.leaf_boards, G); A(e->leaf_logits, G * TTT_ACTIONS); A(e->leaf_value, G);
  A(p.error_flag, 1);
  A(p.hist_board, GT); A(p.hist_action, GT); A(p.hist_visits, GTA); A(p.hist_tree_size, GT);
  A(p.hist_children, GT); A(p.hist_bias, GT); A(p.hist_prior, GTA); A(p.hist_value_sum, GTA);
  A(p.hist_root_value_sum, GT);
  double *bias_tab = nullptr, *sqrt_tab = nullptr;
  A(bias_tab, e->tab_len); A(sqrt_tab, e->tab_len);
  A(e->d_noise, G * TTT_ACTIONS); A(e->d_uniforms, G * 3);
  A(e->d_game_noise, GTA); A(e->d_game_uniforms, GT * 3);
  A(e->d_stamps, (size_t)n_slots * 6);                 // (one workgroup per slot at most)
  A(p.next_game, 1);
  A(e->prog_dev, 1);
#undef A
  p.leaf_logits = e->leaf_logits;
  p.leaf_value = e->leaf_value;
  p.cap = e->cap;
  p.n_games = n_games;
  p.n_slots = n_slots;
  p.table = nullptr;
  p.tab_len = e->tab_len;
  p.sims = cfg->mcts_simulations;
  p.negate_player = game->negate_player;
  p.value_factor = cfg->value_factor;
  p.frac = cfg->root_exploration_fraction;
  p.one_minus_frac = 1.0 - cfg->root_exploration_fraction;
  p.training = cfg->training;
  p.softmax_moves = cfg->number_of_softmax_moves;
  p.eps_softmax = cfg->epsilon_softmax_exploration;
  p.eps_random = cfg->epsilon_random_exploration;
  {   // a workgroup's 16 network rows are filled only when there are 16 slots for every CU: fewer slots are spread over the
      // chip (1024 slots: 4 to each of 256 workgroups instead of 16 to each of 64 -- a pass costs the same matrix
      // instructions however many of its columns hold a leaf, and a tree phase waits for the slowest of fewer rows)
    hipDeviceProp_t prop;
    int n_cu = 256;
    if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    p.slots_per_wg = std::min(16, std::max(1, (n_slots + n_cu - 1) / n_cu));
    if (const char* v = getenv("NZ_SLOTS_PER_WG")) p.slots_per_wg = std::min(16, std::max(1, atoi(v)));   // tuning experiments
  }
  // a row runs at most this many simulations between two network passes (results do not depend on it).  With four slots to
  // a workgroup more helps at 100 simulations a move (1024 slots, one box: 6 / 10 / 16 / 32 -> 28.8 / 30.1 / 30.8 / 31.4 k
  // games/s) and hurts at configs[2]'s 400 (12 / 16 / 24 / 32 -> 8.76 / 8.68 / 8.52 / 8.52 k): 16 everywhere
  p.sims_per_cycle = 16;
  if (const char* v = getenv("NZ_SIMS_PER_CYCLE")) p.sims_per_cycle = std::max(1, atoi(v));   // tuning experiments
  {   // every cycle finishes at least one simulation or one move of every live row of the workgroup
    const double games_per_slot = std::ceil((double)p.n_games / (double)p.n_slots) + 2.0;
    const double bound = 16.0 * games_per_slot * TTT_MAX_MOVES * ((double)cfg->mcts_simulations + 2.0);
    p.max_cycles = (int32_t)std::min(bound, 2.0e9);
  }

  // Explorer.calculate_exploration_bias / calculate_ucb_factor (Explorer.py:103-112):
  // log() and sqrt() of the parent visit count, from the host libm
  std::vector<double> hb(e->tab_len), hs(e->tab_len);
  for (int n = 0; n < e->tab_len; ++n) {
    hb[n] = std::log(((double)n + cfg->pb_c_base + 1.0) / cfg->pb_c_base) + cfg->pb_c_init;
    hs[n] = std::sqrt((double)n);
  }
  if (hipMemcpy(bias_tab, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(sqrt_tab, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(e, NZ_ERR_HIP, "table upload failed"));
  p.bias_tab = bias_tab;
  p.sqrt_tab = sqrt_tab;

  if (hipHostMalloc((void**)&e->h_children, G * sizeof(int32_t)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_alive, G * sizeof(int32_t)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_noise, G * TTT_ACTIONS * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_uniforms, G * 3 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_game_noise, GTA * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_game_uniforms, GT * 3 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_next_noise, GTA * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_next_uniforms, GT * 3 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_desync, G * sizeof(int32_t)) != hipSuccess)
    return bail(fail(e, NZ_ERR_HIP, "pinned host allocation failed"));
  memset(e->h_noise, 0, G * TTT_ACTIONS * sizeof(double));
  memset(e->h_uniforms, 0, G * 3 * sizeof(double));
  memset(e->h_game_noise, 0, GTA * sizeof(double));
  memset(e->h_game_uniforms, 0, GT * 3 * sizeof(double));
  memset(e->h_next_noise, 0, GTA * sizeof(double));
  memset(e->h_next_uniforms, 0, GT * 3 * sizeof(double));

  launch_reset(p, nullptr);
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(e, NZ_ERR_HIP, "reset kernel failed"));
  *out = e;
  return NZ_OK;
}

void nz_engine_destroy(nz_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->fallback) nz_engine_destroy(e->fallback);
  if (e->weights_dev && !e->borrowed_net) (void)hipFree(e->weights_dev);
  if (e->table_dev && !e->borrowed_net) (void)hipFree(e->table_dev);
  if (e->h_game_noise) (void)hipHostFree(e->h_game_noise);
  if (e->h_game_uniforms) (void)hipHostFree(e->h_game_uniforms);
  if (e->h_desync) (void)hipHostFree(e->h_desync);
  if (e->h_next_noise) (void)hipHostFree(e->h_next_noise);
  if (e->h_next_uniforms) (void)hipHostFree(e->h_next_uniforms);
  if (e->h_children) (void)hipHostFree(e->h_children);
  if (e->h_alive) (void)hipHostFree(e->h_alive);
  if (e->h_noise) (void)hipHostFree(e->h_noise);
  if (e->h_uniforms) (void)hipHostFree(e->h_uniforms);
  for (nz_rng* r : e->rngs) nz_rng_destroy(r);
  for (auto& sp : e->spans) {
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  delete e;
}

nz_status nz_engine_dims(const nz_engine* e, nz_dims* out) {
  if (!e || !out) return NZ_ERR_ARG;
  out->n_games = e->n_games;
  out->n_slots = e->n_slots;
  out->num_actions = TTT_ACTIONS;
  out->max_moves = TTT_MAX_MOVES;
  out->state_channels = 2;
  out->rows = 3;
  out->cols = 3;
  out->node_capacity = e->cap;
  return NZ_OK;
}

nz_status nz_engine_set_weights(nz_engine* e, const nz_net_desc* net, const float* const* weights,
                                int32_t n_tensors, int32_t recurrent_iterations) {
  if (!e || !net || !weights) return NZ_ERR_ARG;
  if (net->hex) return fail(e, NZ_ERR_ARG, "hexagonal convolutions are built for nz_boardnet_* only");
  if (net->in_channels != 2 || net->policy_channels != 1)
    return fail(e, NZ_ERR_ARG, "Tic-Tac-Toe nets take 2 input planes and 1 policy plane");
  if (net->width <= 0 || net->width > 64 || net->width % 4 != 0)
    return fail(e, NZ_ERR_ARG, "width must be a multiple of 4 in (0, 64]");
  if (net->num_blocks < 0 || recurrent_iterations < 0) return fail(e, NZ_ERR_ARG, "negative block/iteration count");
  const int arch = net->arch;
  if (arch != NZ_ARCH_RECURRENT && arch != NZ_ARCH_RESNET && arch != NZ_ARCH_CONVNET)
    return fail(e, NZ_ERR_ARG, "unknown architecture %d", arch);
  const int trunk_k = arch == NZ_ARCH_CONVNET ? net->kernel_size : 3;
  if (trunk_k != 1 && trunk_k != 3) return fail(e, NZ_ERR_ARG, "ConvNet kernel_size must be 1 or 3");
  const bool recall = arch == NZ_ARCH_RECURRENT && net->recall;
  const int iterations = arch == NZ_ARCH_RECURRENT ? recurrent_iterations : 1;
  const int trunk_tensors = arch == NZ_ARCH_CONVNET ? 1 + net->num_blocks : 1 + (recall ? 1 : 0) + 2 * net->num_blocks;
  const int expect = trunk_tensors + 2 + 4;
  if (n_tensors != expect) return fail(e, NZ_ERR_ARG, "expected %d weight tensors, got %d", expect, n_tensors);
  if (1 + iterations * (trunk_tensors - 1) + 24 > NET_MAX_JOBS)
    return fail(e, NZ_ERR_ARG, "too many layers for one fused launch (%d iterations)", recurrent_iterations);
  NZ_HIP(e, hipSetDevice(e->device));

  const int W = net->width, IN = net->in_channels;
  // tensor shapes in state_dict order
  struct Shape { int cout, cin, k; };
  std::vector<Shape> shapes;
  shapes.push_back({W, IN, trunk_k});
  if (recall) shapes.push_back({W, W + IN, 3});
  for (int i = (int)shapes.size(); i < trunk_tensors; ++i) shapes.push_back({W, W, trunk_k});
  for (int i = 0; i < 2; ++i)
    shapes.push_back({head_channel(W, net->policy_channels, 2, i + 1), head_channel(W, net->policy_channels, 2, i), 3});
  for (int i = 0; i < 4; ++i) shapes.push_back({head_channel(W, 1, 4, i + 1), head_channel(W, 1, 4, i), 3});

  std::vector<std::vector<float>> host(n_tensors);
  for (int i = 0; i < n_tensors; ++i) {
    const size_t n_in = (size_t)shapes[i].cout * shapes[i].cin * shapes[i].k * shapes[i].k;
    std::vector<float> raw(n_in);
    NZ_HIP(e, hipMemcpy(raw.data(), weights[i], n_in * sizeof(float), hipMemcpyDefault));
    if (shapes[i].k == 3) {
      host[i].swap(raw);
    } else {                                   // 1x1 conv = a 3x3 conv whose only tap is the centre
      host[i].assign((size_t)shapes[i].cout * shapes[i].cin * 9, 0.f);
      for (size_t j = 0; j < n_in; ++j) host[i][j * 9 + 4] = raw[j];
    }
  }

  std::vector<float> packed;
  std::vector<PackedConv> convs(n_tensors);
  double flops = 0.0;   // in-bounds taps only: 2 * Cout * Cin * 49 per 3x3 conv application (9 for 1x1)
  for (int i = 0; i < n_tensors; ++i) {
    const bool with_planes = (i == 0) || (recall && i == 1);
    const int cin_main = with_planes ? shapes[i].cin - IN : shapes[i].cin;
    convs[i] = pack_conv(host[i].data(), shapes[i].cout, shapes[i].cin, cin_main, packed);
  }

  // stages; activation buffers 0/1 ping-pong, `cur` holds the running trunk output.
  // act: 1 relu, 2 tanh, 3 elu
  NetProgram& pg = e->prog_host;
  memset(&pg, 0, sizeof(pg));
  bool ok = true;
  int cur = 0;
  auto stage = [&](std::vector<StageConv> convs_in_stage) {
    for (const StageConv& sc : convs_in_stage)
      flops += 2.0 * shapes[sc.tensor].cout * shapes[sc.tensor].cin * (shapes[sc.tensor].k == 3 ? 49.0 : 9.0);
    ok = ok && add_stage(pg, convs, convs_in_stage);
  };
  if (arch == NZ_ARCH_CONVNET) {                                      // ConvNet.py:20-40: (conv, ELU) x (1 + num_layers)
    stage({{0, 0, cur, -1, 3, true}});
    for (int i = 1; i < trunk_tensors; ++i) { stage({{i, cur, cur ^ 1, -1, 3, true}}); cur ^= 1; }
  } else {
    stage({{0, 0, cur, -1, 1, true}});                                // projection / input block + ReLU
    const int first_block = recall ? 2 : 1;
    for (int it = 0; it < iterations; ++it) {
      if (recall) { stage({{1, cur, cur ^ 1, -1, 0, true}}); cur ^= 1; }    // cat([thought, x]) conv, no activation
      for (int b = 0; b < net->num_blocks; ++b) {                         // relu(conv2(relu(conv1(t))) + t)
        stage({{first_block + 2 * b, cur, cur ^ 1, -1, 1, true}});
        stage({{first_block + 2 * b + 1, cur ^ 1, cur, cur, 1, true}});
      }
    }
  }
  const int ph = trunk_tensors, vh = ph + 2;
  const int vact = net->value_activation == NZ_ACT_RELU ? 1 : 2;
  const int side = cur ^ 1;
  // two buffers: the policy head runs first (trunk -> side -> logits), then the value head
  // ping-pongs between the two (the trunk output is dead after its first layer)
  stage({{ph, cur, side, -1, 1, true}});
  stage({{ph + 1, side, NET_DST_POLICY, -1, 0, true}});
  stage({{vh, cur, side, -1, vact, true}});
  stage({{vh + 1, side, cur, -1, vact, true}});
  stage({{vh + 2, cur, side, -1, vact, true}});
  stage({{vh + 3, side, NET_DST_VALUE, -1, 0, true}});                      // per-cell outputs; net_tile takes the mean
  if (!ok) return fail(e, NZ_ERR_ARG, "network too deep for one fused launch (%d iterations)", recurrent_iterations);
  for (int w = 0; w < NET_WAVES_HOST; ++w) {     // prefetch chain: each job names the next weight stream
    int32_t next = -1;
    for (int j = pg.n_jobs[w] - 1; j >= 0; --j) {
      pg.jobs[w][j].next_w_off = next;
      if (pg.jobs[w][j].og != OG_NONE && pg.jobs[w][j].kgroups > 0) next = pg.jobs[w][j].w_off;
    }
    pg.first_w_off[w] = next;
  }
  e->algorithmic_flops_per_position = flops;
  {   // what the matrix cores execute for one tile of 16 positions (net_dev.hpp): six bf16 MFMAs per
      // (input cell, tap) pair and 32-channel K group, one float32 MFMA per pair for the input planes
    double bf16 = 0.0, f32 = 0.0;
    for (int w = 0; w < NET_WAVES_HOST; ++w)
      for (int j = 0; j < pg.n_jobs[w]; ++j) {
        const NetJob& job = pg.jobs[w][j];
        if (job.og == OG_NONE) continue;
        bf16 += (double)og_taps[job.og] * job.kgroups * 6.0 * (2.0 * 16 * 16 * 32);
        f32 += (double)og_taps[job.og] * job.extra * (2.0 * 16 * 16 * 4);
      }
    e->executed_bf16_flops_per_position = bf16 / 16.0;
    e->executed_f32_flops_per_position = f32 / 16.0;
  }

  if (e->weights_dev) { (void)hipFree(e->weights_dev); e->weights_dev = nullptr; }
  NZ_HIP(e, hipMalloc((void**)&e->weights_dev, packed.size() * sizeof(float)));
  NZ_HIP(e, hipMemcpy(e->weights_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  NZ_HIP(e, hipMemcpy(e->prog_dev, &pg, sizeof(pg), hipMemcpyHostToDevice));
  e->net = *net;
  e->iters = recurrent_iterations;
  e->have_net = true;
  e->have_table = false;
  e->tp.table = nullptr;
  return NZ_OK;
}

nz_status nz_engine_set_table(nz_engine* e, const float* table, int32_t n_rows) {
  if (!e || !table) return NZ_ERR_ARG;
  if (n_rows != TTT_TABLE_ROWS) return fail(e, NZ_ERR_ARG, "table must have %d rows", TTT_TABLE_ROWS);
  NZ_HIP(e, hipSetDevice(e->device));
  if (!e->table_dev) NZ_HIP(e, hipMalloc((void**)&e->table_dev, (size_t)TTT_TABLE_ROWS * 10 * sizeof(float)));
  NZ_HIP(e, hipMemcpy(e->table_dev, table, (size_t)TTT_TABLE_ROWS * 10 * sizeof(float), hipMemcpyDefault));
  e->tp.table = e->table_dev;
  e->have_table = true;
  return NZ_OK;
}

nz_status nz_engine_reset(nz_engine* e, void* stream) {
  if (!e) return NZ_ERR_ARG;
  NZ_HIP(e, hipSetDevice(e->device));
  Span sp(e, as_stream(stream), 2);
  launch_reset(e->tp, as_stream(stream));
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_root_children(nz_engine* e, int32_t* n_children_dev, void* stream) {
  if (!e || !n_children_dev) return NZ_ERR_ARG;
  NZ_HIP(e, hipMemcpyAsync(n_children_dev, e->tp.n_root_children, e->n_games * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, as_stream(stream)));
  return NZ_OK;
}

nz_status nz_engine_alive(nz_engine* e, int32_t* alive_dev, void* stream) {
  if (!e || !alive_dev) return NZ_ERR_ARG;
  NZ_HIP(e, hipMemcpyAsync(alive_dev, e->tp.alive, e->n_games * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           as_stream(stream)));
  return NZ_OK;
}

// noise + simulations of one move for every live game (no action yet)
static nz_status search_lockstep(nz_engine* e, const double* noise_dev, hipStream_t s) {
  const TreeParams& p = e->tp;
  if (e->cfg.training) {
    Span sp(e, s, 2);
    launch_noise(p, noise_dev, s);
  }
  if (p.table != nullptr) {
    // table evaluator: nothing leaves the tree kernel, one launch does the whole search
    Span sp(e, s, 0);
    launch_advance(p, 0, s);
  } else {
    const int sims = e->cfg.mcts_simulations;
    for (int it = 0; it <= sims; ++it) {
      {
        Span sp(e, s, 0);
        launch_advance(p, it, s);
      }
      if (it == sims) break;
      {
        Span sp(e, s, 1);
        launch_net(e->prog_dev, 0, e->weights_dev, p.leaf_boards, nullptr,
                   p.leaf_count + (it & 1), e->n_games, e->leaf_logits, e->leaf_value, nullptr, nullptr, s);
      }
    }
  }
  return NZ_OK;
}

static nz_status check_lockstep_call(nz_engine* e, const double* noise_dev, const double* uniforms_dev, bool need_uni) {
  if (e->n_slots != e->n_games)
    return fail(e, NZ_ERR_STATE, "the lock-step route needs n_slots == n_games (every game of the round in flight)");
  if (!e->have_net && !e->have_table) return fail(e, NZ_ERR_STATE, "no network: call nz_engine_set_weights first");
  if (e->cfg.training && (!noise_dev || (need_uni && !uniforms_dev)))
    return fail(e, NZ_ERR_ARG, "training search needs noise and uniforms");
  return NZ_OK;
}

nz_status nz_engine_move(nz_engine* e, const double* noise_dev, const double* uniforms_dev, void* stream) {
  if (!e) return NZ_ERR_ARG;
  nz_status st = check_lockstep_call(e, noise_dev, uniforms_dev, true);
  if (st != NZ_OK) return st;
  NZ_HIP(e, hipSetDevice(e->device));
  hipStream_t s = as_stream(stream);
  search_lockstep(e, noise_dev, s);
  {
    Span sp(e, s, 2);
    launch_finish_move(e->tp, uniforms_dev, nullptr, s);
  }
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_search(nz_engine* e, const double* noise_dev, void* stream) {
  if (!e) return NZ_ERR_ARG;
  nz_status st = check_lockstep_call(e, noise_dev, nullptr, false);
  if (st != NZ_OK) return st;
  NZ_HIP(e, hipSetDevice(e->device));
  search_lockstep(e, noise_dev, as_stream(stream));
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_apply(nz_engine* e, const int32_t* actions_dev, const double* uniforms_dev, void* stream) {
  if (!e) return NZ_ERR_ARG;
  if (e->n_slots != e->n_games)
    return fail(e, NZ_ERR_STATE, "the lock-step route needs n_slots == n_games (every game of the round in flight)");
  if (!actions_dev && e->cfg.training && !uniforms_dev)
    return fail(e, NZ_ERR_ARG, "a training engine choosing its own action needs uniforms");
  NZ_HIP(e, hipSetDevice(e->device));
  hipStream_t s = as_stream(stream);
  {
    Span sp(e, s, 2);
    launch_finish_move(e->tp, uniforms_dev, actions_dev, s);
  }
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_last_actions(nz_engine* e, int32_t* actions_dev, void* stream) {
  if (!e || !actions_dev) return NZ_ERR_ARG;
  NZ_HIP(e, hipSetDevice(e->device));
  launch_last_actions(e->tp, actions_dev, as_stream(stream));
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_live_games(nz_engine* e, int32_t* n_live_host, void* stream) {
  if (!e || !n_live_host) return NZ_ERR_ARG;
  hipStream_t s = as_stream(stream);
  NZ_HIP(e, hipMemcpyAsync(e->h_alive, e->tp.alive, e->n_games * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  NZ_HIP(e, hipStreamSynchronize(s));
  int n = 0;
  for (int g = 0; g < e->n_games; ++g) n += e->h_alive[g] != 0;
  *n_live_host = n;
  return check_device_flag(e, s);
}

// The per-move draws of one game in the reference's order (SURVEY.md appendix A
// rule 13): gamma x n_children, then random() x 2 unless this is a softmax
// move, then at most one more random() inside np.random.choice.
static void draw_move(nz_rng* r, const nz_search_cfg& c, int move, int n_children, double* noise9, double* uni3) {
  nz_rng_gamma(r, c.root_dist_alpha, c.root_dist_beta, n_children, noise9);
  double u1 = 0.0, u2 = 0.0, u3 = 0.0;
  bool choice;
  if (move < c.number_of_softmax_moves) {
    choice = true;
  } else {
    u1 = nz_rng_double(r);
    u2 = nz_rng_double(r);
    choice = (u1 < c.epsilon_softmax_exploration) || (u2 < c.epsilon_random_exploration);
  }
  if (choice) u3 = nz_rng_double(r);
  uni3[0] = u1;
  uni3[1] = u2;
  uni3[2] = u3;
}

static void ensure_rngs(nz_engine* e) {
  if ((int)e->rngs.size() == e->n_games) return;
  for (nz_rng* r : e->rngs) nz_rng_destroy(r);
  e->rngs.assign(e->n_games, nullptr);
  for (int g = 0; g < e->n_games; ++g) e->rngs[g] = nz_rng_create(0);
}

// Lock-step play: one host round trip per move to learn each root's child count
// before drawing.  `seeds[g]` seeds game g's stream.
static nz_status play_lockstep(nz_engine* e, const uint32_t* seeds, void* stream) {
  if (e->n_slots != e->n_games)
    return fail(e, NZ_ERR_STATE, "the lock-step route needs n_slots == n_games (every game of the round in flight)");
  hipStream_t s = as_stream(stream);
  const int G = e->n_games;
  const nz_search_cfg& c = e->cfg;
  if (c.training) {
    ensure_rngs(e);
    for (int g = 0; g < G; ++g) nz_rng_seed(e->rngs[g], seeds[g]);
  }
  nz_status st = nz_engine_reset(e, stream);
  if (st != NZ_OK) return st;
  for (int move = 0; move < TTT_MAX_MOVES; ++move) {
    NZ_HIP(e, hipMemcpyAsync(e->h_children, e->tp.n_root_children, G * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NZ_HIP(e, hipMemcpyAsync(e->h_alive, e->tp.alive, G * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NZ_HIP(e, hipStreamSynchronize(s));
    int live = 0;
    for (int g = 0; g < G; ++g) live += e->h_alive[g] != 0;
    if (live == 0) break;
    if (c.training) {
      for (int g = 0; g < G; ++g)
        if (e->h_alive[g])
          draw_move(e->rngs[g], c, move, e->h_children[g], e->h_noise + (size_t)g * TTT_ACTIONS, e->h_uniforms + g * 3);
      NZ_HIP(e, hipMemcpyAsync(e->d_noise, e->h_noise, (size_t)G * TTT_ACTIONS * sizeof(double), hipMemcpyHostToDevice, s));
      NZ_HIP(e, hipMemcpyAsync(e->d_uniforms, e->h_uniforms, (size_t)G * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    st = nz_engine_move(e, e->d_noise, e->d_uniforms, stream);
    if (st != NZ_OK) return st;
  }
  return check_device_flag(e, s);
}

nz_status nz_engine_play_lockstep(nz_engine* e, uint64_t base_seed, void* stream) {
  if (!e) return NZ_ERR_ARG;
  if (!e->have_net && !e->have_table) return fail(e, NZ_ERR_STATE, "no network: call nz_engine_set_weights first");
  NZ_HIP(e, hipSetDevice(e->device));
  std::vector<uint32_t> seeds(e->n_games);
  for (int g = 0; g < e->n_games; ++g) seeds[g] = (uint32_t)((base_seed + (uint64_t)g) & 0xffffffffu);
  return play_lockstep(e, seeds.data(), stream);
}

// Replay the games the persistent kernel flagged (their pre-drawn randomness did
// not fit) on a small lock-step engine and copy their records back.
static nz_status replay_desynced(nz_engine* e, const std::vector<int>& games, uint64_t base_seed, void* stream) {
  hipStream_t s = as_stream(stream);
  const int n = (int)games.size();
  if (e->fallback && e->fallback->n_games < n) {
    nz_engine_destroy(e->fallback);
    e->fallback = nullptr;
  }
  if (!e->fallback) {
    nz_engine* f = nullptr;
    nz_status st = nz_engine_create(&f, &e->cfg, &e->game, std::max(n, 16), e->device);
    if (st != NZ_OK) return fail(e, st, "fallback engine: %s", nz_last_error(nullptr));
    f->borrowed_net = true;
    e->fallback = f;
  }
  nz_engine* f = e->fallback;
  f->weights_dev = e->weights_dev;
  f->table_dev = e->table_dev;
  f->tp.table = e->tp.table;
  f->have_net = e->have_net;
  f->have_table = e->have_table;
  f->prog_host = e->prog_host;
  NZ_HIP(e, hipMemcpy(f->prog_dev, &e->prog_host, sizeof(NetProgram), hipMemcpyHostToDevice));
  std::vector<uint32_t> seeds(f->n_games, 0u);
  for (int i = 0; i < n; ++i) seeds[i] = (uint32_t)((base_seed + (uint64_t)games[i]) & 0xffffffffu);
  nz_status st = play_lockstep(f, seeds.data(), stream);
  if (st != NZ_OK) return fail(e, st, "fallback replay: %s", f->error.c_str());
  const TreeParams &a = f->tp, &b = e->tp;
  for (int i = 0; i < n; ++i) {
    const size_t g = games[i];
#define ROW(field, per_game)                                                                               \
  NZ_HIP(e, hipMemcpyAsync(b.field + g * (per_game), a.field + (size_t)i * (per_game),                     \
                           (per_game) * sizeof(*a.field), hipMemcpyDeviceToDevice, s))
    ROW(hist_board, TTT_MAX_MOVES); ROW(hist_action, TTT_MAX_MOVES); ROW(hist_tree_size, TTT_MAX_MOVES);
    ROW(hist_children, TTT_MAX_MOVES); ROW(hist_bias, TTT_MAX_MOVES); ROW(hist_root_value_sum, TTT_MAX_MOVES);
    ROW(hist_visits, TTT_MAX_MOVES * TTT_ACTIONS); ROW(hist_prior, TTT_MAX_MOVES * TTT_ACTIONS);
    ROW(hist_value_sum, TTT_MAX_MOVES * TTT_ACTIONS);
    ROW(length, 1); ROW(outcome, 1); ROW(board, 1);
    ROW(sim_count, 1); ROW(exp_count, 1); ROW(sel_nodes, 1); ROW(sel_children, 1); ROW(new_nodes, 1);
#undef ROW
  }
  NZ_HIP(e, hipStreamSynchronize(s));
  return NZ_OK;
}

// every game's draws for a whole round into (noise [G][T][A], uniforms [G][T][3]), assuming the root of move m >= 1
// has 9 - m children (selfplay.hip); games are dealt to host threads
static void draw_round(nz_engine* e, uint64_t base_seed, double* noise, double* uniforms) {
  const int G = e->n_games;
  const nz_search_cfg& c = e->cfg;
  const int n_threads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  auto work = [&](int t) {
    for (int g = t; g < G; g += n_threads) {
      nz_rng* r = e->rngs[g];
      nz_rng_seed(r, (uint32_t)((base_seed + (uint64_t)g) & 0xffffffffu));
      for (int m = 0; m < TTT_MAX_MOVES; ++m)
        draw_move(r, c, m, m == 0 ? 0 : TTT_ACTIONS - m, noise + ((size_t)g * TTT_MAX_MOVES + m) * TTT_ACTIONS,
                  uniforms + ((size_t)g * TTT_MAX_MOVES + m) * 3);
    }
  };
  if (G < 256 || n_threads == 1) {
    for (int t = 0; t < n_threads; ++t) work(t);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
}

nz_status nz_engine_play(nz_engine* e, uint64_t base_seed, void* stream) {
  return nz_engine_play_next(e, base_seed, 0, 0, stream);
}

nz_status nz_engine_play_next(nz_engine* e, uint64_t base_seed, int32_t have_next, uint64_t next_base_seed, void* stream) {
  if (!e) return NZ_ERR_ARG;
  if (!e->have_net && !e->have_table) return fail(e, NZ_ERR_STATE, "no network: call nz_engine_set_weights first");
  NZ_HIP(e, hipSetDevice(e->device));
  hipStream_t s = as_stream(stream);
  const int G = e->n_games;
  const nz_search_cfg& c = e->cfg;

  nz_status st = nz_engine_reset(e, stream);
  if (st != NZ_OK) return st;
  if (c.training) {
    ensure_rngs(e);
    const size_t GT = (size_t)G * TTT_MAX_MOVES;
    if (e->next_ready && e->next_seed == base_seed) {      // drawn under the previous round's kernel
      std::swap(e->h_game_noise, e->h_next_noise);
      std::swap(e->h_game_uniforms, e->h_next_uniforms);
    } else {
      draw_round(e, base_seed, e->h_game_noise, e->h_game_uniforms);
    }
    e->next_ready = false;
    NZ_HIP(e, hipMemcpyAsync(e->d_game_noise, e->h_game_noise, GT * TTT_ACTIONS * sizeof(double), hipMemcpyHostToDevice, s));
    NZ_HIP(e, hipMemcpyAsync(e->d_game_uniforms, e->h_game_uniforms, GT * 3 * sizeof(double), hipMemcpyHostToDevice, s));
  }
  {
    Span sp(e, s, 0);
    launch_selfplay(e->tp, e->prog_dev, 0, e->weights_dev,
                    c.training ? e->d_game_noise : nullptr, c.training ? e->d_game_uniforms : nullptr,
                    e->stamps ? e->d_stamps : nullptr, s);
  }
  NZ_HIP(e, hipGetLastError());
  if (c.training && have_next) {                            // the next round's draws, while the kernel runs
    draw_round(e, next_base_seed, e->h_next_noise, e->h_next_uniforms);
    e->next_ready = true;
    e->next_seed = next_base_seed;
  }
  st = check_device_flag(e, s);
  if (st != NZ_OK) return st;
  if (c.training) {
    NZ_HIP(e, hipMemcpyAsync(e->h_desync, e->tp.desync, G * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NZ_HIP(e, hipStreamSynchronize(s));
    std::vector<int> bad;
    for (int g = 0; g < G; ++g)
      if (e->h_desync[g]) bad.push_back(g);
    if (!bad.empty()) {
      e->desync_total += (int64_t)bad.size();
      st = replay_desynced(e, bad, base_seed, stream);
      if (st != NZ_OK) return st;
    }
  }
  return NZ_OK;
}

nz_status nz_engine_phase_stamps(nz_engine* e, int32_t enable, double* out4_host) {
  if (!e) return NZ_ERR_ARG;
  if (out4_host) {
    const int blocks = selfplay_blocks(e->n_slots, e->tp.slots_per_wg);
    std::vector<unsigned long long> h((size_t)blocks * 6);
    NZ_HIP(e, hipSetDevice(e->device));
    NZ_HIP(e, hipDeviceSynchronize());
    NZ_HIP(e, hipMemcpy(h.data(), e->d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[4] = {0, 0, 0, 0};
    double max_total = 0;
    for (int b = 0; b < blocks; ++b) {
      for (int i = 0; i < 4; ++i) sum[i] += (double)h[b * 4 + i];
      max_total = std::max(max_total, (double)h[b * 4 + 3]);
    }
    out4_host[0] = sum[0] / blocks;              // mean tree/net cycles per workgroup
    out4_host[1] = sum[1] / std::max(sum[3], 1.0);   // share of ticks in the tree phase
    out4_host[2] = sum[2] / std::max(sum[3], 1.0);   // share of ticks in the net phase
    out4_host[3] = sum[3] / blocks / std::max(max_total, 1.0);   // mean / max workgroup lifetime
    out4_host[4] = sum[2] / std::max(sum[0], 1.0);   // shader-clock ticks per network phase
    out4_host[5] = sum[1] / std::max(sum[0], 1.0);   // shader-clock ticks per tree phase
    out4_host[6] = max_total;                        // ticks of the longest-lived workgroup
    out4_host[7] = (double)blocks;
    double fin = 0, exp = 0;
    for (int b = 0; b < blocks; ++b) {
      fin += (double)h[(size_t)blocks * 4 + b * 2];
      exp += (double)h[(size_t)blocks * 4 + b * 2 + 1];
    }
    out4_host[8] = fin / std::max(sum[0], 1.0);      // wave 0: ticks per cycle in end-of-move bookkeeping
    out4_host[9] = exp / std::max(sum[0], 1.0);      // wave 0: ticks per cycle in the pending expansion
  }
  e->stamps = enable != 0;
  return NZ_OK;
}

nz_status nz_engine_desync_count(const nz_engine* e, int64_t* count_host) {
  if (!e || !count_host) return NZ_ERR_ARG;
  *count_host = e->desync_total;
  return NZ_OK;
}

nz_status nz_engine_export(nz_engine* e, float* states, int32_t* visits, int32_t* actions, int32_t* lengths,
                           int32_t* outcomes, int32_t* tree_size, int32_t* n_children, double* bias, void* stream) {
  if (!e) return NZ_ERR_ARG;
  NZ_HIP(e, hipSetDevice(e->device));
  hipStream_t s = as_stream(stream);
  Span sp(e, s, 2);
  if (states) launch_export_states(e->tp, states, s);
  if (visits || actions || tree_size || n_children || bias)
    launch_export_visits(e->tp, visits, actions, tree_size, n_children, bias, s);
  if (lengths)
    NZ_HIP(e, hipMemcpyAsync(lengths, e->tp.length, e->n_games * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (outcomes)
    NZ_HIP(e, hipMemcpyAsync(outcomes, e->tp.outcome, e->n_games * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_engine_export_trace(nz_engine* e, double* child_prior, double* child_value_sum, double* root_value_sum,
                                 void* stream) {
  if (!e) return NZ_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const size_t GT = (size_t)e->n_games * TTT_MAX_MOVES;
  if (child_prior)
    NZ_HIP(e, hipMemcpyAsync(child_prior, e->tp.hist_prior, GT * TTT_ACTIONS * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (child_value_sum)
    NZ_HIP(e, hipMemcpyAsync(child_value_sum, e->tp.hist_value_sum, GT * TTT_ACTIONS * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (root_value_sum)
    NZ_HIP(e, hipMemcpyAsync(root_value_sum, e->tp.hist_root_value_sum, GT * sizeof(double), hipMemcpyDeviceToDevice, s));
  return NZ_OK;
}

nz_status nz_engine_counters(nz_engine* e, int64_t* simulations_host, int64_t* expansions_host, void* stream) {
  if (!e) return NZ_ERR_ARG;
  hipStream_t s = as_stream(stream);
  std::vector<int32_t> a(e->n_games), b(e->n_games);
  NZ_HIP(e, hipMemcpyAsync(a.data(), e->tp.sim_count, a.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  NZ_HIP(e, hipMemcpyAsync(b.data(), e->tp.exp_count, b.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  NZ_HIP(e, hipStreamSynchronize(s));
  int64_t sa = 0, sb = 0;
  for (int g = 0; g < e->n_games; ++g) { sa += a[g]; sb += b[g]; }
  if (simulations_host) *simulations_host = sa;
  if (expansions_host) *expansions_host = sb;
  return NZ_OK;
}

// n_out values of {simulations, expansions, scored nodes, scored children, nodes created}; the caller says how many its
// array holds, so a caller built against an older header (four values) is never written past
nz_status nz_engine_counters_n(nz_engine* e, int64_t* out_host, int32_t n_out, void* stream) {
  if (!e || !out_host || n_out < 1 || n_out > 5) return NZ_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int32_t* src[5] = {e->tp.sim_count, e->tp.exp_count, e->tp.sel_nodes, e->tp.sel_children, e->tp.new_nodes};
  std::vector<int32_t> h(e->n_games);
  for (int i = 0; i < n_out; ++i) {
    NZ_HIP(e, hipMemcpyAsync(h.data(), src[i], h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NZ_HIP(e, hipStreamSynchronize(s));
    int64_t sum = 0;
    for (int32_t v : h) sum += v;
    out_host[i] = sum;
  }
  return NZ_OK;
}

nz_status nz_engine_counters_ex(nz_engine* e, int64_t* out4_host, void* stream) {   // the first four, as first published
  return nz_engine_counters_n(e, out4_host, 4, stream);
}

nz_status nz_engine_net_flops(const nz_engine* e, double* flops_host) {
  if (!e || !flops_host) return NZ_ERR_ARG;
  *flops_host = e->algorithmic_flops_per_position;
  return NZ_OK;
}

nz_status nz_engine_net_matrix_flops(const nz_engine* e, double* bf16_flops_host, double* f32_flops_host) {
  if (!e || !bf16_flops_host || !f32_flops_host) return NZ_ERR_ARG;
  *bf16_flops_host = e->executed_bf16_flops_per_position;
  *f32_flops_host = e->executed_f32_flops_per_position;
  return NZ_OK;
}

nz_status nz_net_forward(nz_engine* e, const float* states_dev, int32_t batch, float* logits_dev, float* value_dev,
                         float* probs_dev, void* stream) {
  if (!e || !states_dev || !logits_dev || !value_dev) return NZ_ERR_ARG;
  if (!e->have_net) return fail(e, NZ_ERR_STATE, "no network: call nz_engine_set_weights first");
  if (batch <= 0) return NZ_OK;
  NZ_HIP(e, hipSetDevice(e->device));
  Span sp(e, as_stream(stream), 1);
  launch_net(e->prog_dev, 0, e->weights_dev, nullptr, states_dev, nullptr, batch, logits_dev,
             value_dev, probs_dev, nullptr, as_stream(stream));
  NZ_HIP(e, hipGetLastError());
  return NZ_OK;
}

nz_status nz_net_forward_stamps(nz_engine* e, const float* states_dev, int32_t batch, float* logits_dev,
                                float* value_dev, double* ticks4_host) {
  if (!e || !states_dev || !logits_dev || !value_dev || !ticks4_host) return NZ_ERR_ARG;
  if (!e->have_net) return fail(e, NZ_ERR_STATE, "no network: call nz_engine_set_weights first");
  NZ_HIP(e, hipSetDevice(e->device));
  const int blocks = (batch + 15) / 16;
  unsigned long long* d = nullptr;
  NZ_HIP(e, hipMalloc((void**)&d, (size_t)blocks * 4 * sizeof(unsigned long long)));
  launch_net(e->prog_dev, 0, e->weights_dev, nullptr, states_dev, nullptr, batch, logits_dev, value_dev, nullptr, d,
             nullptr);
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipError_t err = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(e, NZ_ERR_HIP, "stamp read-back failed: %s", hipGetErrorString(err));
  for (int i = 0; i < 4; ++i) {
    double sum = 0;
    for (int b = 0; b < blocks; ++b) sum += (double)h[(size_t)b * 4 + i];
    ticks4_host[i] = sum / blocks;
  }
  return NZ_OK;
}

nz_status nz_engine_profile(nz_engine* e, int32_t enable) {
  if (!e) return NZ_ERR_ARG;
  for (auto& sp : e->spans) {
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  e->spans.clear();
  e->profile = enable != 0;
  return NZ_OK;
}

nz_status nz_engine_profile_read(nz_engine* e, double* ms_host, int64_t* launches_host, int64_t* net_positions_host) {
  if (!e) return NZ_ERR_ARG;
  NZ_HIP(e, hipSetDevice(e->device));
  NZ_HIP(e, hipDeviceSynchronize());
  double ms[3] = {0, 0, 0};
  int64_t n[3] = {0, 0, 0};
  for (auto& sp : e->spans) {
    float t = 0.f;
    NZ_HIP(e, hipEventElapsedTime(&t, sp.a, sp.b));
    ms[sp.cls] += t;
    n[sp.cls] += 1;
  }
  for (int i = 0; i < 3; ++i) {
    if (ms_host) ms_host[i] = ms[i];
    if (launches_host) launches_host[i] = n[i];
  }
  if (net_positions_host) *net_positions_host = 0;
  return NZ_OK;
}

}  // extern "C"
