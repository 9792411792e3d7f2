// SCS rules as batch operators (C ABI nz_scs_*): one game per thread.  The rules themselves are
// scs_dev.hpp; these kernels only run them over a batch so that they can be checked against the
// oracle step by step (tests/test_gpu_scs.py) before the search is built on them.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nuzero_amd.h"
#include "scs_dev.hpp"

using namespace nz;

struct nz_scs {
  int device = 0, n_games = 0;
  ScsRules host_rules;
  ScsRules* rules = nullptr;                 // one description, or one per game (nz_scs_set_maps)
  int rules_stride = 0;                      // 0: every game reads rules[0]; 1: game g reads rules[g]
  ScsState* states = nullptr;
  std::string error;
};

namespace {
thread_local std::string g_scs_error;

// (`rs`: 0 = one description for every game, 1 = game g has its own, with its own map)
__global__ void scs_reset_kernel(const ScsRules* r, int rs, ScsState* st, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) Scs(r[(size_t)g * rs], st[g]).reset();
}
__global__ void scs_step_kernel(const ScsRules* r, int rs, ScsState* st, const int32_t* actions, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int a = actions[g];
  if (a >= 0 && !st[g].terminal) Scs(r[(size_t)g * rs], st[g]).step(a);
}
__global__ void scs_mask_kernel(const ScsRules* r0, int rs, ScsState* st, int8_t* mask, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const ScsRules* r = r0 + (size_t)g * rs;
  const int na = r->planes * r->tiles;
  int8_t* m = mask + (size_t)g * na;
  for (int i = 0; i < na; ++i) m[i] = 0;
  if (!st[g].terminal) Scs(*r, st[g]).for_each_legal([&](int a) { m[a] = 1; });
}
__global__ void scs_image_kernel(const ScsRules* r0, int rs, ScsState* st, float* img, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const ScsRules* r = r0 + (size_t)g * rs;
  Scs(*r, st[g]).state_image(img + (size_t)g * r->channels * r->tiles);
}
__global__ void scs_status_kernel(const ScsState* st, int32_t* out, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const ScsState& s = st[g];
  int32_t* o = out + g * 7;
  o[0] = s.player; o[1] = s.sub_phase; o[2] = s.stage; o[3] = s.turn; o[4] = s.terminal; o[5] = s.terminal_value;
  o[6] = s.length;
}

nz_status scs_fail(nz_scs* h, nz_status code, const char* fmt, ...) {
  char buf[384];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->error = buf; else g_scs_error = buf;
  return code;
}
#define SCS_HIP(h, call)                                                                        \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) return scs_fail((h), NZ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// hex neighbourhood of SCS_Game.check_tiles (:1048-1094): n, ne, se, s, sw, nw
void neighbours(int rows, int cols, int r, int c, int out[6]) {
  const bool even = c % 2 == 0;
  for (int i = 0; i < 6; ++i) out[i] = -1;
  auto idx = [&](int rr, int cc) { return rr * cols + cc; };
  if (r - 1 != -1) out[0] = idx(r - 1, c);
  if (r + 1 != rows) out[3] = idx(r + 1, c);
  if (!(c == 0 || (r == 0 && even))) out[5] = even ? idx(r - 1, c - 1) : idx(r, c - 1);
  if (!(c == 0 || (r == rows - 1 && !even))) out[4] = even ? idx(r, c - 1) : idx(r + 1, c - 1);
  if (!(c == cols - 1 || (r == 0 && even))) out[1] = even ? idx(r - 1, c + 1) : idx(r, c + 1);
  if (!(c == cols - 1 || (r == rows - 1 && !even))) out[2] = even ? idx(r, c + 1) : idx(r + 1, c + 1);
}
int blocks(int n) { return (n + 127) / 128; }
}  // namespace

// nz_scs_desc -> ScsRules (what SCS_Game.load_game_from_config derives, SCS_Game.py:147-240,1570-1779)
bool nz::scs_fill_rules(const nz_scs_desc* d, ScsRules* out, std::string* err) {
  const int T = d->rows * d->cols;
  auto bad = [&](const char* m) { *err = m; return false; };
  if (d->rows <= 0 || d->cols <= 0 || T > SCS_MAX_TILES) return bad("board larger than 100 tiles");
  if (d->stacking < 1 || d->stacking > SCS_MAX_STACK) return bad("stacking limit must be 1..3");
  if (d->n_units < 1 || d->n_units > SCS_MAX_UNITS) return bad("1..32 units supported");
  if (d->turns < 1 || d->turns >= SCS_MAX_TURNS) return bad("1..15 turns supported");
  if (d->n_vp[0] < 1 || d->n_vp[1] < 1) return bad("each player needs a victory point");
  ScsRules& r = *out;
  memset(&r, 0, sizeof(r));
  const int S = d->stacking;
  r.rows = d->rows; r.cols = d->cols; r.tiles = T; r.turns = d->turns; r.stacking = S; r.n_units = d->n_units;
  r.planes = 1 + 6 * S + 1 + S + 1 + S + S;                     // SCS_Game.py:147-180
  r.placement_limit = 1;
  r.movement_limit = 1 + 6 * S;
  r.target_limit = r.movement_limit + 1;
  r.attackers_limit = r.target_limit + S;
  r.confirm_limit = r.attackers_limit + 1;
  r.no_move_limit = r.confirm_limit + S;
  r.no_fight_limit = r.no_move_limit + S;
  r.channels = 3 + 2 + 2 * 18 + 2 * (3 * S * 3) + 1 + S + 4 + 1 + 1;   // SCS_Game.py:183-240
  for (int t = 0; t < T; ++t) {
    int nb[6];
    neighbours(d->rows, d->cols, t / d->cols, t % d->cols, nb);
    for (int i = 0; i < 6; ++i) r.neighbour[t][i] = (int8_t)nb[i];
    r.attack_mod[t] = d->terrain[t * 3 + 0];
    r.defense_mod[t] = d->terrain[t * 3 + 1];
    r.cost[t] = (int32_t)d->terrain[t * 3 + 2];
    for (int k = 0; k < 3; ++k) r.terrain_f[t][k] = (float)d->terrain[t * 3 + k];
  }
  r.n_vp[0] = d->n_vp[0];
  r.n_vp[1] = d->n_vp[1];
  for (int p = 0, k = 0; p < 2; ++p)
    for (int i = 0; i < d->n_vp[p]; ++i, ++k) r.vp[p][i] = (int8_t)(d->vp[k * 2] * d->cols + d->vp[k * 2 + 1]);
  for (int u = 0; u < d->n_units; ++u) {
    r.u_player[u] = (int8_t)d->units[u * 5 + 0];
    r.u_turn[u] = (int8_t)d->units[u * 5 + 1];
    r.u_attack[u] = (int8_t)d->units[u * 5 + 2];
    r.u_defense[u] = (int8_t)d->units[u * 5 + 3];
    r.u_mov[u] = (int8_t)d->units[u * 5 + 4];
    for (int t = 0; t < T; ++t) r.arrival[u][t] = d->arrival[(size_t)u * T + t];
  }
  return true;
}

// A game's own map over a description: terrain [tiles][3] (attack modifier, defense modifier, cost) and the victory
// points [n_vp[0] + n_vp[1]][2] (row, column), the counts as in the description.
void nz::scs_apply_map(ScsRules* r, const float* terrain, const int32_t* vp) {
  for (int t = 0; t < r->tiles; ++t) {
    r->attack_mod[t] = terrain[t * 3 + 0];
    r->defense_mod[t] = terrain[t * 3 + 1];
    r->cost[t] = (int32_t)terrain[t * 3 + 2];
    for (int k = 0; k < 3; ++k) r->terrain_f[t][k] = terrain[t * 3 + k];
  }
  for (int p = 0, k = 0; p < 2; ++p)
    for (int i = 0; i < r->n_vp[p]; ++i, ++k) r->vp[p][i] = (int8_t)(vp[k * 2] * r->cols + vp[k * 2 + 1]);
}

extern "C" {

// Every game of the batch on its own map (host arrays: terrain float32 [G][tiles][3], vp int32 [G][n_vp0 + n_vp1][2]);
// the games are reset.  NULL terrain: back to the description's one map.
nz_status nz_scs_set_maps(nz_scs* h, const float* terrain_host, const int32_t* vp_host, void* stream) {
  if (!h || (terrain_host && !vp_host)) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  SCS_HIP(h, hipDeviceSynchronize());
  const int want = terrain_host ? h->n_games : 1;
  std::vector<ScsRules> rows((size_t)want, h->host_rules);
  if (terrain_host) {
    const ScsRules& b = h->host_rules;
    const int T = b.tiles, nv = b.n_vp[0] + b.n_vp[1];
    for (int g = 0; g < want; ++g) nz::scs_apply_map(&rows[g], terrain_host + (size_t)g * T * 3, vp_host + (size_t)g * nv * 2);
  }
  ScsRules* dev = nullptr;
  SCS_HIP(h, hipMalloc((void**)&dev, rows.size() * sizeof(ScsRules)));
  if (hipMemcpy(dev, rows.data(), rows.size() * sizeof(ScsRules), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(dev);
    return scs_fail(h, NZ_ERR_HIP, "upload failed");
  }
  (void)hipFree(h->rules);
  h->rules = dev;
  h->rules_stride = terrain_host ? 1 : 0;
  return nz_scs_reset(h, stream);
}

const char* nz_scs_last_error(const nz_scs* h) { return h ? h->error.c_str() : g_scs_error.c_str(); }

nz_status nz_scs_create(nz_scs** out, const nz_scs_desc* d, int32_t n_games, int32_t device) {
  if (!out || !d) return scs_fail(nullptr, NZ_ERR_ARG, "null argument");
  *out = nullptr;
  if (n_games <= 0) return scs_fail(nullptr, NZ_ERR_ARG, "n_games must be positive");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return scs_fail(nullptr, NZ_ERR_HIP, "no HIP device %d (the SCS operators have no CPU fallback)", device);
  nz_scs* h = new nz_scs;
  h->device = device;
  h->n_games = n_games;
  std::string err;
  if (!nz::scs_fill_rules(d, &h->host_rules, &err)) {
    delete h;
    return scs_fail(nullptr, NZ_ERR_ARG, "%s", err.c_str());
  }
  ScsRules& r = h->host_rules;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&h->rules, sizeof(ScsRules)) != hipSuccess ||
      hipMalloc((void**)&h->states, (size_t)n_games * sizeof(ScsState)) != hipSuccess ||
      hipMemcpy(h->rules, &r, sizeof(r), hipMemcpyHostToDevice) != hipSuccess) {
    nz_scs_destroy(h);
    return scs_fail(nullptr, NZ_ERR_HIP, "device allocation failed");
  }
  hipLaunchKernelGGL(scs_reset_kernel, dim3(blocks(n_games)), dim3(128), 0, nullptr, h->rules, 0, h->states, n_games);
  if (hipDeviceSynchronize() != hipSuccess) {
    nz_scs_destroy(h);
    return scs_fail(nullptr, NZ_ERR_HIP, "reset kernel failed");
  }
  *out = h;
  return NZ_OK;
}

void nz_scs_destroy(nz_scs* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->rules) (void)hipFree(h->rules);
  if (h->states) (void)hipFree(h->states);
  delete h;
}

nz_status nz_scs_dims(const nz_scs* h, int32_t* planes, int32_t* rows, int32_t* cols, int32_t* channels) {
  if (!h) return NZ_ERR_ARG;
  if (planes) *planes = h->host_rules.planes;
  if (rows) *rows = h->host_rules.rows;
  if (cols) *cols = h->host_rules.cols;
  if (channels) *channels = h->host_rules.channels;
  return NZ_OK;
}

nz_status nz_scs_reset(nz_scs* h, void* stream) {
  if (!h) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_reset_kernel, dim3(blocks(h->n_games)), dim3(128), 0, (hipStream_t)stream, h->rules, h->rules_stride, h->states,
                     h->n_games);
  SCS_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_step(nz_scs* h, const int32_t* actions_dev, void* stream) {
  if (!h || !actions_dev) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_step_kernel, dim3(blocks(h->n_games)), dim3(128), 0, (hipStream_t)stream, h->rules, h->rules_stride, h->states,
                     actions_dev, h->n_games);
  SCS_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_legal_mask(nz_scs* h, int8_t* mask_dev, void* stream) {
  if (!h || !mask_dev) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_mask_kernel, dim3(blocks(h->n_games)), dim3(128), 0, (hipStream_t)stream, h->rules, h->rules_stride, h->states,
                     mask_dev, h->n_games);
  SCS_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_state_image(nz_scs* h, float* image_dev, void* stream) {
  if (!h || !image_dev) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_image_kernel, dim3(blocks(h->n_games)), dim3(128), 0, (hipStream_t)stream, h->rules, h->rules_stride, h->states,
                     image_dev, h->n_games);
  SCS_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_scs_status(nz_scs* h, int32_t* status_dev, void* stream) {
  if (!h || !status_dev) return NZ_ERR_ARG;
  SCS_HIP(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(scs_status_kernel, dim3(blocks(h->n_games)), dim3(128), 0, (hipStream_t)stream, h->states,
                     status_dev, h->n_games);
  SCS_HIP(h, hipGetLastError());
  return NZ_OK;
}

}  // extern "C"
