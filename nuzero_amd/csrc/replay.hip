// Replay buffer on the device (C ABI nz_replay_*): the positions of finished self-play games stay in HBM, where the
// engine's export buffers already are, and training batches are assembled by one gather kernel.
//
// Replaces, for the data path, Training/ReplayBuffer.py:24-53 (save_game's per-position tuples, get_slice,
// get_sample) and the per-sample Python work of Training/AlphaZero.py:846-852,892-903 (torch.cat of the states,
// torch.tensor of every target policy): a position is a row {state [state_floats] f32, policy [num_actions] f32,
// value f32, game_index i32} in a physical slot; WHICH slot a new position takes and which slots a batch reads is
// decided on the host (nuzero_amd/replay_device.py: window in games with per-position eviction, random.shuffle,
// np.random.choice -- the reference's own generators, so the same seeds give the same batches), the kernels below
// only move data.  HBM-bound copies: one workgroup per row, coalesced 4-byte accesses.
//
// Policy targets are formed exactly as the games do (tic_tac_toe.py:177-182, SCS_Game.py:1517-1521:
// visit / sum(visits) in double precision for the root's children, 0 elsewhere) and rounded to float32 the way
// torch.tensor(list of Python floats) does (AlphaZero.py:901).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/nuzero_amd.h"

struct nz_replay {
  int device = 0;
  int64_t capacity = 0;
  int32_t state_floats = 0, num_actions = 0;
  float* states = nullptr;      // [capacity][state_floats]
  float* policies = nullptr;    // [capacity][num_actions]
  float* values = nullptr;      // [capacity]
  int32_t* game_index = nullptr;
  std::string error;
};

namespace {
thread_local std::string g_err;
nz_status rfail(nz_replay* h, nz_status code, const char* fmt, ...) {
  char buf[384];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->error = buf; else g_err = buf;
  return code;
}
#define R_HIP(h, call)                                                                             \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess) return rfail((h), NZ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

struct AppendArgs {
  float* states; float* policies; float* values; int32_t* game_index;   // the buffer
  const float* src_states;       // [N][state_floats]
  const int32_t* visits;         // dense: [N][A] visit counts, or nullptr
  const float* src_policies;     // ready-made float32 policies [N][A], or nullptr
  const int32_t* child_action;   // sparse: [N][max_children]
  const int32_t* child_visit;    //         [N][max_children]
  const int32_t* n_children;     //         [N]
  const int32_t* row_value;      // [N / rows_per_game] terminal value of the row's game
  const int64_t* dst_slot;       // [N] physical slot, -1: row is not stored
  int32_t state_floats, num_actions, max_children, rows_per_game, game_index_value;
  int64_t capacity;
  int32_t* error_flag;
};

// one workgroup per source row
__global__ __launch_bounds__(256) void append_kernel(AppendArgs a) {
  const int64_t r = blockIdx.x;
  const int64_t slot = a.dst_slot[r];
  if (slot < 0) return;
  if (slot >= a.capacity) {
    if (threadIdx.x == 0) atomicOr(a.error_flag, 1);
    return;
  }
  const int tid = threadIdx.x;
  const float* s = a.src_states + r * a.state_floats;
  float* d = a.states + slot * a.state_floats;
  for (int i = tid; i < a.state_floats; i += 256) d[i] = s[i];
  float* pol = a.policies + slot * a.num_actions;
  const int A = a.num_actions;
  if (a.src_policies != nullptr) {
    for (int i = tid; i < A; i += 256) pol[i] = a.src_policies[r * A + i];
  } else {
    // sum of the root's children's visits: an integer, exact in any order
    __shared__ long long s_total;
    if (tid == 0) s_total = 0;
    __syncthreads();
    long long part = 0;
    if (a.visits != nullptr) {
      for (int i = tid; i < A; i += 256) part += a.visits[r * A + i];
    } else {
      const int k = a.n_children[r];
      for (int i = tid; i < k; i += 256) part += a.child_visit[r * a.max_children + i];
    }
    if (part) atomicAdd((unsigned long long*)&s_total, (unsigned long long)part);
    __syncthreads();
    const double total = (double)s_total;
    if (a.visits != nullptr) {
      for (int i = tid; i < A; i += 256) {
        const int v = a.visits[r * A + i];
        pol[i] = v ? (float)((double)v / total) : 0.0f;
      }
    } else {
      for (int i = tid; i < A; i += 256) pol[i] = 0.0f;
      __syncthreads();
      const int k = a.n_children[r];
      for (int i = tid; i < k; i += 256) {
        const int act = a.child_action[r * a.max_children + i];
        if (act < 0 || act >= A) { atomicOr(a.error_flag, 2); continue; }
        pol[act] = (float)((double)a.child_visit[r * a.max_children + i] / total);
      }
    }
  }
  if (tid == 0) {
    a.values[slot] = (float)a.row_value[r / a.rows_per_game];
    a.game_index[slot] = a.game_index_value;
  }
}

struct GatherArgs {
  const float* states; const float* policies; const float* values; const int32_t* game_index;
  const int64_t* slots;          // [B]
  float* out_states; float* out_policies; float* out_values; int32_t* out_game_index;
  int32_t state_floats, num_actions;
  int64_t capacity;
  int32_t* error_flag;
};

// one workgroup per batch row
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
  const int64_t b = blockIdx.x;
  const int64_t slot = a.slots[b];
  if (slot < 0 || slot >= a.capacity) {
    if (threadIdx.x == 0) atomicOr(a.error_flag, 4);
    return;
  }
  const int tid = threadIdx.x;
  if (a.out_states)
    for (int i = tid; i < a.state_floats; i += 256) a.out_states[b * a.state_floats + i] = a.states[slot * a.state_floats + i];
  if (a.out_policies)
    for (int i = tid; i < a.num_actions; i += 256) a.out_policies[b * a.num_actions + i] = a.policies[slot * a.num_actions + i];
  if (tid == 0) {
    if (a.out_values) a.out_values[b] = a.values[slot];
    if (a.out_game_index) a.out_game_index[b] = a.game_index[slot];
  }
}

}  // namespace

extern "C" {

const char* nz_replay_last_error(const nz_replay* h) { return h ? h->error.c_str() : g_err.c_str(); }

nz_status nz_replay_create(nz_replay** out, int64_t capacity, int32_t state_floats, int32_t num_actions, int32_t device) {
  if (!out) return rfail(nullptr, NZ_ERR_ARG, "null argument");
  *out = nullptr;
  if (capacity <= 0 || state_floats <= 0 || num_actions <= 0) return rfail(nullptr, NZ_ERR_ARG, "bad sizes");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return rfail(nullptr, NZ_ERR_HIP, "no HIP device %d (no CPU fallback)", device);
  (void)hipSetDevice(device);
  nz_replay* h = new nz_replay;
  h->device = device; h->capacity = capacity; h->state_floats = state_floats; h->num_actions = num_actions;
  const size_t n = (size_t)capacity;
  if (hipMalloc((void**)&h->states, n * state_floats * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&h->policies, n * num_actions * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&h->values, (n + 1) * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&h->game_index, (n + 1) * sizeof(int32_t)) != hipSuccess) {
    nz_replay_destroy(h);
    return rfail(nullptr, NZ_ERR_HIP, "device allocation failed (%lld positions)", (long long)capacity);
  }
  // the word after the last game index is the kernels' error flag
  if (hipMemset(h->game_index + n, 0, sizeof(int32_t)) != hipSuccess) {
    nz_replay_destroy(h);
    return rfail(nullptr, NZ_ERR_HIP, "memset failed");
  }
  *out = h;
  return NZ_OK;
}

void nz_replay_destroy(nz_replay* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(h->states); (void)hipFree(h->policies); (void)hipFree(h->values); (void)hipFree(h->game_index);
  delete h;
}

nz_status nz_replay_append(nz_replay* h, const float* states_dev, const int32_t* visits_dev, const float* policies_dev,
                           const int32_t* child_action_dev, const int32_t* child_visit_dev, const int32_t* n_children_dev,
                           int32_t max_children, const int32_t* game_value_dev, int32_t rows_per_game,
                           const int64_t* dst_slot_dev, int64_t n_rows, int32_t game_index, void* stream) {
  if (!h || !states_dev || !game_value_dev || !dst_slot_dev) return NZ_ERR_ARG;
  const int modes = (visits_dev != nullptr) + (policies_dev != nullptr) + (child_action_dev != nullptr);
  if (modes != 1) return rfail(h, NZ_ERR_ARG, "exactly one of visits / policies / child lists must be given");
  if (child_action_dev && (!child_visit_dev || !n_children_dev || max_children <= 0))
    return rfail(h, NZ_ERR_ARG, "child lists need actions, visits, counts and their row length");
  if (rows_per_game <= 0) return rfail(h, NZ_ERR_ARG, "rows_per_game must be positive");
  if (n_rows <= 0) return NZ_OK;
  R_HIP(h, hipSetDevice(h->device));
  AppendArgs a{h->states, h->policies, h->values, h->game_index, states_dev, visits_dev, policies_dev, child_action_dev,
               child_visit_dev, n_children_dev, game_value_dev, dst_slot_dev, h->state_floats, h->num_actions, max_children,
               rows_per_game, game_index, h->capacity, h->game_index + h->capacity};
  hipLaunchKernelGGL(append_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, a);
  R_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_replay_gather(nz_replay* h, const int64_t* slots_dev, int64_t batch, float* states_out, float* policies_out,
                           float* values_out, int32_t* game_index_out, void* stream) {
  if (!h || !slots_dev) return NZ_ERR_ARG;
  if (batch <= 0) return NZ_OK;
  R_HIP(h, hipSetDevice(h->device));
  GatherArgs a{h->states, h->policies, h->values, h->game_index, slots_dev, states_out, policies_out, values_out,
               game_index_out, h->state_floats, h->num_actions, h->capacity, h->game_index + h->capacity};
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);
  R_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_replay_check(nz_replay* h, void* stream) {
  if (!h) return NZ_ERR_ARG;
  R_HIP(h, hipSetDevice(h->device));
  int32_t f = 0;
  R_HIP(h, hipMemcpyAsync(&f, h->game_index + h->capacity, sizeof(f), hipMemcpyDeviceToHost, (hipStream_t)stream));
  R_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  if (f) return rfail(h, NZ_ERR_OVERFLOW, "device check failed (flag %d: 1 slot beyond capacity, 2 action out of range, "
                                          "4 batch slot out of range)", f);
  return NZ_OK;
}

nz_status nz_replay_dims(const nz_replay* h, int64_t* capacity, int32_t* state_floats, int32_t* num_actions) {
  if (!h) return NZ_ERR_ARG;
  if (capacity) *capacity = h->capacity;
  if (state_floats) *state_floats = h->state_floats;
  if (num_actions) *num_actions = h->num_actions;
  return NZ_OK;
}

}  // extern "C"
