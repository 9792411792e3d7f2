// Stand-alone network kernel: Network_Manager.inference for a batch of 3x3
// boards (Neural_Networks/Network_Manager.py:46-64), one workgroup per 16
// positions.  The arithmetic is net_dev.hpp's net_tile.
#include "net_dev.hpp"

namespace nz {
namespace {

template <bool STAMPS>
__global__ __launch_bounds__(NET_THREADS) void net_kernel(const NetProgram* __restrict__ prog, int n_layers,
                                                          const float* __restrict__ W,
                                                          const uint32_t* __restrict__ boards,
                                                          const float* __restrict__ states, int in_channels,
                                                          const int32_t* __restrict__ count_ptr, int max_positions,
                                                          int policy_channels, float* __restrict__ logits,
                                                          float* __restrict__ value,
                                                          unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) float lds[NET_LDS_FLOATS];
  float* const inp = lds + NET_BUFFERS * ACT_FLOATS;

  int count = max_positions;
  if (count_ptr != nullptr) {
    const int c = *count_ptr;
    count = c < count ? c : count;
  }
  const int tile0 = blockIdx.x * POS;
  if (tile0 >= count) return;

  // K groups may reach past a narrow layer's channels (their weights are zero): no NaN bit patterns in LDS
  for (int idx = threadIdx.x; idx < NET_BUFFERS * ACT_FLOATS; idx += NET_THREADS) lds[idx] = 0.0f;
  // input planes -> inp[cell][pos][4]
  for (int idx = threadIdx.x; idx < CELLS * POS * 4; idx += NET_THREADS) {
    const int c = idx & 3, pp = (idx >> 2) & 15, cell = idx >> 6;
    const int gp = tile0 + pp;
    float v = 0.0f;
    if (gp < count && c < in_channels) {
      if (boards != nullptr) v = (float)((boards[gp] >> (cell + 16 * c)) & 1u);
      else v = states[((size_t)gp * in_channels + c) * CELLS + cell];
    }
    inp[idx] = v;
  }
  __syncthreads();
  net_tile<STAMPS>(prog, W, lds, inp, policy_channels, count - tile0,
                   logits + (size_t)tile0 * policy_channels * CELLS, value + tile0,
                   STAMPS ? stamps + blockIdx.x * 4 : nullptr);
}

// scipy softmax over each position's logits (Explorer.py:159), numpy sum order for n = 9
__global__ void softmax_kernel(const float* __restrict__ logits, float* __restrict__ probs, int batch, int n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const float* x = logits + (size_t)b * n;
  float m = x[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, x[i]);
  float e[9];
  for (int i = 0; i < n; ++i) e[i] = expf(x[i] - m);
  float s;
  if (n < 8) {
    s = 0.f;
    for (int i = 0; i < n; ++i) s += e[i];
  } else {
    s = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    for (int i = 8; i < n; ++i) s += e[i];
  }
  for (int i = 0; i < n; ++i) probs[(size_t)b * n + i] = e[i] / s;
}

}  // namespace

void launch_net(const NetProgram* prog_dev, int n_layers, const float* packed_weights, const uint32_t* boards,
                const float* states, const int32_t* count_dev, int max_positions, float* logits, float* value,
                float* probs, unsigned long long* stamps, hipStream_t s) {
  const int blocks = (max_positions + POS - 1) / POS;
  if (blocks <= 0) return;
  if (stamps != nullptr)
    hipLaunchKernelGGL(net_kernel<true>, dim3(blocks), dim3(NET_THREADS), 0, s, prog_dev, n_layers, packed_weights,
                       boards, states, 2, count_dev, max_positions, 1, logits, value, stamps);
  else
    hipLaunchKernelGGL(net_kernel<false>, dim3(blocks), dim3(NET_THREADS), 0, s, prog_dev, n_layers, packed_weights,
                       boards, states, 2, count_dev, max_positions, 1, logits, value, stamps);
  if (probs != nullptr)
    hipLaunchKernelGGL(softmax_kernel, dim3((max_positions + 255) / 256), dim3(256), 0, s, logits, probs,
                       max_positions, 9);
}

}  // namespace nz
