// Fused policy/value network kernel (gfx950, FP32 MFMA).
//
// Computes Network_Manager.inference (Neural_Networks/Network_Manager.py:46-64)
// for the square-conv RecurrentNet (Neural_Networks/Architectures/
// RecurrentNet.py:82-99; BasicBlock blocks.py:37-41; Reduce_PolicyHead
// blocks.py:130-170; Reduce_ValueHead blocks.py:46-92) on a batch of 3x3 boards:
// every conv is 3x3, stride 1, zero 'same' padding, bias-free.
//
// One workgroup runs the WHOLE network for 16 positions; activations never
// leave LDS and the weights stream from L2 straight into registers.
//
// Mapping.  On a 3x3 board with a 3x3 kernel every output cell o sees input
// cell i through exactly one tap (tap = i - o + centre) when |dy|,|dx| <= 1, so
// a conv layer is, per output cell, a dense [16 positions] x [C_in * n_valid(o)]
// x [C_out] product.  The zero-padding taps are never multiplied: 49 of the 81
// (cell, tap) pairs exist.  v_mfma_f32_16x16x4_f32 takes M = 16 positions,
// N = 16 output channels, K = 4 input channels; lane l supplies A[pos = l & 15]
// [k = l >> 4] and B[k = l >> 4][cout = l & 15] and receives C[pos = 4*(l>>4)+r]
// [cout = l & 15] in register r.  FP32 inputs/accumulation are required by the
// 1e-5 parity tolerance; the MFMA result is an exact k-ordered fmaf chain, and
// each position's row is independent of the others, so a position's outputs do
// not depend on which batch slot it occupies.
//
// Per 16-channel K group a wave loads 9 activation vectors (one ds_read_b128
// per input cell) and 9 weight vectors (one global_load_dwordx4 per tap) and
// issues 196 MFMAs from them.
//
// LDS layout: act[buf][cell][pos][64 ch], the 16-byte slot index XOR-ed with
// the position so that ds_read_b128 (A operands) and ds_write_b32 (epilogue) are
// bank-conflict free.
#include "engine.h"

namespace nz {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int POS = 16;            // positions per workgroup (MFMA M)
constexpr int CELLS = 9;
constexpr int ROW = 64;            // channels per (cell, pos) row
constexpr int NWAVES = 4;
constexpr int NTHREADS = NWAVES * 64;
constexpr int ACT_FLOATS = CELLS * POS * ROW;
constexpr int INP_FLOATS = CELLS * POS * 4;

__device__ __forceinline__ int act_addr(int cell, int pos, int ch) {
  return ((cell * POS + pos) << 6) + ((((ch >> 2) ^ pos) & 15) << 2) + (ch & 3);
}

template <int I, int TAP>
struct TapMap {   // output cell reached from input cell I through tap TAP
  static constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
  static constexpr int oy = I / 3 - dy, ox = I % 3 - dx;
  static constexpr bool valid = oy >= 0 && oy < 3 && ox >= 0 && ox < 3;
  static constexpr int o = valid ? oy * 3 + ox : 0;
};

template <int J, int TAP, int I>
__device__ __forceinline__ void mfma_pair(f32x4 (&acc)[CELLS], const f32x4 (&av)[CELLS], const f32x4 (&bw)[9]) {
  if constexpr (TapMap<I, TAP>::valid) {
    constexpr int o = TapMap<I, TAP>::o;
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[I][J], bw[TAP][J], acc[o], 0, 0, 0);
  }
}
template <int J, int TAP>
__device__ __forceinline__ void mfma_tap(f32x4 (&acc)[CELLS], const f32x4 (&av)[CELLS], const f32x4 (&bw)[9]) {
  mfma_pair<J, TAP, 0>(acc, av, bw); mfma_pair<J, TAP, 1>(acc, av, bw); mfma_pair<J, TAP, 2>(acc, av, bw);
  mfma_pair<J, TAP, 3>(acc, av, bw); mfma_pair<J, TAP, 4>(acc, av, bw); mfma_pair<J, TAP, 5>(acc, av, bw);
  mfma_pair<J, TAP, 6>(acc, av, bw); mfma_pair<J, TAP, 7>(acc, av, bw); mfma_pair<J, TAP, 8>(acc, av, bw);
}
template <int J>
__device__ __forceinline__ void mfma_step(f32x4 (&acc)[CELLS], const f32x4 (&av)[CELLS], const f32x4 (&bw)[9]) {
  // tap-major: consecutive MFMAs accumulate into different output cells
  mfma_tap<J, 0>(acc, av, bw); mfma_tap<J, 1>(acc, av, bw); mfma_tap<J, 2>(acc, av, bw);
  mfma_tap<J, 3>(acc, av, bw); mfma_tap<J, 4>(acc, av, bw); mfma_tap<J, 5>(acc, av, bw);
  mfma_tap<J, 6>(acc, av, bw); mfma_tap<J, 7>(acc, av, bw); mfma_tap<J, 8>(acc, av, bw);
}

template <int TAP, int I>
__device__ __forceinline__ void mfma_extra_pair(f32x4 (&acc)[CELLS], const float (&ax)[CELLS], const float (&bx)[9]) {
  if constexpr (TapMap<I, TAP>::valid) {
    constexpr int o = TapMap<I, TAP>::o;
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[I], bx[TAP], acc[o], 0, 0, 0);
  }
}
template <int TAP>
__device__ __forceinline__ void mfma_extra_tap(f32x4 (&acc)[CELLS], const float (&ax)[CELLS], const float (&bx)[9]) {
  mfma_extra_pair<TAP, 0>(acc, ax, bx); mfma_extra_pair<TAP, 1>(acc, ax, bx); mfma_extra_pair<TAP, 2>(acc, ax, bx);
  mfma_extra_pair<TAP, 3>(acc, ax, bx); mfma_extra_pair<TAP, 4>(acc, ax, bx); mfma_extra_pair<TAP, 5>(acc, ax, bx);
  mfma_extra_pair<TAP, 6>(acc, ax, bx); mfma_extra_pair<TAP, 7>(acc, ax, bx); mfma_extra_pair<TAP, 8>(acc, ax, bx);
}

__global__ __launch_bounds__(NTHREADS) void net_kernel(const NetProgram* __restrict__ prog, int n_layers,
                                                       const float* __restrict__ W,
                                                       const uint32_t* __restrict__ boards,
                                                       const float* __restrict__ states, int in_channels,
                                                       const int32_t* __restrict__ count_ptr, int max_positions,
                                                       int policy_channels, float* __restrict__ logits,
                                                       float* __restrict__ value) {
  __shared__ __attribute__((aligned(16))) float lds[2 * ACT_FLOATS + INP_FLOATS];
  float* const inp = lds + 2 * ACT_FLOATS;

  int count = max_positions;
  if (count_ptr != nullptr) {
    const int c = *count_ptr;
    count = c < count ? c : count;
  }
  const int tile0 = blockIdx.x * POS;
  if (tile0 >= count) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int pos = lane & 15;     // A row / C column owner
  const int quad = lane >> 4;    // K slice / C row group

  // ---- input planes -> inp[cell][pos][4] --------------------------------------
  for (int idx = tid; idx < CELLS * POS * 4; idx += NTHREADS) {
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

  for (int L = 0; L < n_layers; ++L) {
    const NetLayer ly = prog->layers[L];
    const float* __restrict__ src = lds + (ly.src & 1) * ACT_FLOATS;

    for (int nt = wave; nt < ly.ntiles; nt += NWAVES) {
      f32x4 acc[CELLS];
#pragma unroll
      for (int o = 0; o < CELLS; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};

      // main channels, 16 per group
      const f32x4* __restrict__ wl = reinterpret_cast<const f32x4*>(W + ly.w_off);
      for (int kg = 0; kg < ly.kgroups; ++kg) {
        f32x4 av[CELLS], bw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) bw[t] = wl[((size_t)(t * ly.kgroups + kg) * ly.ntiles + nt) * 64 + lane];
#pragma unroll
        for (int i = 0; i < CELLS; ++i)
          av[i] = *reinterpret_cast<const f32x4*>(src + act_addr(i, pos, kg * 16 + quad * 4));
        mfma_step<0>(acc, av, bw);
        mfma_step<1>(acc, av, bw);
        mfma_step<2>(acc, av, bw);
        mfma_step<3>(acc, av, bw);
      }
      // the (<= 4) raw input planes as one extra K step (projection / recall conv)
      if (ly.extra) {
        float ax[CELLS], bx[9];
        const float* __restrict__ wx = W + ly.wx_off;
#pragma unroll
        for (int t = 0; t < 9; ++t) bx[t] = wx[(size_t)(t * ly.ntiles + nt) * 64 + lane];
#pragma unroll
        for (int i = 0; i < CELLS; ++i) ax[i] = inp[(i * POS + pos) * 4 + quad];
        mfma_extra_tap<0>(acc, ax, bx); mfma_extra_tap<1>(acc, ax, bx); mfma_extra_tap<2>(acc, ax, bx);
        mfma_extra_tap<3>(acc, ax, bx); mfma_extra_tap<4>(acc, ax, bx); mfma_extra_tap<5>(acc, ax, bx);
        mfma_extra_tap<6>(acc, ax, bx); mfma_extra_tap<7>(acc, ax, bx); mfma_extra_tap<8>(acc, ax, bx);
      }

      // ---- epilogue: lane holds C[pos = 4*quad + r][cout = 16*nt + (lane & 15)]
      const int cout = nt * 16 + (lane & 15);
      if (ly.dst < 2) {
        float* __restrict__ dst = lds + ly.dst * ACT_FLOATS;
        const float* __restrict__ res = ly.res >= 0 ? lds + ly.res * ACT_FLOATS : nullptr;
#pragma unroll
        for (int o = 0; o < CELLS; ++o) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = act_addr(o, quad * 4 + r, cout);
            float v = acc[o][r];
            if (res != nullptr) v += res[a];
            if (ly.act == 1) v = fmaxf(v, 0.0f);
            else if (ly.act == 2) v = tanhf(v);
            dst[a] = v;
          }
        }
      } else if (ly.dst == 2) {          // policy logits [B][P][9]
        if (cout < policy_channels) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gp = tile0 + quad * 4 + r;
            if (gp < count) {
#pragma unroll
              for (int o = 0; o < CELLS; ++o)
                logits[((size_t)gp * policy_channels + cout) * CELLS + o] = acc[o][r];
            }
          }
        }
      } else {                           // value: mean over (C=1,H,W), tanh (blocks.py:82-84)
        if (cout == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gp = tile0 + quad * 4 + r;
            if (gp < count) {
              float s = 0.0f;
#pragma unroll
              for (int o = 0; o < CELLS; ++o) s += acc[o][r];
              value[gp] = tanhf(s / 9.0f);
            }
          }
        }
      }
    }
    __syncthreads();
  }
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
                float* probs, hipStream_t s) {
  // in_channels / policy_channels for Tic-Tac-Toe nets; generalised with the game descriptor later
  const int blocks = (max_positions + POS - 1) / POS;
  if (blocks <= 0) return;
  hipLaunchKernelGGL(net_kernel, dim3(blocks), dim3(NTHREADS), 0, s, prog_dev, n_layers, packed_weights, boards,
                     states, 2, count_dev, max_positions, 1, logits, value);
  if (probs != nullptr)
    hipLaunchKernelGGL(softmax_kernel, dim3((max_positions + 255) / 256), dim3(256), 0, s, logits, probs,
                       max_positions, 9);
}

}  // namespace nz
