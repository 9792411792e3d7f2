// Fused policy/value network for one tile of 16 positions (gfx950, FP32 MFMA).
// Device function shared by the stand-alone network kernel (net.hip) and the
// persistent self-play kernel (selfplay.hip).
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
#pragma once
#include "engine.h"

namespace nz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int POS = 16;            // positions per workgroup (MFMA M)
constexpr int CELLS = 9;
constexpr int ROW = 64;            // channels per (cell, pos) row
constexpr int NET_WAVES = 4;
constexpr int NET_THREADS = NET_WAVES * 64;
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


// Run the layer program on the 16 positions whose input planes are in
// `inp` ([cell][pos][4 planes]); `lds` holds the two activation buffers.  Every
// thread of the 256-thread workgroup must call it.  Outputs: logits
// [pos][policy_channels][9] and value [pos] for pos < n_valid (any address space).
__device__ __forceinline__ void net_tile(const NetProgram* __restrict__ prog, int n_layers,
                                         const float* __restrict__ W, float* __restrict__ lds,
                                         const float* __restrict__ inp, int policy_channels, int n_valid,
                                         float* logits, float* value) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int pos = lane & 15;     // A row / C column owner
  const int quad = lane >> 4;    // K slice / C row group
  const int tile0 = 0;
  const int count = n_valid;
  for (int L = 0; L < n_layers; ++L) {
    const NetLayer ly = prog->layers[L];
    const float* __restrict__ src = lds + (ly.src & 1) * ACT_FLOATS;

    for (int nt = wave; nt < ly.ntiles; nt += NET_WAVES) {
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

}  // namespace nz
