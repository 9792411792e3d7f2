// Policy/value network for boards of any size (the SCS maps: 5x5 ... 10x10, 86+ input planes):
// the square-conv (hex=False) RecurrentNet / ResNet / ConvNet of the reference evaluated on the
// gathered leaf batch of a simulation wave (C ABI nz_boardnet_*).
//
// The fused whole-network kernel of net_dev.hpp keeps a 3x3 board's activations in LDS; a 10x10
// board with 256 filters does not fit, so here every convolution is its own launch and the
// activations live in HBM/L2 as rows of channels (contiguous, padded to a multiple of 16).  Row
// order: positions in groups of 16, row = (group * H*W + cell) * 16 + position-in-group, so the
// 16 rows of one MFMA tile are ONE board cell of 16 positions.  A convolution is an implicit
// GEMM on the FP32 matrix cores: rows = board cells of all positions, K = 9 taps x input
// channels, columns = output channels.
//   * a tap that falls off the board (zero 'same' padding, RecurrentNet.py:47-52) does so for
//     the whole tile, so it is skipped: the MFMAs executed are exactly the algorithmic ones
//     (169 of 225 (cell, tap) pairs on a 5x5 board);
//   * A operand (activations): lane (row r = lane & 15, k-quarter q = lane >> 4) loads the four
//     channels 16 kg + 4 q .. + 3 of the tap's neighbour cell as one 16-byte load;
//   * B operand (weights): packed on the host per (column tile, tap, channel group) in exactly
//     the per-lane order v_mfma_f32_16x16x4_f32 wants, so one coalesced 16-byte load per lane.
//   * one wavefront owns MT x 16 rows and NT x 16 output channels; a workgroup is 4 wavefronts
//     on consecutive row tiles that share the weight stream through L1/L2.
//   * epilogue: residual add (blocks.py:37-41), activation, NHWC store.
// torch.cat([thought, x]) of the recall connection (RecurrentNet.py:91) is never materialised:
// the K loop runs over two sources.
//
// Reference (paths relative to the reference repo): Neural_Networks/Architectures/
// RecurrentNet.py:18-99, ResNet.py:13-70, ConvNet.py:12-57, blocks.py:12-41 (BasicBlock),
// 46-92 (Reduce_ValueHead), 130-170 (Reduce_PolicyHead); Network_Manager.py:46-64 (inference);
// Search/Explorer.py:158-162 (softmax over all logits, value .item()).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/nuzero_amd.h"
#include "fused16_dev.hpp"
#include "boardnet_internal.h"

using namespace nz;

namespace {


struct ConvArgs {
  const float* src0;     // [rows][c0] NHWC, c0 a multiple of 16
  const float* src1;     // second K source (recall concat) or nullptr
  const float* w;        // packed [ntile][taps + 1][kg0 + kg1][64 lanes][4]
  const uint32_t* ws;    // wide layers: weights as three bf16 pieces, [cout tile of 128][taps + 1][32-channel group]
                         // [piece][column tile 8][64 lanes][8 bf16], or nullptr
  const float* res;      // residual [rows][cd] or nullptr
  float* dst;            // [rows][cd]
  const int32_t* n_dev;  // live positions on the device (nullptr: n_host)
  int32_t n_host, hw, h, wd;
  int32_t c0, c1;        // K extent of each source in channels (multiples of 16)
  int32_t s0, s1, cd;    // channel strides of the sources and of dst
  int32_t act;           // 0 none, 1 relu, 2 tanh, 3 elu
  int32_t hex;           // 0: 3x3 square taps; 1: 7 hexagonal taps (centre + 6 neighbours, odd columns shifted down)
};

__device__ __forceinline__ float activate(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return tanhf(v);
    case 3: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// K steps in flight per wavefront: the loads of step i + DEPTH are issued before the MFMAs of step i.
template <int MT, int NT, int DEPTH, bool HEX>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_pos = p.n_dev ? *p.n_dev : p.n_host;
  const int n_groups = (n_pos + 15) >> 4;
  // one wavefront: board cell `cell` of the position groups g0 .. g0 + MT - 1
  const int task = blockIdx.x * 4 + wave;
  const int cell = task % p.hw, g0 = (task / p.hw) * MT;
  if (g0 >= n_groups) return;                        // uniform per wavefront
  const int nt0 = blockIdx.y * NT;
  const int kg0 = p.c0 >> 4, kg1 = p.src1 ? (p.c1 >> 4) : 0, kgt = kg0 + kg1;

  // Taps.  Square: tap = 3 (dy + 1) + (dx + 1).  Hexagonal (hexagdly's addressing: odd columns sit half a cell
  // lower): 0 N, 1 centre, 2 S (same column), 3 NW, 4 SW (column - 1), 5 NE, 6 SE (column + 1); the upper / lower
  // neighbour in an adjacent column is row - 1 / row for a cell in an even column, row / row + 1 in an odd one.
  constexpr int ntaps = HEX ? 7 : 9;
  const int cy = cell / p.wd, cx = cell % p.wd;
  auto tap_dy = [&](int tap) { return HEX ? (tap < 3 ? tap - 1 : ((tap - 3) & 1) - 1 + (cx & 1)) : tap / 3 - 1; };
  auto tap_dx = [&](int tap) { return HEX ? (tap < 3 ? 0 : (tap < 5 ? -1 : 1)) : tap % 3 - 1; };
  uint32_t vmask = 0;                                // taps that stay on the board for this cell
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int y = cy + tap_dy(tap), x = cx + tap_dx(tap);
    if (tap < ntaps && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.wd) vmask |= 1u << tap;
  }
  int row[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    live[m] = g0 + m < n_groups;
    row[m] = ((g0 + m) * p.hw + cell) * 16 + (lane & 15);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // packed weights: [column tile][ntaps + 1][kgt][64 lanes][4]; "tap ntaps" is a block of zeros that the
  // steps past the end of the K loop read, so the pipeline needs no conditional loads
  const size_t tile_stride = (size_t)(ntaps + 1) * kgt * 256;
  const float* wbase = p.w + (size_t)nt0 * tile_stride + lane * 4;
  const int q4 = (lane >> 4) * 4;

  const int total = __popc(vmask) * kgt;
  const int rounds = (total + DEPTH - 1) / DEPTH;
  uint32_t taps_left = vmask;
  int tap = __ffs(taps_left) - 1, kg = 0;            // the centre tap is always on the board
  taps_left &= taps_left - 1;

  f32x4 a[DEPTH][MT], b[DEPTH][NT];
  auto issue = [&](int d) {
#ifdef NZ_ABLATE_CONV_NOLOAD       // timing experiment: no memory operands at all
    {
      const float f = (float)(kg + tap);
#pragma unroll
      for (int m = 0; m < MT; ++m) a[d][m] = f32x4{f, f, f, f};
#pragma unroll
      for (int n = 0; n < NT; ++n) b[d][n] = f32x4{f, f, f, f};
      if (++kg == kgt) {
        kg = 0;
        if (taps_left) { tap = __ffs(taps_left) - 1; taps_left &= taps_left - 1; }
        else tap = ntaps;
      }
      return;
    }
#endif
    const int shift = (tap_dy(tap) * p.wd + tap_dx(tap)) * 16;
    const bool second = kg >= kg0;
    const float* src = second ? p.src1 : p.src0;
    const int cs = second ? p.s1 : p.s0;
    const int ch = (second ? kg - kg0 : kg) * 16 + q4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#ifdef NZ_ABLATE_CONV_A            // timing experiment: every activation load hits the same rows
      if (live[m] && tap < ntaps) a[d][m] = *reinterpret_cast<const f32x4*>(src + (size_t)row[m] * cs + q4);
#else
      if (live[m] && tap < ntaps) a[d][m] = *reinterpret_cast<const f32x4*>(src + (size_t)(row[m] + shift) * cs + ch);
#endif
      else a[d][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#ifdef NZ_ABLATE_CONV_B            // timing experiment: every weight load hits the same line
    const float* wk = wbase;
#else
    const float* wk = wbase + ((size_t)tap * kgt + kg) * 256;
#endif
#pragma unroll
    for (int n = 0; n < NT; ++n) b[d][n] = *reinterpret_cast<const f32x4*>(wk + n * tile_stride);
    if (++kg == kgt) {
      kg = 0;
      if (taps_left) { tap = __ffs(taps_left) - 1; taps_left &= taps_left - 1; }
      else tap = ntaps;
    }
  };
  auto mac = [&](int d) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][m][j], b[d][n][j], acc[m][n], 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      mac(d);
      issue(d);
    }
  }

  const int col = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (!live[m]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = ((g0 + m) * p.hw + cell) * 16 + r4 + r;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const size_t o = (size_t)orow * p.cd + (nt0 + n) * 16 + col;
        float v = acc[m][n][r];
        if (p.res) v += p.res[o];
        p.dst[o] = activate(v, p.act);
      }
    }
  }
}

// ---- wide layers (output channels a multiple of 128) on the BF16 matrix cores ----------------------
// Same arithmetic as the fused 3x3 network (net_dev.hpp): every float32 product from six bf16 MFMA terms
// on exact three-way splits, float32 accumulation -- 96 cycles per 32 channels instead of 256.  At that
// rate the operands cannot come straight from L2 any more, so a workgroup (4 wavefronts) computes a
// 256-position x 128-channel tile of ONE board cell: per K step (tap, 32 input channels) it stages the
// 256 x 32 activations (read as float32, split once, 3 x 16 KB of bf16) and the 128 x 32 weights (already
// split on the host, 24 KB) in LDS, double buffered, one barrier per step; each wavefront owns 128
// positions x 64 channels = 32 accumulator tiles and issues 192 MFMAs per step from 36 LDS fragment reads.
// Weights are the MFMA's A operand, so a lane ends up with four consecutive channels of one position:
// the epilogue is one 16-byte store (and residual load) per accumulator tile.
#ifndef NZ_WIDE_INTERLEAVE
#define NZ_WIDE_INTERLEAVE 0      // 1: the four column tiles' MFMA chains term by term (measured: no difference)
#endif
#ifdef NZ_ABLATE_WIDE_MFMA        // timing experiment: no matrix instructions (results wrong)
#define WIDE_MFMA(w, x, c) f32x4{(c)[0] + __builtin_bit_cast(float, (w)[0] & (x)[0]), (c)[1], (c)[2], (c)[3]}
#else
#define WIDE_MFMA(w, x, c) wide_mfma(w, x, c)
#endif
#ifndef NZ_WIDE_OVERLAP
#define NZ_WIDE_OVERLAP 2         // 2: the next step's staging in the gaps of this step's MFMAs, operands fetched two steps ahead
#endif
#ifndef NZ_WIDE_XCD
#define NZ_WIDE_XCD 1             // workgroups that share an XCD take consecutive tiles
#endif
#ifndef NZ_WIDE_KQ_OUTER
#define NZ_WIDE_KQ_OUTER 1        // K loop: channel groups outside, taps inside
#endif
#if NZ_WIDE_OVERLAP == 2 && !NZ_WIDE_KQ_OUTER
#error "the two-steps-ahead loop moves its cursor in channel-group order"
#endif
constexpr int WIDE_GROUPS = 16;          // position groups (of 16) per workgroup tile
constexpr int WIDE_NT = 8;               // 16-channel column tiles per workgroup tile

#ifdef NZ_WIDE_STAMPS      // diagnostic build: where a K step of conv_wide_kernel goes (workgroup 0's wavefront 0; nz_boardnet_wide_stamps)
__device__ unsigned long long g_wide_stamps[8];
#define WIDE_STAMP(slot) { if (stamping) { const unsigned long long now = __builtin_amdgcn_s_memtime(); wtk[slot] += now - wts; wts = now; } }
#else
#define WIDE_STAMP(slot)
#endif
template <bool HEX>
__global__ __launch_bounds__(256) void conv_wide_kernel(ConvArgs p) {
  __shared__ u32x4 sA[2][3][WIDE_GROUPS * 16][4];      // [buffer][piece][position][16-byte slot, swizzled]
  __shared__ u32x4 sB[2][3][WIDE_NT][64];              // [buffer][piece][column tile][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_pos = p.n_dev ? *p.n_dev : p.n_host;
  const int n_groups = (n_pos + 15) >> 4;
  // Which tile: the dispatcher deals consecutive workgroups round the eight XCDs (each with its own L2), so the ids that
  // share an XCD are given CONSECUTIVE tiles -- one 128-channel tile's weights and neighbouring cells of one position
  // block: with the K loop below running a 32-channel group's taps back to back, a cell's slice fetched for one tap is
  // in that L2 when the neighbours' workgroups want it for theirs (25 cells x 9 taps = 45 cells' slices instead of 225).
  int bid = blockIdx.x + gridDim.x * blockIdx.y;
#if NZ_WIDE_XCD
  {
    const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
#endif
  const int ct = bid / (int)gridDim.x;                 // tile of 128 output channels
  const int bx = bid - ct * (int)gridDim.x;
  const int cell = bx % p.hw, gbase = (bx / p.hw) * WIDE_GROUPS;
  if (gbase >= n_groups) return;                       // uniform per workgroup
  const int kq0 = p.c0 >> 5, kq1 = p.src1 ? (p.c1 >> 5) : 0, kqt = kq0 + kq1;      // 32-channel groups

  constexpr int ntaps = HEX ? 7 : 9;
  const int cy = cell / p.wd, cx = cell % p.wd;
  auto tap_dy = [&](int tap) { return HEX ? (tap < 3 ? tap - 1 : ((tap - 3) & 1) - 1 + (cx & 1)) : tap / 3 - 1; };
  auto tap_dx = [&](int tap) { return HEX ? (tap < 3 ? 0 : (tap < 5 ? -1 : 1)) : tap % 3 - 1; };
  uint32_t vmask = 0;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int y = cy + tap_dy(tap), x = cx + tap_dx(tap);
    if (tap < ntaps && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.wd) vmask |= 1u << tap;
  }
  const int total = __popc(vmask) * kqt;

  // staging roles: 8 lanes cover the 128 bytes (32 channels) of one row, so a load instruction touches 8 full cache
  // lines; thread t fetches 16 bytes (4 channels) of positions (t >> 3) + 32 j, j = 0..7, and 6 x 16 bytes of weights
  const int my_chunk = tid & 7;
  int my_row[8];
  bool my_live[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ppos = (tid >> 3) + 32 * j, group = gbase + (ppos >> 4);
    my_live[j] = group < n_groups;
    my_row[j] = (group * p.hw + cell) * 16 + (ppos & 15);
  }
#if NZ_WIDE_OVERLAP == 2
  // byte offsets of this thread's eight row slices within either source (a dead row -- a position group past the batch --
  // reads the tile's first group instead of branching round its load: its outputs are never stored, and a row of the MFMA
  // only sees its own operand row)
  uint32_t rowoff0[8], rowoff1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ppos = (tid >> 3) + 32 * j;
    const int row = ((my_live[j] ? gbase + (ppos >> 4) : gbase) * p.hw + cell) * 16 + (ppos & 15);
    rowoff0[j] = ((uint32_t)row * (uint32_t)p.s0 + (uint32_t)my_chunk * 4u) * 4u;
    rowoff1[j] = ((uint32_t)row * (uint32_t)p.s1 + (uint32_t)my_chunk * 4u) * 4u;
  }
#endif
  const u32x4* wtile = reinterpret_cast<const u32x4*>(p.ws) + (size_t)ct * (ntaps + 1) * kqt * (3 * WIDE_NT * 64);
  uint32_t taps_left = vmask;
  int tap = __ffs(taps_left) - 1, kq = 0, fs = 0;      // the cursor: step fs = (kq, tap); the centre tap is always on the board
  taps_left &= taps_left - 1;
  // fetched operands wait in registers: two sets, so that a step's loads are issued TWO steps ahead (with one wavefront per
  // SIMD a load that is waited for is time nothing else fills: a quarter of the L2 requests miss)
  f32x4 ra2[2][8];
  u32x4 rb2[2][6];
  typedef std::integral_constant<int, 0> Set0;
  typedef std::integral_constant<int, 1> Set1;
  auto fetch = [&](auto set) {                         // global -> registers for the cursor's step, then advance
    f32x4 (&ra)[8] = ra2[decltype(set)::value];
    u32x4 (&rb)[6] = rb2[decltype(set)::value];
    const int shift = (tap_dy(tap) * p.wd + tap_dx(tap)) * 16;
    const bool second = kq >= kq0;
    const float* src = (second ? p.src1 : p.src0) + (second ? kq - kq0 : kq) * 32 + my_chunk * 4;
    const int cs = second ? p.s1 : p.s0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#ifdef NZ_ABLATE_WIDE_FETCH        // timing experiment: no activation loads (results wrong)
      ra[j] = f32x4{(float)j, 1.f, 2.f, (float)shift};
#else
      // (a branch per load; reading a live row instead and zeroing at the staging measured slower: 14.9 k against 15.4 k)
      ra[j] = my_live[j] ? *reinterpret_cast<const f32x4*>(src + (size_t)(my_row[j] + shift) * cs) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    const u32x4* wsrc = wtile + ((size_t)tap * kqt + kq) * (3 * WIDE_NT * 64);
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[j] = wsrc[tid + j * 256];
#if NZ_WIDE_KQ_OUTER       // K order: 32-channel group by group, each group's taps back to back (see above)
    if (fs + 1 < total) {    // (the cursor never leaves the last step: later fetches repeat it)
      ++fs;
      if (!taps_left) { taps_left = vmask; ++kq; }
      tap = __ffs(taps_left) - 1;
      taps_left &= taps_left - 1;
    }
#else
    if (++kq == kqt) {
      kq = 0;
      if (taps_left) { tap = __ffs(taps_left) - 1; taps_left &= taps_left - 1; }
      else tap = ntaps;
    }
#endif
  };
  // registers -> LDS
  auto stage_a = [&](int buf, int j) {                 // row slice j: 4 channels -> 8 bytes per piece (activations split here, once)
    f32x4 (&ra)[8] = ra2[0];
    unsigned char* ab = reinterpret_cast<unsigned char*>(&sA[buf][0][0][0]);
    const int ppos = (tid >> 3) + 32 * j;
    const int off = ppos * 64 + (((my_chunk >> 1) ^ ((ppos >> 2) & 3)) << 4) + (my_chunk & 1) * 8;
    const float x0 = ra[j][0], x1 = ra[j][1], x2 = ra[j][2], x3 = ra[j][3];
#ifdef NZ_ABLATE_WIDE_SPLIT        // timing experiment: no split arithmetic (results wrong)
    const float r0 = x1, r1 = x0, r2 = x3, r3 = x2;
#else
    const float r0 = x0 - wide_trunc(x0), r1 = x1 - wide_trunc(x1), r2 = x2 - wide_trunc(x2), r3 = x3 - wide_trunc(x3);
#endif
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(ab + 0 * (WIDE_GROUPS * 16 * 64) + off) = u32x2{wide_pack_hi16(x0, x1), wide_pack_hi16(x2, x3)};
    *reinterpret_cast<u32x2*>(ab + 1 * (WIDE_GROUPS * 16 * 64) + off) = u32x2{wide_pack_hi16(r0, r1), wide_pack_hi16(r2, r3)};
#ifdef NZ_ABLATE_WIDE_SPLIT
    *reinterpret_cast<u32x2*>(ab + 2 * (WIDE_GROUPS * 16 * 64) + off) = u32x2{wide_pack_hi16(x0, r1), wide_pack_hi16(r2, x3)};
#else
    *reinterpret_cast<u32x2*>(ab + 2 * (WIDE_GROUPS * 16 * 64) + off) =
        u32x2{wide_pack_hi16(r0 - wide_trunc(r0), r1 - wide_trunc(r1)), wide_pack_hi16(r2 - wide_trunc(r2), r3 - wide_trunc(r3))};
#endif
  };
  auto stage_b = [&](int buf) {
    u32x4 (&rb)[6] = rb2[0];
    u32x4* fb = &sB[buf][0][0][0];
#pragma unroll
    for (int j = 0; j < 6; ++j) fb[tid + j * 256] = rb[j];
  };

  // compute roles: wavefront = (half of the positions, half of the channels)
  const int ph = wave & 1, ch = wave >> 1;
  const int pos = lane & 15, quad = lane >> 4;
  f32x4 acc[8][4];
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef NZ_WIDE_STAMPS
  const bool stamping = blockIdx.x == 0 && blockIdx.y == 0 && tid < 64;
  unsigned long long wtk[5] = {0, 0, 0, 0, 0}, wts = __builtin_amdgcn_s_memtime();
#endif
  fetch(Set0{});
#pragma unroll
  for (int j = 0; j < 8; ++j) stage_a(0, j);
  stage_b(0);
#if NZ_WIDE_OVERLAP == 2
  if (total > 1) fetch(Set1{});
#endif
  __syncthreads();
#if NZ_WIDE_OVERLAP == 2
  // step s: MFMAs on LDS buffer s & 1; the registers of set (s + 1) & 1 (step s + 1's operands, fetched a step ago) are
  // staged into the other LDS buffer in the MFMAs' gaps; step s + 2's operands are fetched into set s & 1 (staged during
  // step s - 1).  Two copies of the step's code, one per register set.
  auto step2 = [&](int s, auto set) {
    constexpr int par = decltype(set)::value;
    const int buf = par;
    WIDE_STAMP(4);
    // step s + 2's operands: 14 loads of 1 KB per wavefront -- 57 KB per step through the CU's 64-byte-a-cycle vector
    // memory path, 900 cycles a wavefront issuing them in one burst stands still for (measured: 1,124) -- go out one by
    // one between the MFMAs: six in the first position group, eight in the last three (the staging fills the gaps of the
    // groups between).  Behind the last steps the cursor stays where it is (the loads repeat).
    const bool ahead = fs + 1 < total;
    const bool second = kq >= kq0;
    const char* const fsrc = reinterpret_cast<const char*>((second ? p.src1 : p.src0) + (second ? kq - kq0 : kq) * 32) +
                             (ptrdiff_t)(tap_dy(tap) * p.wd + tap_dx(tap)) * 16 * (second ? p.s1 : p.s0) * 4;
    const u32x4* const fw = wtile + ((size_t)tap * kqt + kq) * (3 * WIDE_NT * 64);
    f32x4 (&fa)[8] = ra2[par];
    u32x4 (&fb)[6] = rb2[par];
    auto fetch_granule = [&](int i) {
      if (i < 8) {
#ifdef NZ_ABLATE_WIDE_FETCH
        fa[i] = f32x4{(float)i, 1.f, 2.f, (float)second};
#else
        fa[i] = *reinterpret_cast<const f32x4*>(fsrc + (second ? rowoff1[i] : rowoff0[i]));
#endif
      } else if (i < 14) fb[i - 8] = fw[tid + (i - 8) * 256];
    };
    {   // the cursor moves on (scalar selects: no branch in the step)
      const bool wrap = taps_left == 0;
      const uint32_t tl = wrap ? vmask : taps_left;
      const int ntap = __ffs(tl) - 1;
      kq = ahead ? kq + (wrap ? 1 : 0) : kq;
      tap = ahead ? ntap : tap;
      taps_left = ahead ? (tl & (tl - 1)) : taps_left;
      fs += ahead ? 1 : 0;
    }
    WIDE_STAMP(0);                   // loads' addresses
    f32x4 (&ra)[8] = ra2[par ^ 1];
    u32x4 (&rb)[6] = rb2[par ^ 1];
    u32x4 wf[4][3];
    u32x4 xf[2][3];
    auto load_x = [&](int g, int slot_buf) {
      const int prow = (ph * 8 + g) * 16 + pos;
      const int slot = quad ^ ((prow >> 2) & 3);
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) xf[slot_buf][piece] = sA[buf][piece][prow][slot];
    };
    load_x(0, 0);                    // (the first chain's operands first: the MFMAs start when they are there)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) wf[n][piece] = sB[buf][piece][ch * 4 + n][lane];
#ifdef NZ_WIDE_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WIDE_STAMP(1);                   // first fragments there
    // The staging of step s + 1 (split arithmetic, LDS stores) rides in the gaps of this step's MFMAs: an MFMA's 16-cycle
    // gap hides eight cycles of other vector issue and no more, and with one wavefront per SIMD nothing else fills the
    // time -- so it is cut into 94 granules of at most two vector instructions and a store (per 4-channel row slice: piece
    // 0 packed and stored, the four first residuals one by one, piece 1, the four second residuals, piece 2; then the
    // weights' six stores), one behind each MFMA from the second position group on, a scheduling barrier after each.
    // Unconditional, so that it shares the MFMAs' basic block: behind the last step it stages stale registers nobody reads.
    float sr[4] = {0.f, 0.f, 0.f, 0.f};
    int soff = 0;
    auto stage_granule = [&](int q) {
      unsigned char* ab = reinterpret_cast<unsigned char*>(&sA[buf ^ 1][0][0][0]);
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      if (q >= 94) return;
      if (q >= 88) { (&sB[buf ^ 1][0][0][0])[tid + (q - 88) * 256] = rb[q - 88]; return; }
      const int j = q / 11, t = q % 11;
      if (t == 0) {
        const int ppos = (tid >> 3) + 32 * j;
        soff = ppos * 64 + (((my_chunk >> 1) ^ ((ppos >> 2) & 3)) << 4) + (my_chunk & 1) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = ra[j][i];
      }
      if (t == 0 || t == 5 || t == 10)
        *reinterpret_cast<u32x2*>(ab + (t / 5) * (WIDE_GROUPS * 16 * 64) + soff) = u32x2{wide_pack_hi16(sr[0], sr[1]), wide_pack_hi16(sr[2], sr[3])};
      else {
        const int i = (t - 1) % 5;
        sr[i] = sr[i] - wide_trunc(sr[i]);
      }
    };
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g + 1 < 8) load_x(g + 1, (g + 1) & 1);
      const u32x4 x0 = xf[g & 1][0], x1 = xf[g & 1][1], x2 = xf[g & 1][2];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#define NZ_WIDE_STEP(B, X, T)                                                  \
  acc[g][n] = WIDE_MFMA(wf[n][B], X, acc[g][n]);                               \
  {                                                                            \
    /* the 94 staging granules spread evenly over the 168 gaps behind MFMAs 24..191, the 14 loads over the 74 gaps they leave */ \
    constexpr int SG = 94, FG = 14, GAPS = 168, FREE = GAPS - SG;              \
    const int m = g * 24 + n * 6 + T - 24;                                     \
    if (m >= 0) {                                                              \
      const int q = m * SG / GAPS;                                             \
      if (m == 0 || q != (m - 1) * SG / GAPS) stage_granule(q);                \
      else {                                                                   \
        const int r = m - ((m - 1) * SG / GAPS + 1), f = (r * FG + FREE - 1) / FREE; \
        if (f < FG && f * FREE / FG == r) fetch_granule(f);                    \
      }                                                                        \
    }                                                                          \
  }                                                                            \
  __builtin_amdgcn_sched_barrier(0);
        NZ_WIDE_STEP(1, x1, 0) NZ_WIDE_STEP(0, x2, 1) NZ_WIDE_STEP(2, x0, 2) NZ_WIDE_STEP(0, x1, 3) NZ_WIDE_STEP(1, x0, 4) NZ_WIDE_STEP(0, x0, 5)
#undef NZ_WIDE_STEP
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WIDE_STAMP(2);                   // MFMAs and staging issued
    __syncthreads();
    WIDE_STAMP(3);                   // barrier
  };
  for (int s = 0; s < total; s += 2) {
    step2(s, Set0{});
    if (s + 1 < total) step2(s + 1, Set1{});
  }
#else
  for (int s = 0; s < total; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < total;
    if (more) fetch(Set0{});                           // the next step's operands fly under this step's MFMAs
    f32x4 (&ra)[8] = ra2[0];
    u32x4 (&rb)[6] = rb2[0];
    u32x4 wf[4][3];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) wf[n][piece] = sB[buf][piece][ch * 4 + n][lane];
    u32x4 xf[2][3];
    auto load_x = [&](int g, int slot_buf) {
      const int prow = (ph * 8 + g) * 16 + pos;
      const int slot = quad ^ ((prow >> 2) & 3);
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) xf[slot_buf][piece] = sA[buf][piece][prow][slot];
    };
    load_x(0, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g + 1 < 8) load_x(g + 1, (g + 1) & 1);             // the next group's fragments fly under this group's MFMAs
      const u32x4 x0 = xf[g & 1][0], x1 = xf[g & 1][1], x2 = xf[g & 1][2];
#if NZ_WIDE_INTERLEAVE
      // small terms first; the four column tiles' chains side by side (a term of each in turn: an MFMA that waits for the
      // one before it issues late)
#define NZ_WIDE_TERM(B, X)                                                          \
  _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[g][n] = wide_mfma(wf[n][B], X, acc[g][n]); \
  __builtin_amdgcn_sched_barrier(0);
      NZ_WIDE_TERM(1, x1) NZ_WIDE_TERM(0, x2) NZ_WIDE_TERM(2, x0) NZ_WIDE_TERM(0, x1) NZ_WIDE_TERM(1, x0) NZ_WIDE_TERM(0, x0)
#undef NZ_WIDE_TERM
#else
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        acc[g][n] = wide_mfma(wf[n][1], x1, acc[g][n]);       // small terms first; one dependent chain
        acc[g][n] = wide_mfma(wf[n][0], x2, acc[g][n]);
        acc[g][n] = wide_mfma(wf[n][2], x0, acc[g][n]);
        acc[g][n] = wide_mfma(wf[n][0], x1, acc[g][n]);
        acc[g][n] = wide_mfma(wf[n][1], x0, acc[g][n]);
        acc[g][n] = wide_mfma(wf[n][0], x0, acc[g][n]);
        __builtin_amdgcn_sched_barrier(0);                     // keep the chain together (net_dev.hpp)
      }
#endif
    }
    if (more) {                                                // staging after the MFMAs: slices between the chains were slower
#pragma unroll
      for (int j = 0; j < 8; ++j) stage_a(buf ^ 1, j);
      stage_b(buf ^ 1);
    }
    __syncthreads();
  }
#endif

#ifdef NZ_WIDE_STAMPS
  if (stamping && lane == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_wide_stamps[i], wtk[i]);
    atomicAdd(&g_wide_stamps[5], (unsigned long long)total);
    atomicAdd(&g_wide_stamps[6], 1ull);
  }
#endif
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int group = gbase + ph * 8 + g;
    if (group >= n_groups) continue;
    const size_t orow = (size_t)(group * p.hw + cell) * 16 + pos;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const size_t o = orow * p.cd + ct * 128 + (ch * 4 + n) * 16 + quad * 4;
      f32x4 v = acc[g][n];
      if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
      *reinterpret_cast<f32x4*>(p.dst + o) = v;
    }
  }
}

__device__ __forceinline__ size_t row_of(size_t n, int cell, int hw) { return ((n >> 4) * hw + cell) * 16 + (n & 15); }

// [n][C][H*W] (what Game.generate_network_input stacks) -> rows of cp channels, zero-padded in the
// channels and in the positions that fill up the last group of 16.
__global__ void nchw_to_rows_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* n_dev,
                                    int n_host, int c, int cp, int hw) {
  const int n_pos = n_dev ? *n_dev : n_host;
  const size_t n_pad = (size_t)((n_pos + 15) & ~15);
  const size_t total = n_pad * hw * cp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cp);
    const size_t r = i / cp;                       // output row
    const size_t n = (r >> 4) / hw * 16 + (r & 15);
    const int cell = (int)((r >> 4) % hw);
    out[i] = (ch < c && n < (size_t)n_pos) ? in[(n * c + ch) * hw + cell] : 0.f;
  }
}

// One wavefront per position: logits in the reference's action order (plane-major: action =
// plane * H*W + cell), softmax over ALL of them (Explorer.py:159), value = tanh(mean)
// (blocks.py:82-84).
__global__ __launch_bounds__(64) void finalize_kernel(const float* __restrict__ pol, int pp, int planes,
                                                      const float* __restrict__ val, int vp, int hw,
                                                      const int32_t* n_dev, int n_host, float* __restrict__ logits,
                                                      float* __restrict__ probs, float* __restrict__ value) {
  const int n_pos = n_dev ? *n_dev : n_host;
  const int n = blockIdx.x;
  if (n >= n_pos) return;
  const int lane = threadIdx.x;
  const int A = planes * hw;
  float mx = -INFINITY;
  for (int a = lane; a < A; a += 64) {
    const float v = pol[row_of(n, a % hw, hw) * pp + a / hw];
    if (logits) logits[(size_t)n * A + a] = v;
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (probs) {
    float sum = 0.f;
    for (int a = lane; a < A; a += 64) {
      const float e = expf(pol[row_of(n, a % hw, hw) * pp + a / hw] - mx);
      probs[(size_t)n * A + a] = e;
      sum += e;
    }
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    for (int a = lane; a < A; a += 64) probs[(size_t)n * A + a] /= sum;
  }
  float s = 0.f;
  for (int c = lane; c < hw; c += 64) s += val[row_of(n, c, hw) * vp];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) value[n] = tanhf(s / (float)hw);
}


// ---- the whole network in ONE launch (narrow nets on small boards) ---------------------------------------------
// A simulation wave of SCS self-play evaluates a few hundred leaves with a 32-filter network: one launch per layer
// is 15 launches of ~10 us with the chip mostly idle.  Here one workgroup runs ALL layers for its share of the
// positions (P = ceil(leaves / workgroups) of them; the leaf count is read from device memory), activations in LDS:
//   * rows are (position, cell) pairs, row = local position * H*W + cell; an MFMA tile is 16 consecutive rows, so a
//     conv tap is a per-lane row shift (same position, neighbouring cell) with a per-lane on-board test -- off-board
//     taps contribute zero operands (the per-layer kernel skips them tile-wide; adding a zero product is exact, so the
//     two kernels give the same floats);
//   * same arithmetic and the same packed weights as conv_kernel: v_mfma_f32_16x16x4_f32, K order = tap, channel group;
//   * the (output row tile, 16-channel column tile) jobs of a layer are dealt to the sixteen wavefronts, one barrier
//     per layer; the weight stream comes from L2 (every workgroup reads the same ~0.5 MB);
//   * the softmax over all logits and the value's mean + tanh (finalize_kernel) run at the end, one wavefront per
//     position, from LDS.
constexpr int FUSED_BUFFERS = 8;
constexpr int FUSED_WAVES = 16;             // wavefronts per workgroup: a layer's jobs run side by side, four per SIMD
constexpr int FUSED_THREADS = FUSED_WAVES * 64;
constexpr int FUSED_PAD = 4;               // floats added to every LDS row: spreads the 16 rows of a tile over the banks
struct FusedOp {                           // one conv layer, every LDS address resolved by the host (float offsets)
  const float* w;                          // packed weights (PackedConv::dev)
  int32_t off0, cs0, off1, cs1;            // sources: offset and floats per row (off1 = -1: one source)
  int32_t offd, csd, offr, csr;            // destination; residual (offr = -1: none)
  int32_t kg0, kg1, ntiles, act;           // 16-channel groups of each source, 16-channel output tiles, activation
  int32_t w_lds, w_chunks;                 // 1: weights staged in LDS; 16-byte pieces of one column tile (taps x groups x 64)
  int32_t w_slot, w_after_barrier;         // which weight buffer; 1: it is the buffer the previous layer reads (store after its barrier)
};
constexpr int FUSED_OP_WORDS = sizeof(FusedOp) / 4;
constexpr int FUSED_LDS_OPS = 48;          // descriptors kept in LDS (more layers: read from memory per layer)
struct FusedProgram {
  int32_t n_ops, hw, h, wd, planes, hex;
  int32_t zrow_off, ops_off;               // float offsets of the zero row and of the descriptors' copy (-1: none) in LDS
  int32_t wbuf_off[2];                     // the two weight buffers (the second may be the input rows' space, or the first again)
  int32_t in_off, in_cs, pol_off, pol_cs, val_off, val_cs;     // input rows, policy logits, value plane
  FusedOp ops[FUSED_MAX_OPS];
};


constexpr int FUSED_ZROW = 256;            // floats of zeros in LDS: the operand row of a tap that falls off the board
constexpr int FUSED_WREGS = 3;             // 16-byte pieces of the next layer's weights a thread carries
typedef const __attribute__((address_space(1))) f32x4* gptr4;

// A layer's weights [col tile][tap < NTAPS][kg][lane][4] (what conv_job reads from LDS) out of the packed stream
// [col tile][NTAPS + 1][kg][lane][4]: piece i of 16 bytes, i = tid + 1024 j.
template <int NTAPS>
__device__ __forceinline__ void fetch_weights(const FusedOp& op, f32x4 (&wreg)[FUSED_WREGS], int tid) {
  const int kgt = op.kg0 + op.kg1, per_tile = op.w_chunks, total = op.ntiles * per_tile;
#pragma unroll
  for (int j = 0; j < FUSED_WREGS; ++j) {
    const int i = tid + j * FUSED_THREADS;
    if (i < total) {
      int ct = 0, within = i;                  // (at most a handful of column tiles: no division)
      while (within >= per_tile) { within -= per_tile; ++ct; }
      wreg[j] = *((gptr4)op.w + (size_t)ct * ((NTAPS + 1) * kgt * 64) + within);
    }
  }
}
template <int NTAPS>
__device__ __forceinline__ void store_weights(const FusedOp& op, float* wbuf, const f32x4 (&wreg)[FUSED_WREGS], int tid) {
  const int total = op.ntiles * op.w_chunks;
#pragma unroll
  for (int j = 0; j < FUSED_WREGS; ++j) {
    const int i = tid + j * FUSED_THREADS;
    if (i < total) *reinterpret_cast<f32x4*>(wbuf + (size_t)i * 4) = wreg[j];
  }
}
template <int NTAPS>
__device__ __forceinline__ void stage_weights(const FusedOp& op, float* wbuf, int tid, bool) {
  f32x4 wreg[FUSED_WREGS];
  fetch_weights<NTAPS>(op, wreg, tid);
  store_weights<NTAPS>(op, wbuf, wreg, tid);
}

// One (row tile, column tile) job as straight-line code: NTAPS x KGT steps of four MFMAs, K order = tap, channel group
// (conv_kernel's).  Every index is a compile-time constant; the A operands come from LDS (aoff[tap]: this lane's
// operand row), the B operands from the LDS copy of the layer's weights (WLDS) or straight from the packed stream in L2.
template <int NTAPS, int KGT, bool WLDS>
__device__ __forceinline__ void conv_job(f32x4& acc, const float* __restrict__ lds, const int (&aoff)[NTAPS],
                                         const float* __restrict__ wl, const float* __restrict__ wg) {
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap) {
#pragma unroll
    for (int kg = 0; kg < KGT; ++kg) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + aoff[tap] + kg * 16);
      f32x4 b;
      if constexpr (WLDS) b = *reinterpret_cast<const f32x4*>(wl + (tap * KGT + kg) * 256);
      else b = *((gptr4)(wg + (tap * KGT + kg) * 256));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
  }
}
// any channel-group count, two K sources (the recall concatenation): taps unrolled, channel groups a run-time loop
template <int NTAPS>
__device__ __forceinline__ void conv_job_generic(f32x4& acc, const float* __restrict__ lds, const int (&aoff0)[NTAPS],
                                                 const int (&aoff1)[NTAPS], int kg0, int kgt, const float* w) {
  // `w`: LDS copy ([tap][kg] contiguous) or the packed stream (same order within a column tile)
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap) {
    for (int kg = 0; kg < kgt; ++kg) {
      const f32x4 a = kg < kg0 ? *reinterpret_cast<const f32x4*>(lds + aoff0[tap] + kg * 16)
                               : *reinterpret_cast<const f32x4*>(lds + aoff1[tap] + (kg - kg0) * 16);
      const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)(tap * kgt + kg) * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
  }
}

template <bool HEX>
__global__ __launch_bounds__(FUSED_THREADS) void fused_net_kernel(const FusedProgram* __restrict__ prog, const float* __restrict__ in_rows,
                                                        int in_channels, const int32_t* __restrict__ n_dev, int n_host,
                                                        float* __restrict__ logits, float* __restrict__ probs,
                                                        float* __restrict__ value) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_pos = n_dev ? *n_dev : n_host;
  const int P = (n_pos + (int)gridDim.x - 1) / (int)gridDim.x;
  const int p0 = blockIdx.x * P;
  if (p0 >= n_pos) return;                                  // uniform per workgroup
  const int np = min(P, n_pos - p0);
  const int hw = prog->hw, H = prog->h, Wd = prog->wd, n_ops = prog->n_ops;
#ifdef NZ_FUSED_STAMPS     // diagnostic build: where one workgroup's time goes (printed by workgroup 0)
  unsigned long long tk_in = 0, tk_job[32], tk_bar[32], tk_fin = 0, ts = __builtin_amdgcn_s_memtime();
#define FSTAMP(x) { const unsigned long long now = __builtin_amdgcn_s_memtime(); x = now - ts; ts = now; }
#else
#define FSTAMP(x)
#endif
  const int rows = np * hw, row_tiles = (rows + 15) >> 4;
  constexpr int ntaps = HEX ? 7 : 9;
  const int q4 = (lane >> 4) * 4;
  const int zoff = prog->zrow_off;

  {   // this workgroup's input rows -> LDS buffer 0 (global row of (position n, cell c): ((n >> 4) * hw + c) * 16 + (n & 15))
    const int cs = prog->in_cs;
    float* dst = lds + prog->in_off;
    const int chunks = in_channels >> 2;                    // 16-byte pieces per row
    for (int i = tid; i < rows * chunks; i += FUSED_THREADS) {
      const int r = i / chunks, c4 = (i - r * chunks) << 2;
      const int pl = r / hw, cell = r - pl * hw, n = p0 + pl;
      const size_t grow = ((size_t)(n >> 4) * hw + cell) * 16 + (n & 15);
      *reinterpret_cast<f32x4*>(dst + r * cs + c4) = *reinterpret_cast<const f32x4*>(in_rows + grow * in_channels + c4);
    }
  }
  // the layer descriptors -> LDS (a scalar load chain per layer otherwise), the row of zeros an off-board tap reads
  uint32_t* const sops = reinterpret_cast<uint32_t*>(lds + prog->ops_off);
  const bool ops_in_lds = prog->ops_off >= 0;
  if (ops_in_lds)
    for (int i = tid; i < n_ops * FUSED_OP_WORDS; i += FUSED_THREADS) sops[i] = reinterpret_cast<const uint32_t*>(prog->ops)[i];
  for (int i = tid; i < FUSED_ZROW; i += FUSED_THREADS) lds[zoff + i] = 0.f;
  auto load_op = [&](int o) {
    FusedOp op;
    if (ops_in_lds) {
      uint32_t w[FUSED_OP_WORDS];
#pragma unroll
      for (int i = 0; i < FUSED_OP_WORDS; ++i) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sops[o * FUSED_OP_WORDS + i]);
      __builtin_memcpy(&op, w, sizeof(op));
    } else {
      op = prog->ops[o];
    }
    return op;
  };
  if (n_ops > 0 && prog->ops[0].w_lds) stage_weights<ntaps>(prog->ops[0], lds + prog->wbuf_off[prog->ops[0].w_slot], tid, true);
  __syncthreads();
  FSTAMP(tk_in);

  // this lane's operand rows per tap, relative to a buffer: recomputed only when the wave's row tile changes
  int rt_cached = -1, srow[ntaps];
  uint32_t on_mask = 0;
  FusedOp next = n_ops > 0 ? load_op(0) : FusedOp{};
  for (int o = 0; o < n_ops; ++o) {
    const FusedOp op = next;
    if (o + 1 < n_ops) next = load_op(o + 1);
    // the NEXT layer's weights start their way from L2 now and are parked in registers under this layer's MFMAs
    const bool next_lds = o + 1 < n_ops && next.w_lds;
    f32x4 wreg[FUSED_WREGS];
    if (next_lds) fetch_weights<ntaps>(next, wreg, tid);
    float* dst = lds + op.offd;
    const float* res = op.offr >= 0 ? lds + op.offr : nullptr;
    const int kg0 = op.kg0, kgt = op.kg0 + op.kg1;
    const size_t tile_stride = (size_t)(ntaps + 1) * kgt * 256;
    const int n_jobs = row_tiles * op.ntiles;
    const float* wbuf = lds + prog->wbuf_off[op.w_slot];
    for (int job = wave; job < n_jobs; job += FUSED_WAVES) {
      const int rt = job / op.ntiles, ct = job - rt * op.ntiles;
      if (rt != rt_cached) {
        rt_cached = rt;
        const int row = rt * 16 + (lane & 15);
        const bool row_ok = row < rows;
        const int pl = row / hw, cell = row - pl * hw;
        const int cy = cell / Wd, cx = cell - cy * Wd;
        on_mask = 0;
#pragma unroll
        for (int tap = 0; tap < ntaps; ++tap) {
          const int dy = HEX ? (tap < 3 ? tap - 1 : ((tap - 3) & 1) - 1 + (cx & 1)) : tap / 3 - 1;
          const int dx = HEX ? (tap < 3 ? 0 : (tap < 5 ? -1 : 1)) : tap % 3 - 1;
          if (row_ok && (unsigned)(cy + dy) < (unsigned)H && (unsigned)(cx + dx) < (unsigned)Wd) on_mask |= 1u << tap;
          srow[tap] = row + dy * Wd + dx;
        }
      }
      // per tap: where this lane's operand row starts in LDS (float index); off the board -> the row of zeros
      int aoff0[ntaps], aoff1[ntaps];
#pragma unroll
      for (int tap = 0; tap < ntaps; ++tap) {
        const bool on = (on_mask >> tap) & 1u;
        aoff0[tap] = on ? op.off0 + srow[tap] * op.cs0 + q4 : zoff + q4;
        aoff1[tap] = on && op.off1 >= 0 ? op.off1 + srow[tap] * op.cs1 + q4 : zoff + q4;
      }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* wl = wbuf + (size_t)ct * (ntaps * kgt * 256) + lane * 4;        // LDS copy: [ct][tap][kg][lane][4]
      const float* wg = op.w + (size_t)ct * tile_stride + lane * 4;                // packed stream in L2
      if (op.off1 < 0 && op.w_lds && kgt == 2) conv_job<ntaps, 2, true>(acc, lds, aoff0, wl, wg);
      else if (op.off1 < 0 && op.w_lds && kgt == 1) conv_job<ntaps, 1, true>(acc, lds, aoff0, wl, wg);
      else if (op.off1 < 0 && op.w_lds && kgt == 3) conv_job<ntaps, 3, true>(acc, lds, aoff0, wl, wg);
      else if (op.off1 < 0 && op.w_lds && kgt == 4) conv_job<ntaps, 4, true>(acc, lds, aoff0, wl, wg);
      else if (op.off1 < 0 && kgt == 6) conv_job<ntaps, 6, false>(acc, lds, aoff0, wl, wg);
      else if (op.off1 < 0 && kgt == 2) conv_job<ntaps, 2, false>(acc, lds, aoff0, wl, wg);
      else conv_job_generic<ntaps>(acc, lds, aoff0, aoff1, kg0, kgt, op.w_lds ? wl : wg);
      const int col = ct * 16 + (lane & 15), r4 = (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = rt * 16 + r4 + r;
        if (orow < rows) {
          float v = acc[r];
          if (res) v += res[orow * op.csr + col];
          dst[orow * op.csd + col] = activate(v, op.act);
        }
      }
    }
#ifdef NZ_FUSED_STAMPS
    if (o < 32) FSTAMP(tk_job[o]);
#endif
    // The next layer's weights go to LDS: into the OTHER weight buffer right away when there is one (it was last
    // read a layer ago), otherwise into the same one once every wave is done with it.  One barrier ends the layer.
    if (next_lds && !next.w_after_barrier) store_weights<ntaps>(next, lds + prog->wbuf_off[next.w_slot], wreg, tid);
    __syncthreads();
    if (next_lds && next.w_after_barrier) {
      store_weights<ntaps>(next, lds + prog->wbuf_off[next.w_slot], wreg, tid);
      __syncthreads();
    }
#ifdef NZ_FUSED_STAMPS
    if (o < 32) FSTAMP(tk_bar[o]);
#endif
  }

  // softmax over ALL logits (Explorer.py:159), value = tanh(mean) (blocks.py:82-84): finalize_kernel's arithmetic in
  // finalize_kernel's order (lane i sums actions i, i + 64, ...), one wavefront per position; exp() is kept in LDS
  // between the passes so that the probabilities are written once
  float* pol = lds + prog->pol_off;
  const int pp = prog->pol_cs;
  const float* val = lds + prog->val_off;
  const int vp = prog->val_cs;
  const int A = prog->planes * hw;
  const int cell0 = lane % hw, plane0 = lane / hw, dcell = 64 % hw, dplane = 64 / hw;
  for (int pl = wave; pl < np; pl += FUSED_WAVES) {
    const size_t n = (size_t)(p0 + pl);
    float* prow = pol + pl * hw * pp;
    float mx = -INFINITY;
    for (int i = lane, cell = cell0, plane = plane0; i < A; i += 64) {
      const float v = prow[cell * pp + plane];
      if (logits) logits[n * A + i] = v;
      mx = fmaxf(mx, v);
      cell += dcell; plane += dplane;
      if (cell >= hw) { cell -= hw; ++plane; }
    }
    for (int w = 32; w; w >>= 1) mx = fmaxf(mx, __shfl_xor(mx, w));
    if (probs) {
      float sum = 0.f;
      for (int i = lane, cell = cell0, plane = plane0; i < A; i += 64) {
        const float e = expf(prow[cell * pp + plane] - mx);
        prow[cell * pp + plane] = e;
        sum += e;
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
      for (int w = 32; w; w >>= 1) sum += __shfl_xor(sum, w);
      for (int i = lane, cell = cell0, plane = plane0; i < A; i += 64) {
        probs[n * A + i] = prow[cell * pp + plane] / sum;
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
    }
    float sv = 0.f;
    for (int c = lane; c < hw; c += 64) sv += val[(pl * hw + c) * vp];
    for (int w = 32; w; w >>= 1) sv += __shfl_xor(sv, w);
    if (lane == 0) value[n] = tanhf(sv / (float)hw);
  }
#ifdef NZ_FUSED_STAMPS
  FSTAMP(tk_fin);
  if (blockIdx.x == 0 && (tid == 0 || tid == FUSED_THREADS - 64)) {
    printf("fused wg0 wave %d np %d rows %d: input %llu finalize %llu\n", wave, np, rows, tk_in, tk_fin);
    for (int o = 0; o < n_ops && o < 32; ++o) printf("  wave %d op %d kgt %d ntiles %d wlds %d slot %d after %d: jobs %llu barrier+stage %llu\n", wave, o, prog->ops[o].kg0 + prog->ops[o].kg1, prog->ops[o].ntiles, prog->ops[o].w_lds, prog->ops[o].w_slot, prog->ops[o].w_after_barrier, tk_job[o], tk_bar[o]);
  }
#endif
}

// ---- the one-launch network on the BF16 matrix cores (fused16_net_kernel) -------------------------------------------
// fused_net_kernel with the arithmetic of the Tic-Tac-Toe network (net_dev.hpp): every float32 is the exact sum of
// three bf16 pieces, a product is the six piece products of weight >= 2^-16 on v_mfma_f32_16x16x32_bf16 with float32
// accumulation (96 matrix-pipe cycles per 32 channels where eight v_mfma_f32_16x16x4_f32 take 256).  Activations live
// in LDS as pieces, [piece][row][channel] bf16, split ONCE by the layer that makes them; a row's 16-byte chunks (eight
// channels) are stored at chunk ^ ((row >> 2) & 3), so that the sixteen rows of a tile -- 64-byte rows for 32 channels --
// spread over all banks without padding.  The logits and the value plane stay float32 rows for the softmax.
constexpr int FUSED16_WREGS = 4;           // 16-byte chunks of the next layer's weights a thread carries (64 KB per layer)
typedef const __attribute__((address_space(1))) u32x4* gptr4u;

__device__ __forceinline__ void fetch_weights16(const Fused16Op& op, u32x4 (&wreg)[FUSED16_WREGS], int tid) {
  const int total = op.ntiles * op.w_chunks;
#pragma unroll
  for (int j = 0; j < FUSED16_WREGS; ++j) {
    const int i = tid + j * FUSED_THREADS;
    if (i < total) wreg[j] = *((gptr4u)op.w + i);
  }
}
__device__ __forceinline__ void fetch_weights16_chunk(const Fused16Op& op, u32x4 (&wreg)[FUSED16_WREGS], int tid, int j) {
  const int i = tid + j * FUSED_THREADS;
  if (i < op.ntiles * op.w_chunks) wreg[j] = *((gptr4u)op.w + i);
}
__device__ __forceinline__ void store_weights16(const Fused16Op& op, float* wbuf, const u32x4 (&wreg)[FUSED16_WREGS], int tid) {
  const int total = op.ntiles * op.w_chunks;
#pragma unroll
  for (int j = 0; j < FUSED16_WREGS; ++j) {
    const int i = tid + j * FUSED_THREADS;
    if (i < total) *reinterpret_cast<u32x4*>(wbuf + (size_t)i * 4) = wreg[j];
  }
}
// One (row tile, column tile) job as straight-line code: NTAPS x KGT steps.  `srow[tap]`: this lane's operand row of the
// tap -- an off-board tap reads the buffer's row of zeros (row index = the workgroup's row count, never written).
// `fetch`: this job also issues the loads of the NEXT layer's weights, one 16-byte chunk per thread every STEPS / 4
// steps -- sixteen wavefronts issuing all four at the layer's start queue up at the texture addresser for 1.7 k cycles
// before the first MFMA.
template <int NTAPS, int KGT, bool WLDS>
__device__ __forceinline__ void conv16_job(f32x4& acc, const float* __restrict__ lds, const int (&srow)[NTAPS], int off0,
                                           int cs0, int ps0, int kq, const float* __restrict__ wl,
                                           const uint32_t* __restrict__ wg, bool fetch, const Fused16Op& next,
                                           u32x4 (&wreg)[FUSED16_WREGS], int tid) {
  constexpr int STEPS = NTAPS * KGT;
  static_assert(STEPS >= FUSED16_WREGS, "a chunk per STEPS / FUSED16_WREGS steps");
  constexpr int AHEAD = WLDS ? 0 : 3;         // weights read from L2: their loads run three steps ahead of the MFMAs
  u32x4 bq[AHEAD + 1][3];
  if constexpr (!WLDS) {
#pragma unroll
    for (int st = 0; st < AHEAD && st < STEPS; ++st)
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) bq[st][piece] = *((gptr4u)(wg + (st * 3 + piece) * 256));
  }
#pragma unroll
  for (int tap = 0; tap < NTAPS; ++tap) {
    const int rbase = off0 + srow[tap] * cs0, sw = (srow[tap] >> 2) & 3;
#pragma unroll
    for (int kg = 0; kg < KGT; ++kg) {
      const int st = tap * KGT + kg;
      if constexpr (WLDS) {                     // (the layers that read weights from L2 have no registers to spare)
        if (st % (STEPS / FUSED16_WREGS) == 0 && st / (STEPS / FUSED16_WREGS) < FUSED16_WREGS && fetch)
          fetch_weights16_chunk(next, wreg, tid, st / (STEPS / FUSED16_WREGS));
      }
      const int a0 = rbase + (((kg * 4 + kq) ^ sw) << 2);
      u32x4 a[3];
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) a[piece] = *reinterpret_cast<const u32x4*>(lds + a0 + piece * ps0);
      if constexpr (WLDS) {
        u32x4 b[3];
#pragma unroll
        for (int piece = 0; piece < 3; ++piece) b[piece] = *reinterpret_cast<const u32x4*>(wl + (st * 3 + piece) * 256);
        step16(acc, a, b);
      } else {
        if (st + AHEAD < STEPS) {
#pragma unroll
          for (int piece = 0; piece < 3; ++piece)
            bq[(st + AHEAD) % (AHEAD + 1)][piece] = *((gptr4u)(wg + ((st + AHEAD) * 3 + piece) * 256));
        }
        step16(acc, a, bq[st % (AHEAD + 1)]);
      }
    }
  }
}
template <bool HEX>
__global__ __launch_bounds__(FUSED_THREADS) void fused16_net_kernel(const Fused16Program* __restrict__ prog,
                                                                    const float* __restrict__ in_rows, int in_channels,
                                                                    const int32_t* __restrict__ n_dev, int n_host,
                                                                    float* __restrict__ logits, float* __restrict__ probs,
                                                                    float* __restrict__ value) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The program's header and its first two layer descriptors come in three VECTOR loads issued together (lane i holds
  // dword i, v_readlane makes the scalars): as scalar loads they are a chain of dependent round trips to memory, a
  // few microseconds at the start of every launch.
  typedef const __attribute__((address_space(1))) uint32_t* gptr1u;
  constexpr int HDR_DWORDS = (int)(offsetof(Fused16Program, ops) / 4), OP_DWORDS = (int)(sizeof(Fused16Op) / 4);
  static_assert(HDR_DWORDS <= 64 && sizeof(Fused16Op) % 4 == 0 && OP_DWORDS <= 64, "one dword per lane");
  const gptr1u prog_words = (gptr1u)reinterpret_cast<const uint32_t*>(prog);
  const uint32_t hdr_v = lane < HDR_DWORDS ? prog_words[lane] : 0u;
  const uint32_t op0_v = lane < OP_DWORDS ? prog_words[HDR_DWORDS + lane] : 0u;
  const uint32_t op1_v = lane < OP_DWORDS ? prog_words[HDR_DWORDS + OP_DWORDS + lane] : 0u;
#define HDR(field) ((int)__builtin_amdgcn_readlane(hdr_v, (int)(offsetof(Fused16Program, field) / 4)))
  const int n_pos = n_dev ? *n_dev : n_host;
  const int P = (n_pos + (int)gridDim.x - 1) / (int)gridDim.x;
  const int p0 = blockIdx.x * P;
  if (p0 >= n_pos) return;                                  // uniform per workgroup
  const int np = min(P, n_pos - p0);
  const int hw = HDR(hw), H = HDR(h), Wd = HDR(wd), n_ops = HDR(n_ops);
  const int h_in_off = HDR(in_off), h_in_cs = HDR(in_cs), h_in_ps = HDR(in_ps), zrow_index = HDR(zrow_index);   // (every buffer's row of zeros)
  const int rows = np * hw, row_tiles = (rows + 15) >> 4;
  constexpr int ntaps = HEX ? 7 : 9;
  const int kq = lane >> 4;
#ifdef NZ_FUSED_STAMPS
  unsigned long long tk_in = 0, tk_job[32], tk_bar[32], tk_k[32], tk_fin = 0, ts = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 32; ++i) tk_k[i] = 0;
#endif

#ifdef NZ_FUSED_STAMPS
  unsigned long long tk_args = 0; asm volatile("" :: "s"(np), "s"(hw), "s"(n_ops)); FSTAMP(tk_args);
#endif
  // K groups reach past a narrow layer's channels and row tiles past the last row (zero weights, discarded rows): what
  // they read must be numbers, so everything starts as zeros
  for (int i = HDR(clear_from) + tid * 4, end = HDR(lds_floats); i < end; i += FUSED_THREADS * 4) *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 3 * h_in_cs; i += FUSED_THREADS)     // the input's row of zeros
    lds[h_in_off + (i / h_in_cs) * h_in_ps + zrow_index * h_in_cs + i % h_in_cs] = 0.f;
  // (no barrier here: the input rows written next are none of the floats cleared above; the barrier after them covers both)
#ifdef NZ_FUSED_STAMPS
  unsigned long long tk_clear = 0; FSTAMP(tk_clear);
#endif
  {   // this workgroup's input rows, split into pieces (global row of (position n, cell c): ((n >> 4) * hw + c) * 16 + (n & 15))
    const int cs = h_in_cs, ps = h_in_ps;
    const int chunks = in_channels >> 3;                    // eight channels = one 16-byte chunk of a piece
    for (int i = tid; i < rows * chunks; i += FUSED_THREADS) {
      const int r = i / chunks, c8 = i - r * chunks;
      const int pl = r / hw, cell = r - pl * hw, n = p0 + pl;
      const size_t grow = ((size_t)(n >> 4) * hw + cell) * 16 + (n & 15);
      const float* src = in_rows + grow * in_channels + c8 * 8;
      u32x4 q0, q1, q2;
      wide_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), q0, q1, q2);
      float* d = lds + h_in_off + r * cs + ((c8 ^ ((r >> 2) & 3)) << 2);
      *reinterpret_cast<u32x4*>(d) = q0;
      *reinterpret_cast<u32x4*>(d + ps) = q1;
      *reinterpret_cast<u32x4*>(d + 2 * ps) = q2;
    }
  }
#ifdef NZ_FUSED_STAMPS
  unsigned long long tk_split = 0; FSTAMP(tk_split);
#endif
  const int wbuf_off0 = HDR(wbuf_off[0]), wbuf_off1 = HDR(wbuf_off[1]);
  Fused16Op d0, d1;
  {
    uint32_t words[OP_DWORDS];
#pragma unroll
    for (int i = 0; i < OP_DWORDS; ++i) words[i] = __builtin_amdgcn_readlane(op0_v, i);
    __builtin_memcpy(&d0, words, sizeof(Fused16Op));
#pragma unroll
    for (int i = 0; i < OP_DWORDS; ++i) words[i] = __builtin_amdgcn_readlane(op1_v, i);
    __builtin_memcpy(&d1, words, sizeof(Fused16Op));
  }
  if (n_ops > 0 && d0.w_lds) {
    u32x4 wreg[FUSED16_WREGS];
    fetch_weights16(d0, wreg, tid);
    store_weights16(d0, lds + (d0.w_slot ? wbuf_off1 : wbuf_off0), wreg, tid);
  }
  __syncthreads();
  FSTAMP(tk_in);

  int rt_cached = -1, srow[ntaps];
  // A layer's descriptor is fetched two layers ahead, and with VECTOR loads (lane i holds dword i; v_readlane makes
  // the scalars a layer later): scalar loads come back out of order, so the first wait for an LDS read behind one waits
  // for memory too -- a microsecond per layer.
  const int zero_at_op = HDR(zero_at_op), n_zero = HDR(n_zero), zero_len = HDR(zero_len);
  const gptr1u ops_words = (gptr1u)reinterpret_cast<const uint32_t*>(prog->ops);
  uint32_t dvec = (n_ops > 2 && lane < OP_DWORDS) ? ops_words[2 * OP_DWORDS + lane] : 0u;
  for (int o = 0; o < n_ops; ++o) {
    const Fused16Op op = d0, next = d1;       // both read from memory at least a layer ago
    d0 = d1;
    if (o + 2 < n_ops) {
      uint32_t words[OP_DWORDS];
#pragma unroll
      for (int i = 0; i < OP_DWORDS; ++i) words[i] = __builtin_amdgcn_readlane(dvec, i);
      __builtin_memcpy(&d1, words, sizeof(Fused16Op));
    }
    if (o + 3 < n_ops && lane < OP_DWORDS) dvec = ops_words[(o + 3) * OP_DWORDS + lane];
#ifdef NZ_FUSED_STAMPS
    asm volatile("" :: "s"(d1.off0), "s"(d1.kg0));
    const unsigned long long t_top = __builtin_amdgcn_s_memtime();
#endif
    if (o == zero_at_op) {                            // (the previous layer's barrier is behind us; the first reader is two layers on)
      for (int i = tid; i < n_zero * zero_len; i += FUSED_THREADS)
        lds[prog->zero_off[i / zero_len] + i % zero_len] = 0.f;
    }
    const bool next_lds = o + 1 < n_ops && next.w_lds;
    u32x4 wreg[FUSED16_WREGS];
    bool to_fetch = next_lds;                 // the wavefront's first job issues the loads (a wavefront without one: below)
#ifdef NZ_ABL_F16_NOFETCH
    to_fetch = false;
#endif
    if (to_fetch && !op.w_lds) {
      fetch_weights16(next, wreg, tid);
      to_fetch = false;
    }
#ifdef NZ_FUSED_STAMPS
    const unsigned long long t_fetch = __builtin_amdgcn_s_memtime();
    if (o == 3) { tk_args = t_top - ts; tk_clear = t_fetch - t_top; }
#endif
    const int kgt = op.kg0 + op.kg1;
    const int n_jobs = row_tiles * op.ntiles;
    const float* wbuf = lds + (op.w_slot ? wbuf_off1 : wbuf_off0);
    for (int job = wave; job < n_jobs; job += FUSED_WAVES) {
      const int rt = job / op.ntiles, ct = job - rt * op.ntiles;
      if (rt != rt_cached) {
        rt_cached = rt;
        const int row = rt * 16 + (lane & 15);
        const bool row_ok = row < rows;
        const int pl = row / hw, cell = row - pl * hw;
        const int cy = cell / Wd, cx = cell - cy * Wd;
#pragma unroll
        for (int tap = 0; tap < ntaps; ++tap) {
          const int dy = HEX ? (tap < 3 ? tap - 1 : ((tap - 3) & 1) - 1 + (cx & 1)) : tap / 3 - 1;
          const int dx = HEX ? (tap < 3 ? 0 : (tap < 5 ? -1 : 1)) : tap % 3 - 1;
          const bool on = row_ok && (unsigned)(cy + dy) < (unsigned)H && (unsigned)(cx + dx) < (unsigned)Wd;
          srow[tap] = on ? row + dy * Wd + dx : zrow_index;
        }
      }
#ifdef NZ_FUSED_STAMPS
      const unsigned long long j0 = __builtin_amdgcn_s_memtime();
#endif
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef NZ_ABL_F16_NOK
      if (n_ops < 0)
#endif
      {
      const float* wl = wbuf + (size_t)ct * op.w_chunks * 4 + lane * 4;            // LDS copy of the column tile
      const uint32_t* wg = op.w + (size_t)ct * op.w_chunks * 4 + lane * 4;         // packed stream in L2
      if (op.w_lds && kgt == 1) conv16_job<ntaps, 1, true>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      else if (op.w_lds && kgt == 2) conv16_job<ntaps, 2, true>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      else if (kgt == 1) conv16_job<ntaps, 1, false>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      else if (kgt == 2) conv16_job<ntaps, 2, false>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      else if (kgt == 3) conv16_job<ntaps, 3, false>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      else conv16_job<ntaps, 4, false>(acc, lds, srow, op.off0, op.cs0, op.ps0, kq, wl, wg, to_fetch, next, wreg, tid);
      }
      to_fetch = false;
#ifdef NZ_FUSED_STAMPS
      asm volatile("" :: "v"(acc));
      if (o < 32) tk_k[o] += __builtin_amdgcn_s_memtime() - j0;
#endif
      // epilogue: this lane's row, its four channels; the four values go through each step TOGETHER (one wave-uniform
      // switch, then four independent chains the scheduler interleaves)
      const int orow = rt * 16 + (lane & 15), c0 = ct * 16 + (lane >> 4) * 4;
      if (orow < rows) {
        const int chunk = (((c0 >> 3) ^ ((orow >> 2) & 3)) << 2) + ((c0 & 7) >> 1);     // float offset of the 8 bytes in the row
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[r];
        if (op.offr >= 0) {                                 // the residual's pieces add up to the float32 it was split from
          const float* rp = lds + op.offr + orow * op.csr + chunk;
          const uint2 q0 = *reinterpret_cast<const uint2*>(rp), q1 = *reinterpret_cast<const uint2*>(rp + op.psr),
                      q2 = *reinterpret_cast<const uint2*>(rp + 2 * op.psr);
          auto lo = [](uint32_t w) { return __builtin_bit_cast(float, w << 16); };
          auto hi = [](uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); };
          v[0] += (lo(q0.x) + lo(q1.x)) + lo(q2.x);
          v[1] += (hi(q0.x) + hi(q1.x)) + hi(q2.x);
          v[2] += (lo(q0.y) + lo(q1.y)) + lo(q2.y);
          v[3] += (hi(q0.y) + hi(q1.y)) + hi(q2.y);
        }
#ifdef NZ_ABL_F16_NOACT
        switch (0) {
#else
        switch (op.act) {
#endif
          case 1:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            break;
          case 2:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            break;
          case 3:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : fast_expm1(v[r]);
            break;
          default: break;
        }
        if (op.psd == 0) {                                  // float32 rows (logits, value plane): 16 bytes
          *reinterpret_cast<f32x4*>(lds + op.offd + orow * op.csd + c0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          uint16_t h3[4][3];
#pragma unroll
          for (int r = 0; r < 4; ++r) split3_bits(v[r], h3[r]);
          float* dp = lds + op.offd + orow * op.csd + chunk;
#pragma unroll
          for (int piece = 0; piece < 3; ++piece)
            *reinterpret_cast<uint2*>(dp + piece * op.psd) =
                uint2{(uint32_t)h3[0][piece] | ((uint32_t)h3[1][piece] << 16), (uint32_t)h3[2][piece] | ((uint32_t)h3[3][piece] << 16)};
        }
      }
    }
    if (to_fetch) fetch_weights16(next, wreg, tid);
#ifdef NZ_FUSED_STAMPS
    if (o < 32) FSTAMP(tk_job[o]);
#endif
    float* const next_wbuf = lds + (next.w_slot ? wbuf_off1 : wbuf_off0);
#ifndef NZ_ABL_F16_NOFETCH
    if (next_lds && !next.w_after_barrier) store_weights16(next, next_wbuf, wreg, tid);
#endif
    __syncthreads();
#ifndef NZ_ABL_F16_NOFETCH
    if (next_lds && next.w_after_barrier) {
      store_weights16(next, next_wbuf, wreg, tid);
      __syncthreads();
    }
#endif
#ifdef NZ_FUSED_STAMPS
    if (o < 32) FSTAMP(tk_bar[o]);
#endif
  }

  // softmax over ALL logits and value = tanh(mean) (the two outputs are float32 rows).  A workgroup has few positions and
  // sixteen wavefronts: WPP wavefronts share a position's logits; their partial maxima and sums meet in LDS (the first
  // floats of weight buffer 1, which nothing reads after the last layer's barrier).
  float* pol = lds + HDR(pol_off);
  const int pp = HDR(pol_cs);
  const float* val = lds + HDR(val_off);
  const int vp = HDR(val_cs);
  const int A = HDR(planes) * hw;
  const int WPP = np <= 4 ? 4 : (np <= 8 ? 2 : 1), groups = FUSED_WAVES / WPP;
  const int group = wave / WPP, sub = wave - group * WPP, step = 64 * WPP;
  const int i0 = sub * 64 + lane;
  const int cell0 = i0 % hw, plane0 = i0 / hw, dcell = step % hw, dplane = step / hw;
  float* red = lds + h_in_off;                             // [2][FUSED_WAVES]
  for (int base = 0; base < np; base += groups) {           // (the same trip count for every wavefront: barriers inside)
    const int pl = base + group;
    const bool on = pl < np;
    const size_t n = (size_t)(p0 + pl);
    float* prow = pol + pl * hw * pp;
    float mx = -INFINITY;
    if (on)
      for (int i = i0, cell = cell0, plane = plane0; i < A; i += step) {
        const float v = prow[cell * pp + plane];
        if (logits) logits[n * A + i] = v;
        mx = fmaxf(mx, v);
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
    if (!probs) continue;                                   // (uniform)
    for (int w = 32; w; w >>= 1) mx = fmaxf(mx, __shfl_xor(mx, w));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    for (int j = 0; j < WPP; ++j) mx = fmaxf(mx, red[group * WPP + j]);
    float sum = 0.f;
    if (on)
      for (int i = i0, cell = cell0, plane = plane0; i < A; i += step) {
        const float e = expf(prow[cell * pp + plane] - mx);
        prow[cell * pp + plane] = e;
        sum += e;
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
    for (int w = 32; w; w >>= 1) sum += __shfl_xor(sum, w);
    if (lane == 0) red[FUSED_WAVES + wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int j = 0; j < WPP; ++j) sum += red[FUSED_WAVES + group * WPP + j];
    if (on)
      for (int i = i0, cell = cell0, plane = plane0; i < A; i += step) {
        probs[n * A + i] = prow[cell * pp + plane] / sum;
        cell += dcell; plane += dplane;
        if (cell >= hw) { cell -= hw; ++plane; }
      }
    if (base + groups < np) __syncthreads();                // `red` is written again
  }
  for (int pl = wave; pl < np; pl += FUSED_WAVES) {
    float sv = 0.f;
    for (int c = lane; c < hw; c += 64) sv += val[(pl * hw + c) * vp];
    for (int w = 32; w; w >>= 1) sv += __shfl_xor(sv, w);
    if (lane == 0) value[(size_t)(p0 + pl)] = tanhf(sv / (float)hw);
  }
#ifdef NZ_FUSED_STAMPS
  FSTAMP(tk_fin);
  if (blockIdx.x == 0 && (tid == 0 || tid == FUSED_THREADS - 64)) {
    printf("fused16 wg0 wave %d np %d rows %d: args %llu clear %llu split %llu rest-of-input %llu finalize %llu\n", wave, np, rows, tk_args, tk_clear, tk_split, tk_in, tk_fin);
    for (int o = 0; o < n_ops && o < 32; ++o) printf("  wave %d op %d kg %d ntiles %d wlds %d slot %d: jobs %llu (k loop %llu) barrier+stage %llu\n", wave, o, prog->ops[o].kg0, prog->ops[o].ntiles, prog->ops[o].w_lds, prog->ops[o].w_slot, tk_job[o], tk_k[o], tk_bar[o]);
  }
#endif
}

struct ConvOp {
  int src0, src1, res, dst;     // buffer ids (-1: none)
  int weight;                   // index into packed weights
  int act;
};

struct PackedConv {
  float* dev = nullptr;
  uint32_t* dev_split = nullptr;            // wide layers only (conv_wide_kernel)
  uint32_t* dev16 = nullptr;                // fused16_net_kernel's stream [col tile][tap][32-ch group][piece][lane][4]
  int kg0_32 = 0, kg1_32 = 0;
  int c0p = 0, c1p = 0, coutp = 0, cout = 0, cin = 0;
};

}  // namespace

struct nz_boardnet {
  int device = 0;
  nz_net_desc net{};
  int rows = 0, cols = 0, hw = 0, max_batch = 0;
  int inp = 0;                              // padded input channels
  int widthp = 0;
  std::vector<float*> buffers;              // 0: input rows, 1..3: trunk, 4: policy hidden, 5: policy logits, 6,7: value chain
  std::vector<int> buffer_channels;
  std::vector<PackedConv> convs;
  std::vector<ConvOp> ops;
  int policy_buf = -1, value_buf = -1;
  bool ready = false;
  int64_t flops = 0;
  // one-launch form (fused_net_kernel), built by set_weights when the activations of a workgroup's share fit in LDS
  FusedProgram* fused_dev = nullptr;
  bool use_fused = true;
  int fused_grid = 0;
  size_t fused_lds_bytes = 0;
  Fused16Program* fused16_dev = nullptr;    // the same on the BF16 matrix cores (ConvNet), preferred when built
  Fused16Program* wave_dev = nullptr;       // one position per wavefront (boardnet_wave_program), built on first request
  nz::WaveNet wave{};
  int wave_state = 0;                       // 0 not tried, 1 built, -1 not possible (wave_why)
  std::string wave_why;
  int fused16_grid = 0;
  size_t fused16_lds_bytes = 0;
  std::string error;
};

namespace {
std::string g_err;
nz_status bfail(nz_boardnet* h, nz_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  (h ? h->error : g_err) = buf;
  return st;
}
#define B_HIP(h, call)                                                                            \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) return bfail((h), NZ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

int pad16(int c) { return (c + 15) / 16 * 16; }

std::vector<int> head_channels(int width, int out, int layers) {   // blocks.py:56-66,144-153
  std::vector<int> ch{width};
  const double step = (double)(out - width) / layers;
  double prev = width;
  for (int i = 0; i < layers; ++i) { prev += step; ch.push_back((int)prev); }
  return ch;
}

// weights [cout][c0 + c1][k][k] (k = 1 or 3) -> the per-lane stream conv_kernel reads.
// weights -> the per-lane stream conv_kernel reads.  Square: w0 = [cout][c0 + c1][k][k] (k = 1 or 3), w1 unused.
// Hexagonal (hexagdly.Conv2d, kernel_size 1): w0 = kernel0 [cout][cin][3][1] (the cell's own column: N, centre, S),
// w1 = kernel1 [cout][cin][2][2] ([upper, lower] x [left column, right column]).
bool pack(nz_boardnet* h, const float* w0, const float* w1, int cout, int c0, int c1, int k, int c0p, int c1p) {
  const bool hex = h->net.hex != 0;
  const int ntaps = hex ? 7 : 9;
  PackedConv pc;
  pc.cout = cout; pc.coutp = pad16(cout); pc.cin = c0 + c1; pc.c0p = c0p; pc.c1p = c1p;
  const int ntiles = pc.coutp / 16, kgt = (c0p + c1p) / 16, cin = c0 + c1;
  std::vector<float> host((size_t)ntiles * (ntaps + 1) * kgt * 256, 0.f);      // last tap: zeros (see conv_kernel)
  std::vector<float> wh((size_t)cout * cin * (hex ? 3 : k * k)), wh1(hex ? (size_t)cout * cin * 4 : 0);
  if (hipMemcpy(wh.data(), w0, wh.size() * sizeof(float), hipMemcpyDefault) != hipSuccess) return false;
  if (hex && hipMemcpy(wh1.data(), w1, wh1.size() * sizeof(float), hipMemcpyDefault) != hipSuccess) return false;
  auto weight = [&](int co, int ci, int tap) -> float {
    if (hex) {
      if (tap < 3) return wh[((size_t)co * cin + ci) * 3 + tap];
      const int side = tap >= 5, lower = (tap - 3) & 1;                      // 3 NW, 4 SW, 5 NE, 6 SE
      return wh1[(((size_t)co * cin + ci) * 2 + lower) * 2 + side];
    }
    if (k == 1) return tap == 4 ? wh[(size_t)co * cin + ci] : 0.f;
    return wh[((size_t)co * cin + ci) * 9 + tap];
  };
  for (int nt = 0; nt < ntiles; ++nt)
    for (int tap = 0; tap < ntaps; ++tap)
      for (int kg = 0; kg < kgt; ++kg)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int co = nt * 16 + (lane & 15);
            int ch = kg * 16 + (lane >> 4) * 4 + j;          // channel within the padded concat
            int cin_idx;
            if (ch < c0p) cin_idx = ch < c0 ? ch : -1;
            else { ch -= c0p; cin_idx = ch < c1 ? c0 + ch : -1; }
            if (co >= cout || cin_idx < 0) continue;
            host[(((size_t)nt * (ntaps + 1) + tap) * kgt + kg) * 256 + lane * 4 + j] = weight(co, cin_idx, tap);
          }
  if (hipMalloc((void**)&pc.dev, host.size() * sizeof(float)) != hipSuccess) return false;
  if (hipMemcpy(pc.dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
  if (pc.coutp <= 64) {     // narrow layer: the three-piece bf16 stream of the one-launch BF16 network
    pc.kg0_32 = (c0p + 31) / 32; pc.kg1_32 = (c1p + 31) / 32;
    const int kg32 = pc.kg0_32 + pc.kg1_32;
    std::vector<uint32_t> sp((size_t)ntiles * ntaps * kg32 * 3 * 64 * 4, 0u);
    auto pieces16 = [](float a, uint16_t p3[3]) {                     // a = p3[0] + p3[1] + p3[2], bf16 each (truncating)
      float r = a;
      for (int i = 0; i < 3; ++i) {
        uint32_t bits;
        memcpy(&bits, &r, 4);
        bits &= 0xFFFF0000u;
        p3[i] = (uint16_t)(bits >> 16);
        float t;
        memcpy(&t, &bits, 4);
        r = r - t;
      }
    };
    auto cin16 = [&](int kg, int within) {                            // channel `within` of 32-channel group kg -> tensor index
      if (kg < pc.kg0_32) { const int ch = kg * 32 + within; return ch < c0 ? ch : -1; }
      const int ch = (kg - pc.kg0_32) * 32 + within;
      return ch < c1 ? c0 + ch : -1;
    };
    for (int nt = 0; nt < ntiles; ++nt)
      for (int tap = 0; tap < ntaps; ++tap)
        for (int kg = 0; kg < kg32; ++kg)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int co = nt * 16 + (lane & 15);
              const int ci0 = cin16(kg, (lane >> 4) * 8 + 2 * j), ci1 = cin16(kg, (lane >> 4) * 8 + 2 * j + 1);
              uint16_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
              if (co < cout && ci0 >= 0) pieces16(weight(co, ci0, tap), lo);
              if (co < cout && ci1 >= 0) pieces16(weight(co, ci1, tap), hi);
              for (int piece = 0; piece < 3; ++piece)
                sp[(((((size_t)nt * ntaps + tap) * kg32 + kg) * 3 + piece) * 64 + lane) * 4 + j] =
                    (uint32_t)lo[piece] | ((uint32_t)hi[piece] << 16);
            }
    if (hipMalloc((void**)&pc.dev16, sp.size() * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipMemcpy(pc.dev16, sp.data(), sp.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return false;
  }
  if (pc.coutp % 128 == 0 && c0p % 32 == 0 && c1p % 32 == 0) {      // wide layer: the split form for conv_wide_kernel
    const int kqt = (c0p + c1p) / 32, cts = pc.coutp / 128;
    std::vector<uint32_t> sp((size_t)cts * (ntaps + 1) * kqt * 3 * 8 * 64 * 4, 0u);
    auto cin_of = [&](int ch) {                                       // channel within the padded concat -> tensor index
      if (ch < c0p) return ch < c0 ? ch : -1;
      ch -= c0p;
      return ch < c1 ? c0 + ch : -1;
    };
    auto pieces = [](float a, uint16_t p3[3]) {                       // a = p3[0] + p3[1] + p3[2], bf16 each
      float r = a;
      for (int i = 0; i < 3; ++i) {
        uint32_t bits;
        memcpy(&bits, &r, 4);
        bits &= 0xFFFF0000u;
        p3[i] = (uint16_t)(bits >> 16);
        float t;
        memcpy(&t, &bits, 4);
        r = r - t;
      }
    };
    for (int ct = 0; ct < cts; ++ct)
      for (int tap = 0; tap < ntaps; ++tap)
        for (int kq = 0; kq < kqt; ++kq)
          for (int nt = 0; nt < 8; ++nt)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int co = ct * 128 + nt * 16 + (lane & 15);
                const int ci0 = cin_of(kq * 32 + (lane >> 4) * 8 + 2 * j), ci1 = cin_of(kq * 32 + (lane >> 4) * 8 + 2 * j + 1);
                uint16_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                if (co < cout && ci0 >= 0) pieces(weight(co, ci0, tap), lo);
                if (co < cout && ci1 >= 0) pieces(weight(co, ci1, tap), hi);
                for (int piece = 0; piece < 3; ++piece)
                  sp[((((((size_t)ct * (ntaps + 1) + tap) * kqt + kq) * 3 + piece) * 8 + nt) * 64 + lane) * 4 + j] =
                      (uint32_t)lo[piece] | ((uint32_t)hi[piece] << 16);
              }
    if (hipMalloc((void**)&pc.dev_split, sp.size() * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipMemcpy(pc.dev_split, sp.data(), sp.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return false;
  }
  h->convs.push_back(pc);
  // algorithmic flops per position: taps inside the board only
  int64_t taps = 0;
  if (hex) {
    for (int r = 0; r < h->rows; ++r)
      for (int c = 0; c < h->cols; ++c) {
        const int up = (c & 1) ? r : r - 1, lo = up + 1;
        taps += 1 + (r > 0) + (r + 1 < h->rows);
        for (int dc = -1; dc <= 1; dc += 2)
          if (c + dc >= 0 && c + dc < h->cols) taps += (up >= 0 && up < h->rows) + (lo >= 0 && lo < h->rows);
      }
  } else {
    taps = k == 1 ? (int64_t)h->hw : (int64_t)(3 * h->rows - 2) * (3 * h->cols - 2);
  }
  h->flops += 2 * taps * cin * cout;
  return true;
}

template <int MT, int NT, int DEPTH>
void launch_conv(const ConvArgs& a, int ntiles, hipStream_t s) {
  const int groups = (a.n_host + 15) / 16;
  const int tasks = (groups + MT - 1) / MT * a.hw;           // wavefronts: one board cell of MT position groups
  dim3 grid((tasks + 3) / 4, ntiles / NT);
  if (a.hex) hipLaunchKernelGGL((conv_kernel<MT, NT, DEPTH, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_kernel<MT, NT, DEPTH, false>), grid, dim3(256), 0, s, a);
}

// Tile shape per layer: NT = as many 16-column tiles as divide the layer's width (up to 4), so
// the activation fragment is reused NT times; MT = 2 position groups per wavefront for wide
// layers once there are enough positions to keep every SIMD busy anyway.  Measured alternatives
// (MT = 4; weight fragments shared through LDS by the four wavefronts of a workgroup; operand
// ablations) are in profiles/r01_boardnet_tiles.txt.
void dispatch_conv(const ConvArgs& a, int ntiles, hipStream_t s) {
  static const int force_mt = getenv("NZ_BOARDNET_MT") ? atoi(getenv("NZ_BOARDNET_MT")) : 0;   // tuning experiments
  const int kgt = (a.c0 + a.c1) / 16;
  static const int force_wide = getenv("NZ_BOARDNET_WIDE") ? atoi(getenv("NZ_BOARDNET_WIDE")) : -1;     // tuning experiments
  const int wide_tiles = ((a.n_host + 15) / 16 + WIDE_GROUPS - 1) / WIDE_GROUPS * a.hw * (ntiles / WIDE_NT);
  // one workgroup per tile and per CU: below ~160 tiles the chip is too empty and the per-wavefront kernel wins
  // (w256 5x5: 512 positions = 100 tiles 53 vs 59 TFLOP/s, 1024 = 200 tiles 96 vs 81; w128: 100 tiles 50 vs 58, 400 tiles 91 vs 81)
  // ... and a tile is 256 positions of one cell: with fewer than half of them live (a handful of games on a big board)
  // most of its MFMAs multiply padding, and the per-wavefront kernel wins again
  const bool tiles_filled = (a.n_host + 15) / 16 >= WIDE_GROUPS / 2;
  if (a.ws != nullptr && (force_wide < 0 ? (wide_tiles >= 160 && tiles_filled) : force_wide != 0)) {
    const int groups = (a.n_host + 15) / 16;
    dim3 grid((groups + WIDE_GROUPS - 1) / WIDE_GROUPS * a.hw, ntiles / WIDE_NT);
    if (a.hex) hipLaunchKernelGGL(conv_wide_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv_wide_kernel<false>, grid, dim3(256), 0, s, a);
    return;
  }
  if (ntiles % 4 == 0) {
    const int mt = force_mt ? force_mt : (a.n_host >= 1024 && kgt >= 8) ? 2 : 1;
    if (mt == 2) launch_conv<2, 4, 3>(a, ntiles, s);
    else launch_conv<1, 4, 4>(a, ntiles, s);
  } else if (ntiles % 3 == 0) {
    launch_conv<1, 3, 4>(a, ntiles, s);
  } else if (ntiles % 2 == 0) {
    const int mt = force_mt ? force_mt : a.n_host >= 2048 ? 2 : 1;      // + 7 % at 2048 and 8192 positions; MT = 4 loses
    if (mt >= 2) launch_conv<2, 2, 3>(a, ntiles, s);
    else launch_conv<1, 2, 4>(a, ntiles, s);
  } else {
    const int mt = force_mt ? force_mt : a.n_host >= 2048 ? 2 : 1;
    if (mt >= 2) launch_conv<2, 1, 3>(a, ntiles, s);
    else launch_conv<1, 1, 4>(a, ntiles, s);
  }
}
}  // namespace


namespace {
// The one-launch program for fused_net_kernel, when a workgroup's share of the largest batch fits in LDS: the ops of
// forward_impl with LDS buffers instead of HBM ones.  The value head's two buffers reuse the two trunk buffers that are
// free once the trunk is done (`trunk_out` holds its output).
void build_fused(nz_boardnet* h, int trunk_out) {
  static const int force = getenv("NZ_BOARDNET_FUSED") ? atoi(getenv("NZ_BOARDNET_FUSED")) : -1;   // tuning experiments
  if (force == 0 || (int)h->ops.size() > FUSED_MAX_OPS) return;
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || n_cu <= 0) return;
  const int grid = h->max_batch < n_cu ? h->max_batch : n_cu;
  const int p_max = (h->max_batch + grid - 1) / grid;
  const int rows_max = (p_max * h->hw + 15) / 16 * 16;
  int free_ids[2], nf = 0;
  for (int b = 1; b <= 3; ++b)
    if (b != trunk_out && nf < 2) free_ids[nf++] = b;
  // the heads run after the trunk, policy first: its hidden layer and the value head's two buffers live in the two
  // trunk buffers that are free by then (when they are wide enough); the logits keep their own buffer for the softmax
  int alias[FUSED_BUFFERS] = {0, 1, 2, 3, 4, 5, free_ids[0], free_ids[1]};
  if (h->buffer_channels[4] <= h->buffer_channels[free_ids[0]]) alias[4] = free_ids[0];
  for (int b = 6; b < 8; ++b)
    if (h->buffer_channels[b] > h->buffer_channels[alias[b]]) return;      // cannot happen: the value head narrows
  FusedProgram pg;
  memset(&pg, 0, sizeof(pg));
  int buf_off[FUSED_BUFFERS], buf_cs[FUSED_BUFFERS];
  size_t off = 0;
  pg.zrow_off = 0;
  off += FUSED_ZROW;
  for (int b = 0; b < FUSED_BUFFERS; ++b) {
    if (alias[b] != b) continue;
    buf_off[b] = (int)off;
    buf_cs[b] = h->buffer_channels[b] + FUSED_PAD;
    off += (size_t)rows_max * buf_cs[b];
  }
  for (int b = 0; b < FUSED_BUFFERS; ++b)
    if (alias[b] != b) { buf_off[b] = buf_off[alias[b]]; buf_cs[b] = buf_cs[alias[b]]; }
  const int n_ops = (int)h->ops.size();
  pg.ops_off = -1;
  if (n_ops <= FUSED_LDS_OPS) { pg.ops_off = (int32_t)off; off += (size_t)n_ops * FUSED_OP_WORDS; off = (off + 3) / 4 * 4; }
  const size_t budget = 156 * 1024 / sizeof(float);      // 160 KB of LDS per CU
  if (off > budget) return;
  pg.n_ops = n_ops;
  pg.hw = h->hw; pg.h = h->rows; pg.wd = h->cols;
  pg.planes = h->net.policy_channels; pg.hex = h->net.hex ? 1 : 0;
  pg.in_off = buf_off[0]; pg.in_cs = buf_cs[0];
  pg.pol_off = buf_off[h->policy_buf]; pg.pol_cs = buf_cs[h->policy_buf];
  pg.val_off = buf_off[h->value_buf]; pg.val_cs = buf_cs[h->value_buf];
  const int ntaps = h->net.hex ? 7 : 9;
  // layers whose weights fit in what is left of LDS (and in the registers that carry them there) are staged
  const size_t w_cap = std::min(budget - off, (size_t)FUSED_WREGS * FUSED_THREADS * 4);
  size_t wbuf = 0;
  int last_input_reader = -1;
  for (int i = 0; i < n_ops; ++i) {
    const ConvOp& op = h->ops[i];
    const PackedConv& pc = h->convs[i];
    FusedOp& f = pg.ops[i];
    f.w = pc.dev;
    f.off0 = buf_off[op.src0]; f.cs0 = buf_cs[op.src0];
    f.off1 = op.src1 >= 0 ? buf_off[op.src1] : -1; f.cs1 = op.src1 >= 0 ? buf_cs[op.src1] : 0;
    f.offd = buf_off[op.dst]; f.csd = buf_cs[op.dst];
    f.offr = op.res >= 0 ? buf_off[op.res] : -1; f.csr = op.res >= 0 ? buf_cs[op.res] : 0;
    f.kg0 = pc.c0p / 16; f.kg1 = op.src1 >= 0 ? pc.c1p / 16 : 0;
    f.ntiles = pc.coutp / 16; f.act = op.act;
    f.w_chunks = ntaps * (f.kg0 + f.kg1) * 64;
    const size_t wf = (size_t)f.ntiles * f.w_chunks * 4;
    f.w_lds = wf <= w_cap ? 1 : 0;
    if (f.w_lds) wbuf = std::max(wbuf, wf);
    if (op.src0 == 0 || op.src1 == 0) last_input_reader = i;
  }
  pg.wbuf_off[0] = (int32_t)off;
  off += wbuf;
  // a second weight buffer lets a layer's weights be written while the previous layer still reads its own: the input
  // rows' space once no later layer reads the input (feed-forward nets), else whatever LDS is left, else none
  int second_from = n_ops;                    // first layer index that may USE the second buffer
  pg.wbuf_off[1] = pg.wbuf_off[0];
  if (off + wbuf <= budget) {
    pg.wbuf_off[1] = (int32_t)off;
    off += wbuf;
    second_from = 0;
  } else if ((size_t)rows_max * buf_cs[0] >= wbuf) {
    pg.wbuf_off[1] = buf_off[0];
    second_from = last_input_reader + 2;      // its weights are stored during the layer before it
  }
  int slot = 0;
  bool first = true;
  for (int i = 0; i < n_ops; ++i) {
    FusedOp& f = pg.ops[i];
    if (!f.w_lds) continue;
    if (first) { f.w_slot = 0; f.w_after_barrier = 0; first = false; slot = 0; continue; }
    // the previous staged layer may still be reading `slot` when this one's weights arrive (only if it is layer i - 1)
    const bool prev_reads = i > 0 && pg.ops[i - 1].w_lds;
    if (prev_reads && i >= second_from && pg.wbuf_off[1] != pg.wbuf_off[0]) { slot ^= 1; f.w_after_barrier = 0; }
    else f.w_after_barrier = prev_reads ? 1 : 0;
    // (the input rows' space can only be slot 1; before `second_from` everything stays in slot 0)
    if (i < second_from) { slot = 0; f.w_after_barrier = prev_reads ? 1 : 0; }
    f.w_slot = slot;
  }
  const size_t bytes = off * sizeof(float);
  const hipError_t e = h->net.hex
      ? hipFuncSetAttribute((const void*)fused_net_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
      : hipFuncSetAttribute((const void*)fused_net_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return; }
  FusedProgram* dev = nullptr;
  if (hipMalloc((void**)&dev, sizeof(pg)) != hipSuccess) return;
  if (hipMemcpy(dev, &pg, sizeof(pg), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return; }
  h->fused_dev = dev;
  h->fused_grid = grid;
  h->fused_lds_bytes = bytes;
}
// The one-launch network on the BF16 matrix cores, for ConvNets (a chain: two activation buffers).  LDS, in floats:
// [zeros][input pieces -- after the first layer: weight buffer 1][weight buffer 0][activations A][activations B]; the
// heads read their weights straight from L2 and keep their buffers (logits, value plane, two hidden buffers) in weight
// buffer 0, which the trunk no longer needs by then.
bool build_fused16_for(nz_boardnet* h, int p_max);
bool build_fused16_resnet(nz_boardnet* h, int p_max);
void build_fused16(nz_boardnet* h) {
  static const int force = getenv("NZ_BOARDNET_FUSED16") ? atoi(getenv("NZ_BOARDNET_FUSED16")) : -1;   // tuning experiments
  const nz_net_desc& nd = h->net;
  if (force == 0 || (nd.arch != NZ_ARCH_CONVNET && nd.arch != NZ_ARCH_RESNET) || (int)h->ops.size() > FUSED_MAX_OPS || h->inp % 8 != 0) return;
  const int n_ops = (int)h->ops.size(), n_trunk = n_ops - 6;     // first layer + num_blocks layers, then 2 + 4 head layers
  if (n_trunk < 1) return;
  for (const PackedConv& pc : h->convs)
    if (!pc.dev16) return;
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || n_cu <= 0) return;
  if (const char* e = getenv("NZ_BOARDNET_CUS")) n_cu = std::max(1, std::min(n_cu, atoi(e)));   // a stream with a CU mask: its share of the chip
  // positions per workgroup: an even share of the largest batch over the CUs if that fits in LDS, else as many as fit (the
  // grid then has more workgroups than CUs: they take the CUs in turn)
  const int p_even = (h->max_batch + n_cu - 1) / n_cu;
  for (int p_max = p_even; p_max >= 1; --p_max)
    if (nd.arch == NZ_ARCH_CONVNET ? build_fused16_for(h, p_max) : build_fused16_resnet(h, p_max)) return;
}

bool build_fused16_for(nz_boardnet* h, int p_max) {
  const nz_net_desc& nd = h->net;
  const int n_ops = (int)h->ops.size(), n_trunk = n_ops - 6;
  const int grid = (h->max_batch + p_max - 1) / p_max;
  const int rows = p_max * h->hw;                               // rows past the last one are never read or written
  const int ntaps = nd.hex ? 7 : 9;
  auto cs_of = [](int channels) { return (channels + 31) / 32 * 16; };           // floats per row and piece
  auto pieces_floats = [&](int channels) { return (size_t)3 * (rows + 1) * cs_of(channels); };   // + the row of zeros
  const size_t budget = 160 * 1024 / sizeof(float);
  Fused16Program pg;
  memset(&pg, 0, sizeof(pg));
  size_t off = 0;
  pg.zrow_index = rows;
  const int W = nd.width;
  // trunk weights: every staged layer has the same shape
  size_t wslot = 0;
  for (int i = 1; i < n_trunk; ++i) {
    const PackedConv& pc = h->convs[i];
    wslot = std::max(wslot, (size_t)(pc.coutp / 16) * ntaps * (pc.kg0_32 + pc.kg1_32) * 3 * 64 * 4);
  }
  if (wslot > (size_t)FUSED16_WREGS * FUSED_THREADS * 4) return false;
  const size_t in_floats = pieces_floats(h->inp);
  const size_t region_i = std::max(in_floats, wslot);
  const int off_i = (int)off; off += region_i;
  const int off_w0 = (int)off;
  const std::vector<int> pcn = head_channels(W, nd.policy_channels, 2);
  const std::vector<int> vcn = head_channels(W, 1, 4);
  const int hidden = std::max(std::max(pcn[1], vcn[1]), std::max(vcn[2], vcn[3]));
  // (float32 rows take whole 16-channel output tiles: the padded channels of the last tile are stored too)
  const int pol_cs = pad16(nd.policy_channels) + FUSED_PAD, val_cs = pad16(1) + FUSED_PAD;
  const size_t pol_floats = ((size_t)rows * pol_cs + 3) / 4 * 4;
  const size_t val_floats = ((size_t)rows * val_cs + 3) / 4 * 4;
  const size_t heads = pol_floats + val_floats + 2 * pieces_floats(hidden);
  const size_t region_w0 = std::max(wslot, heads);
  off += region_w0;
  const int off_a = (int)off; off += pieces_floats(W);
  const int off_b = (int)off; off += pieces_floats(W);
  off = (off + 3) / 4 * 4;
  if (off > budget) return false;
  pg.lds_floats = (int32_t)off;
  // (the staged layers leave finite weights in the first `wslot` floats of weight buffer 0 before the heads move in)
  pg.clear_from = off_w0 + (n_trunk > 1 ? (int)wslot : 0);
  pg.n_ops = n_ops;
  pg.hw = h->hw; pg.h = h->rows; pg.wd = h->cols;
  pg.planes = nd.policy_channels; pg.hex = nd.hex ? 1 : 0;
  pg.in_off = off_i; pg.in_cs = cs_of(h->inp); pg.in_ps = (rows + 1) * pg.in_cs;
  pg.wbuf_off[0] = off_w0; pg.wbuf_off[1] = off_i;
  pg.pol_off = off_w0; pg.pol_cs = pol_cs;
  pg.val_off = off_w0 + (int)pol_floats; pg.val_cs = val_cs;
  const int off_h1 = pg.val_off + (int)val_floats, off_h2 = off_h1 + (int)pieces_floats(hidden);
  const int cs_w = cs_of(W), ps_w = (rows + 1) * cs_w, cs_h = cs_of(hidden), ps_h = (rows + 1) * cs_h;
  struct Place { int off, cs, ps; };
  pg.zero_at_op = n_trunk; pg.n_zero = 6; pg.zero_len = cs_h;
  for (int piece = 0; piece < 3; ++piece) {
    pg.zero_off[piece] = off_h1 + piece * ps_h + rows * cs_h;
    pg.zero_off[3 + piece] = off_h2 + piece * ps_h + rows * cs_h;
  }
  const Place in_pl{off_i, pg.in_cs, pg.in_ps}, a_pl{off_a, cs_w, ps_w}, b_pl{off_b, cs_w, ps_w};
  const Place h1_pl{off_h1, cs_h, ps_h}, h2_pl{off_h2, cs_h, ps_h};
  const Place pol_pl{pg.pol_off, pg.pol_cs, 0}, val_pl{pg.val_off, pg.val_cs, 0};
  const Place trunk_out = (n_trunk - 1) & 1 ? b_pl : a_pl, trunk_other = (n_trunk - 1) & 1 ? a_pl : b_pl;
  for (int i = 0; i < n_ops; ++i) {
    const ConvOp& op = h->ops[i];
    const PackedConv& pc = h->convs[i];
    Fused16Op& f = pg.ops[i];
    if (op.src1 >= 0 || op.res >= 0) return false;                     // a ConvNet has neither
    Place src, dst;
    if (i < n_trunk) {
      src = i == 0 ? in_pl : ((i - 1) & 1 ? b_pl : a_pl);
      dst = i & 1 ? b_pl : a_pl;
    } else {
      const int hidx = i - n_trunk;                              // 0, 1: policy head; 2..5: value head
      switch (hidx) {
        case 0: src = trunk_out; dst = trunk_other; break;
        case 1: src = trunk_other; dst = pol_pl; break;
        case 2: src = trunk_out; dst = h1_pl; break;
        case 3: src = h1_pl; dst = h2_pl; break;
        case 4: src = h2_pl; dst = h1_pl; break;
        default: src = h1_pl; dst = val_pl; break;
      }
    }
    f.w = pc.dev16;
    f.off0 = src.off; f.cs0 = src.cs; f.ps0 = src.ps; f.kg0 = pc.kg0_32;
    f.off1 = -1; f.cs1 = 0; f.ps1 = 0; f.kg1 = 0;
    f.offd = dst.off; f.csd = dst.cs; f.psd = dst.ps;
    f.offr = -1; f.csr = 0; f.psr = 0;
    f.ntiles = pc.coutp / 16; f.act = op.act;
    f.w_chunks = ntaps * pc.kg0_32 * 3 * 64;
    // the K groups a layer reads must exist in its source rows
    if (f.kg0 * 16 > f.cs0) return false;
    const size_t wf = (size_t)f.ntiles * f.w_chunks * 4;
    if (i >= 1 && i < n_trunk) {            // layer 1 into buffer 0 (the input still sits in buffer 1), then alternating
      f.w_lds = 1; f.w_slot = (i - 1) & 1; f.w_after_barrier = 0;
    } else if (i >= n_trunk && wf <= region_i && wf <= (size_t)FUSED16_WREGS * FUSED_THREADS * 4) {
      // head layers: buffer 0 is their activations' home, so all of them use buffer 1, written once the layer before is done
      f.w_lds = 1; f.w_slot = 1; f.w_after_barrier = 1;
    } else {
      f.w_lds = 0; f.w_slot = 0; f.w_after_barrier = 0;
    }
  }
  const size_t bytes = off * sizeof(float);
  const hipError_t e = nd.hex
      ? hipFuncSetAttribute((const void*)fused16_net_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
      : hipFuncSetAttribute((const void*)fused16_net_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  Fused16Program* dev = nullptr;
  if (hipMalloc((void**)&dev, sizeof(pg)) != hipSuccess) return false;
  if (hipMemcpy(dev, &pg, sizeof(pg), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return false; }
  h->fused16_dev = dev;
  h->fused16_grid = grid;
  h->fused16_lds_bytes = bytes;
  return true;
}
// The same for ResNets: three trunk buffers (a block reads its input again as the residual).  LDS, in floats:
// [input pieces -- after the first layer: trunk buffers 2 and 3][the weight buffer][trunk buffer 1][logits][value plane];
// one weight buffer only, so a layer's weights are stored once the layer before it is done (they travel from L2 under its
// MFMAs all the same).  The head layers use the two trunk buffers that are free by then.
bool build_fused16_resnet(nz_boardnet* h, int p_max) {
  const nz_net_desc& nd = h->net;
  const int n_ops = (int)h->ops.size(), n_trunk = n_ops - 6;
  if (n_trunk < 1) return false;
  const int grid = (h->max_batch + p_max - 1) / p_max;
  const int rows = p_max * h->hw;
  const int ntaps = nd.hex ? 7 : 9;
  auto cs_of = [](int channels) { return (channels + 31) / 32 * 16; };
  auto pieces_floats = [&](int channels) { return (size_t)3 * (rows + 1) * cs_of(channels); };
  const size_t budget = 160 * 1024 / sizeof(float);
  Fused16Program pg;
  memset(&pg, 0, sizeof(pg));
  pg.zrow_index = rows;
  const int W = nd.width;
  size_t wslot = 0;
  for (int i = 1; i < n_ops; ++i) {
    const PackedConv& pc = h->convs[i];
    wslot = std::max(wslot, (size_t)(pc.coutp / 16) * ntaps * (pc.kg0_32 + pc.kg1_32) * 3 * 64 * 4);
  }
  if (wslot > (size_t)FUSED16_WREGS * FUSED_THREADS * 4) return false;
  size_t off = 0;
  const size_t act = pieces_floats(W);
  const size_t region_i = std::max(pieces_floats(h->inp), 2 * act);
  const int off_i = (int)off; off += region_i;
  const int off_w0 = (int)off; off += wslot;
  const int off_a = (int)off; off += act;
  const int pol_cs = pad16(nd.policy_channels) + FUSED_PAD, val_cs = pad16(1) + FUSED_PAD;
  const int off_pol = (int)off; off += ((size_t)rows * pol_cs + 3) / 4 * 4;
  const int off_val = (int)off; off += ((size_t)rows * val_cs + 3) / 4 * 4;
  if (off > budget) return false;
  pg.lds_floats = (int32_t)off;
  pg.clear_from = off_a;
  pg.n_ops = n_ops;
  pg.hw = h->hw; pg.h = h->rows; pg.wd = h->cols;
  pg.planes = nd.policy_channels; pg.hex = nd.hex ? 1 : 0;
  pg.in_off = off_i; pg.in_cs = cs_of(h->inp); pg.in_ps = (rows + 1) * pg.in_cs;
  pg.wbuf_off[0] = off_w0; pg.wbuf_off[1] = off_w0;
  pg.pol_off = off_pol; pg.pol_cs = pol_cs;
  pg.val_off = off_val; pg.val_cs = val_cs;
  const int cs_w = cs_of(W), ps_w = (rows + 1) * cs_w;
  struct Place { int off, cs, ps; };
  const Place in_pl{off_i, pg.in_cs, pg.in_ps};
  const Place trunk[4] = {in_pl, {off_a, cs_w, ps_w}, {off_i, cs_w, ps_w}, {off_i + (int)act, cs_w, ps_w}};   // buffer ids 0..3
  // buffers 2 and 3 take over the input's space: their rows of zeros are made when the first layer is done
  pg.zero_at_op = 1; pg.n_zero = 6; pg.zero_len = cs_w;
  for (int piece = 0; piece < 3; ++piece) {
    pg.zero_off[piece] = trunk[2].off + piece * ps_w + rows * cs_w;
    pg.zero_off[3 + piece] = trunk[3].off + piece * ps_w + rows * cs_w;
  }
  const int t_out = h->ops[n_trunk].src0;                        // the trunk's output buffer (what the policy head reads)
  if (t_out < 1 || t_out > 3) return false;
  int free_ids[2], nf = 0;
  for (int b = 1; b <= 3; ++b)
    if (b != t_out) free_ids[nf++] = b;
  const Place pol_pl{off_pol, pol_cs, 0}, val_pl{off_val, val_cs, 0};
  auto place_of = [&](int id, bool last_value) -> Place {         // buffer id of the per-layer form -> where it lives here
    if (id >= 0 && id <= 3) return trunk[id];
    if (id == 4) return trunk[free_ids[0]];                       // policy hidden
    if (id == 5) return pol_pl;
    if (last_value) return val_pl;
    return trunk[free_ids[id == 6 ? 0 : 1]];                      // value chain 6, 7 (6 after the policy head is done with it)
  };
  for (int i = 0; i < n_ops; ++i) {
    const ConvOp& op = h->ops[i];
    const PackedConv& pc = h->convs[i];
    Fused16Op& f = pg.ops[i];
    if (op.src1 >= 0 || !pc.dev16) return false;
    const Place src = place_of(op.src0, false), dst = place_of(op.dst, i == n_ops - 1);
    f.w = pc.dev16;
    f.off0 = src.off; f.cs0 = src.cs; f.ps0 = src.ps; f.kg0 = pc.kg0_32;
    f.off1 = -1; f.cs1 = 0; f.ps1 = 0; f.kg1 = 0;
    f.offd = dst.off; f.csd = dst.cs; f.psd = dst.ps;
    if (op.res >= 0) { const Place r = place_of(op.res, false); f.offr = r.off; f.csr = r.cs; f.psr = r.ps; }
    else { f.offr = -1; f.csr = 0; f.psr = 0; }
    f.ntiles = pc.coutp / 16; f.act = op.act;
    f.w_chunks = ntaps * pc.kg0_32 * 3 * 64;
    if (f.kg0 * 16 > f.cs0) return false;
    const size_t wf = (size_t)f.ntiles * f.w_chunks * 4;
    f.w_lds = i >= 1 && wf <= wslot ? 1 : 0;
    f.w_slot = 0;
    f.w_after_barrier = 1;
  }
  const size_t bytes = off * sizeof(float);
  const hipError_t e = nd.hex
      ? hipFuncSetAttribute((const void*)fused16_net_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
      : hipFuncSetAttribute((const void*)fused16_net_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  Fused16Program* dev = nullptr;
  if (hipMalloc((void**)&dev, sizeof(pg)) != hipSuccess) return false;
  if (hipMemcpy(dev, &pg, sizeof(pg), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return false; }
  h->fused16_dev = dev;
  h->fused16_grid = grid;
  h->fused16_lds_bytes = bytes;
  return true;
}
}  // namespace

// The layer program for ONE wavefront evaluating ONE position (the persistent SCS self-play kernel): activations as
// bf16 pieces in the wavefront's own LDS block, every layer's weights streamed from the packed L2 stream, no barriers.
// Block, in floats: [input pieces -- once the first layer is done: the logits' and the value plane's float32 rows]
// [trunk buffers 1, 2, 3]; the head layers' hidden buffers are the two trunk buffers that are free by then
// (build_fused16_resnet's arrangement).  The leaf's float32 planes are staged over the trunk buffers before they are
// split into the input pieces.
bool nz::boardnet_wave_program(nz_boardnet* h, nz::WaveNet* out, std::string* why) {
  auto no = [&](const char* msg) { h->wave_state = -1; h->wave_why = msg; if (why) *why = msg; return false; };
  if (!h || !h->ready) { if (why) *why = "no weights set"; return false; }
  if (h->wave_state == 1) { *out = h->wave; return true; }
  if (h->wave_state == -1) { if (why) *why = h->wave_why; return false; }
  const nz_net_desc& nd = h->net;
  if (nd.arch != NZ_ARCH_CONVNET && nd.arch != NZ_ARCH_RESNET) return no("only feed-forward nets (ConvNet, ResNet) have a per-wavefront form");
  const int n_ops = (int)h->ops.size(), n_trunk = n_ops - 6;
  if (n_trunk < 1 || n_ops > FUSED_MAX_OPS) return no("too many layers");
  if (h->hw > 32) return no("boards of more than 32 cells do not fit two row tiles");
  if (h->inp % 8 != 0 || h->inp > 128) return no("input planes");
  for (const PackedConv& pc : h->convs)
    if (!pc.dev16 || pc.coutp > 64 || pc.kg0_32 > 4) return no("a layer is wider than 64 channels");
  const int rows = h->hw, R1 = rows + 1;
  // pieces buffers: [row][32-channel group][piece][16 floats] (scs_search.hip, wave_conv); rows 256 bytes apart would
  // share their LDS banks, so an even number of groups gets 16 floats of padding per row
  auto cs_of = [](int channels) { const int kg = (channels + 31) / 32; return kg * 48 + (kg % 2 == 0 ? 16 : 0); };
  auto pieces_floats = [&](int channels) { return (size_t)R1 * cs_of(channels); };
  const int W = nd.width;
  if (h->buffer_channels[4] > h->widthp || h->buffer_channels[6] > h->widthp) return no("head layers wider than the trunk");
  Fused16Program pg;
  memset(&pg, 0, sizeof(pg));
  const int pol_cs = pad16(nd.policy_channels) + FUSED_PAD, val_cs = pad16(1) + FUSED_PAD;
  const size_t pol_floats = ((size_t)rows * pol_cs + 3) / 4 * 4, val_floats = ((size_t)rows * val_cs + 3) / 4 * 4;
  size_t off = 0;
  const int off_i = 0;
  off += std::max(pieces_floats(h->inp), pol_floats + val_floats);
  const size_t act = pieces_floats(W);
  const int off_t = (int)off;
  const size_t stage = (size_t)rows * h->inp;
  off += std::max(3 * act, (stage + 3) / 4 * 4);
  if (off * sizeof(float) > 38 * 1024) return no("the activations of one position do not fit a wavefront's share of LDS");
  pg.lds_floats = (int32_t)off;
  pg.zrow_index = rows;
  pg.n_ops = n_ops;
  pg.hw = h->hw; pg.h = h->rows; pg.wd = h->cols;
  pg.planes = nd.policy_channels; pg.hex = nd.hex ? 1 : 0;
  pg.in_off = off_i; pg.in_cs = cs_of(h->inp); pg.in_ps = 16;
  pg.pol_off = off_i; pg.pol_cs = pol_cs;
  pg.val_off = off_i + (int)pol_floats; pg.val_cs = val_cs;
  const int cs_w = cs_of(W), ps_w = 16;
  struct Place { int off, cs, ps; };
  const Place in_pl{off_i, pg.in_cs, pg.in_ps};
  const Place trunk[4] = {in_pl, {off_t, cs_w, ps_w}, {off_t + (int)act, cs_w, ps_w}, {off_t + 2 * (int)act, cs_w, ps_w}};
  const int t_out = h->ops[n_trunk].src0;
  if (t_out < 1 || t_out > 3) return no("internal: trunk output buffer");
  // The two heads are independent chains behind the trunk (blocks.py:56-66,144-153): the pair's leader runs the policy
  // head's two layers, its helper the value head's four, side by side with no meeting in between -- if the policy head's
  // hidden buffer (which then cannot share a trunk buffer with the value head's) fits behind the logits' and the value
  // plane's rows in the input pieces' space, free since the first layer.
  const size_t hid_floats = pieces_floats(h->buffer_channels[4]);
  const size_t off_hid = (pol_floats + val_floats + 3) / 4 * 4;
  const bool solo = off_hid + hid_floats <= pieces_floats(h->inp) && getenv("NZ_SCS_PERSIST_NO_SOLO") == nullptr;
  const int cs_hid = cs_of(h->buffer_channels[4]);
  int free_ids[2], nf = 0;
  for (int b = 1; b <= 3; ++b)
    if (b != t_out) free_ids[nf++] = b;
  const Place pol_pl{pg.pol_off, pol_cs, 0}, val_pl{pg.val_off, val_cs, 0};
  auto place_of = [&](int id, bool last_value) -> Place {
    if (id >= 0 && id <= 3) return trunk[id];
    if (id == 4) return solo ? Place{off_i + (int)off_hid, cs_hid, 16} : trunk[free_ids[0]];
    if (id == 5) return pol_pl;
    if (last_value) return val_pl;
    return trunk[free_ids[id == 6 ? 0 : 1]];
  };
  const int ntaps = nd.hex ? 7 : 9;
  for (int i = 0; i < n_ops; ++i) {
    const ConvOp& op = h->ops[i];
    const PackedConv& pc = h->convs[i];
    Fused16Op& f = pg.ops[i];
    if (op.src1 >= 0) return no("a layer with two sources");
    if (i > 0 && (op.src0 == 0 || op.res == 0)) return no("the input is read after the first layer");
    const Place src = place_of(op.src0, false), dst = place_of(op.dst, i == n_ops - 1);
    f.w = pc.dev16;
    f.off0 = src.off; f.cs0 = src.cs; f.ps0 = src.ps; f.kg0 = pc.kg0_32;
    f.off1 = -1; f.cs1 = 0; f.ps1 = 0; f.kg1 = 0;
    f.offd = dst.off; f.csd = dst.cs; f.psd = dst.ps;
    if (op.res >= 0) { const Place r = place_of(op.res, false); f.offr = r.off; f.csr = r.cs; f.psr = r.ps; }
    else { f.offr = -1; f.csr = 0; f.psr = 0; }
    f.ntiles = pc.coutp / 16; f.act = op.act;
    f.w_chunks = ntaps * pc.kg0_32 * 3 * 64;
    f.w_lds = 0; f.w_slot = 0; f.w_after_barrier = 0;
    if (f.kg0 * 48 > f.cs0) return no("internal: K groups beyond the source rows");
    if (dst.ps != 0 && (f.ntiles + 1) / 2 * 48 > dst.cs) return no("internal: output tiles beyond the destination rows");
  }
  pg.solo_at = solo ? n_trunk : n_ops; pg.solo_pol = 2;
  pg.solo_zero_off = off_i + (int)off_hid + rows * cs_hid; pg.solo_zero_len = solo ? cs_hid : 0;
  Fused16Program* dev = nullptr;
  if (hipMalloc((void**)&dev, sizeof(pg)) != hipSuccess) return no("device allocation failed");
  if (hipMemcpy(dev, &pg, sizeof(pg), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return no("upload failed"); }
  h->wave_dev = dev;
  nz::WaveNet& w = h->wave;
  w.prog = dev; w.lds_floats = pg.lds_floats; w.stage_off = off_t; w.stage_floats = (int32_t)((stage + 3) / 4 * 4);
  w.inp = h->inp; w.in_channels = nd.in_channels;
  w.hw = h->hw; w.rows = h->rows; w.cols = h->cols; w.planes = nd.policy_channels; w.hex = nd.hex ? 1 : 0; w.n_ops = n_ops;
  w.flops = h->flops;
  w.mfmas = 0;
  for (int i = 0; i < n_ops; ++i) w.mfmas += (int64_t)2 * pg.ops[i].ntiles * ntaps * pg.ops[i].kg0 * 6;
  h->wave_state = 1;
  *out = w;
  return true;
}

extern "C" {

const char* nz_boardnet_last_error(const nz_boardnet* h) { return h ? h->error.c_str() : g_err.c_str(); }

void nz_boardnet_destroy(nz_boardnet* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (float* b : h->buffers) (void)hipFree(b);
  for (auto& c : h->convs) { (void)hipFree(c.dev); if (c.dev_split) (void)hipFree(c.dev_split); if (c.dev16) (void)hipFree(c.dev16); }
  if (h->fused_dev) (void)hipFree(h->fused_dev);
  if (h->fused16_dev) (void)hipFree(h->fused16_dev);
  if (h->wave_dev) (void)hipFree(h->wave_dev);
  delete h;
}

nz_status nz_boardnet_create(nz_boardnet** out, const nz_net_desc* net, int32_t rows, int32_t cols,
                             int32_t max_batch, int32_t device) {
  if (!out || !net) return bfail(nullptr, NZ_ERR_ARG, "null argument");
  *out = nullptr;
  if (rows <= 0 || cols <= 0 || max_batch <= 0 || net->in_channels <= 0 || net->policy_channels <= 0 ||
      net->width <= 0 || net->num_blocks < 0)
    return bfail(nullptr, NZ_ERR_ARG, "bad sizes");
  if (net->arch != NZ_ARCH_RECURRENT && net->arch != NZ_ARCH_RESNET && net->arch != NZ_ARCH_CONVNET)
    return bfail(nullptr, NZ_ERR_ARG, "unknown architecture %d", net->arch);
  if (net->arch == NZ_ARCH_CONVNET && net->kernel_size != 1 && net->kernel_size != 3)
    return bfail(nullptr, NZ_ERR_ARG, "ConvNet kernel_size must be 1 or 3");
  if (net->hex && net->arch == NZ_ARCH_CONVNET && net->kernel_size != 1)
    return bfail(nullptr, NZ_ERR_ARG, "hexagonal ConvNet: only kernel_size 1 (centre + 6 neighbours) is built");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return bfail(nullptr, NZ_ERR_HIP, "no HIP device %d (no CPU fallback)", device);
  (void)hipSetDevice(device);
  nz_boardnet* h = new nz_boardnet;
  h->device = device; h->net = *net; h->rows = rows; h->cols = cols; h->hw = rows * cols; h->max_batch = max_batch;
  h->inp = pad16(net->in_channels);
  h->widthp = pad16(net->width);
  const std::vector<int> pc = head_channels(net->width, net->policy_channels, 2);
  const std::vector<int> vc = head_channels(net->width, 1, 4);
  int vmax = 16;
  for (size_t i = 1; i < vc.size(); ++i) vmax = std::max(vmax, pad16(vc[i]));
  h->buffer_channels = {h->inp, h->widthp, h->widthp, h->widthp, pad16(pc[1]), pad16(pc[2]), vmax, vmax};
  for (int c : h->buffer_channels) {
    float* b = nullptr;
    const size_t n = (size_t)((max_batch + 15) / 16 * 16) * h->hw * c;
    if (hipMalloc((void**)&b, n * sizeof(float)) != hipSuccess || hipMemset(b, 0, n * sizeof(float)) != hipSuccess) {
      nz_boardnet_destroy(h);
      return bfail(nullptr, NZ_ERR_HIP, "device allocation failed");
    }
    h->buffers.push_back(b);
  }
  *out = h;
  return NZ_OK;
}

// weights: the reference's state_dict tensors in order (device or host pointers), float32.
nz_status nz_boardnet_set_weights(nz_boardnet* h, const float* const* weights, int32_t n_weights,
                                  int32_t recurrent_iterations) {
  if (!h || !weights) return NZ_ERR_ARG;
  B_HIP(h, hipSetDevice(h->device));
  B_HIP(h, hipDeviceSynchronize());
  for (auto& c : h->convs) { (void)hipFree(c.dev); if (c.dev_split) (void)hipFree(c.dev_split); if (c.dev16) (void)hipFree(c.dev16); }
  h->convs.clear(); h->ops.clear(); h->flops = 0; h->ready = false;
  if (h->fused_dev) { (void)hipFree(h->fused_dev); h->fused_dev = nullptr; }
  if (h->fused16_dev) { (void)hipFree(h->fused16_dev); h->fused16_dev = nullptr; }
  if (h->wave_dev) { (void)hipFree(h->wave_dev); h->wave_dev = nullptr; }
  h->wave_state = 0;
  const nz_net_desc& nd = h->net;
  const int W = nd.width, Wp = h->widthp, IN = nd.in_channels, INp = h->inp;
  const int vact = nd.value_activation == NZ_ACT_RELU ? 1 : 2;
  int wi = 0;
  const int per_conv = nd.hex ? 2 : 1;           // hexagdly.Conv2d holds kernel0 and kernel1
  auto add = [&](int cout, int c0, int c1, int k, int c0p, int c1p, int src0, int src1, int res, int dst, int act) {
    if (wi + per_conv > n_weights) return false;
    if (!pack(h, weights[wi], nd.hex ? weights[wi + 1] : nullptr, cout, c0, c1, k, c0p, c1p)) return false;
    h->ops.push_back(ConvOp{src0, src1, res, dst, wi, act});
    wi += per_conv;
    return true;
  };
  bool ok = true;
  int cur = 1;                                  // trunk buffer holding the current activations
  auto block = [&](int w0) {                    // BasicBlock: relu(conv(relu(conv(t))) + t)
    const int y = cur % 3 + 1, o = y % 3 + 1;
    (void)w0;
    ok = ok && add(W, W, 0, 3, Wp, 0, cur, -1, -1, y, 1) && add(W, W, 0, 3, Wp, 0, y, -1, cur, o, 1);
    cur = o;
  };
  if (nd.arch == NZ_ARCH_RECURRENT) {
    const int per_iter = (nd.recall ? 1 : 0) + 2 * nd.num_blocks;
    if (n_weights != (1 + per_iter + 6) * per_conv) return bfail(h, NZ_ERR_ARG, "RecurrentNet needs %d tensors, got %d", (1 + per_iter + 6) * per_conv, n_weights);
    if (recurrent_iterations < 1) return bfail(h, NZ_ERR_ARG, "recurrent_iterations must be >= 1");
    ok = add(W, IN, 0, 3, INp, 0, 0, -1, -1, 1, 1);
    const int first = wi;
    for (int it = 0; it < recurrent_iterations && ok; ++it) {
      wi = first;
      if (nd.recall) {
        const int o = cur % 3 + 1;
        ok = ok && add(W, W, IN, 3, Wp, INp, cur, 0, -1, o, 0);
        cur = o;
      }
      for (int b = 0; b < nd.num_blocks && ok; ++b) block(wi);
    }
    wi = first + per_iter * per_conv;
  } else if (nd.arch == NZ_ARCH_RESNET) {
    if (n_weights != (1 + 2 * nd.num_blocks + 6) * per_conv) return bfail(h, NZ_ERR_ARG, "ResNet needs %d tensors, got %d", (1 + 2 * nd.num_blocks + 6) * per_conv, n_weights);
    ok = add(W, IN, 0, 3, INp, 0, 0, -1, -1, 1, 1);
    for (int b = 0; b < nd.num_blocks && ok; ++b) block(wi);
  } else {
    if (n_weights != (1 + nd.num_blocks + 6) * per_conv) return bfail(h, NZ_ERR_ARG, "ConvNet needs %d tensors, got %d", (1 + nd.num_blocks + 6) * per_conv, n_weights);
    const int k = nd.kernel_size;
    ok = add(W, IN, 0, k, INp, 0, 0, -1, -1, 1, 3);
    for (int i = 0; i < nd.num_blocks && ok; ++i) {
      const int o = cur % 3 + 1;
      ok = ok && add(W, W, 0, k, Wp, 0, cur, -1, -1, o, 3);
      cur = o;
    }
  }
  const std::vector<int> pc = head_channels(W, nd.policy_channels, 2);
  const std::vector<int> vc = head_channels(W, 1, 4);
  for (size_t i = 1; i < pc.size(); ++i)
    if (pc[i] <= 0) return bfail(h, NZ_ERR_ARG, "policy head channel schedule reaches %d", pc[i]);
  ok = ok && add(pc[1], W, 0, 3, Wp, 0, cur, -1, -1, 4, 1) && add(pc[2], pc[1], 0, 3, pad16(pc[1]), 0, 4, -1, -1, 5, 0);
  int vsrc = cur;
  for (int i = 0; i < 4 && ok; ++i) {
    if (vc[i + 1] <= 0) return bfail(h, NZ_ERR_ARG, "value head channel schedule reaches %d", vc[i + 1]);
    const int dst = 6 + (i & 1);
    ok = add(vc[i + 1], vc[i], 0, 3, pad16(vc[i]), 0, vsrc, -1, -1, dst, i == 3 ? 0 : vact);
    vsrc = dst;
  }
  if (!ok) return bfail(h, NZ_ERR_HIP, "weight upload failed");
  h->policy_buf = 5; h->value_buf = vsrc;
  build_fused(h, cur);
  build_fused16(h);
  h->ready = true;
  return NZ_OK;
}

int64_t nz_boardnet_flops(const nz_boardnet* h) { return h ? h->flops : 0; }

// Diagnostic (-DNZ_WIDE_STAMPS builds only; NZ_ERR_STATE otherwise): conv_wide_kernel's phase ticks summed over its launches
// so far: [0] loads issued, [1] first fragments, [2] MFMAs + staging, [3] barrier, [4] loop overhead, [5] K steps, [6] launches.
nz_status nz_boardnet_wide_stamps(uint64_t* out8) {
#ifdef NZ_WIDE_STAMPS
  if (!out8) return NZ_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return NZ_ERR_HIP;
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wide_stamps), sizeof(h)) != hipSuccess) return NZ_ERR_HIP;
  for (int i = 0; i < 8; ++i) out8[i] = h[i];
  return NZ_OK;
#else
  (void)out8;
  return NZ_ERR_STATE;
#endif
}

nz_status nz_boardnet_fused(nz_boardnet* h, int32_t enable, int32_t* available) {
  if (!h) return NZ_ERR_ARG;
  if (enable >= 0) h->use_fused = enable != 0;
  if (available) *available = h->fused_dev != nullptr || h->fused16_dev != nullptr ? 1 : 0;
  return NZ_OK;
}

nz_status nz_boardnet_dims(const nz_boardnet* h, int32_t* in_channels, int32_t* policy_channels, int32_t* rows,
                           int32_t* cols, int32_t* max_batch) {
  if (!h) return NZ_ERR_ARG;
  if (in_channels) *in_channels = h->net.in_channels;
  if (policy_channels) *policy_channels = h->net.policy_channels;
  if (rows) *rows = h->rows;
  if (cols) *cols = h->cols;
  if (max_batch) *max_batch = h->max_batch;
  return NZ_OK;
}

static nz_status forward_impl(nz_boardnet* h, const float* images_dev, int32_t n, const int32_t* n_dev,
                               float* logits_dev, float* probs_dev, float* value_dev, void* stream) {
  if (!h || !value_dev) return NZ_ERR_ARG;
  if (!h->ready) return bfail(h, NZ_ERR_STATE, "no weights set");
  if (n < 0 || n > h->max_batch) return bfail(h, NZ_ERR_ARG, "batch %d exceeds max_batch %d", n, h->max_batch);
  if (n == 0) return NZ_OK;
  B_HIP(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)((n + 15) / 16 * 16) * h->hw * h->inp;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  if (images_dev != nullptr)        // else: the caller filled the input rows itself (nz_boardnet_input_rows)
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3(blocks), dim3(256), 0, s, images_dev, h->buffers[0], n_dev, n,
                       h->net.in_channels, h->inp, h->hw);
  if (h->fused16_dev != nullptr && h->use_fused) {  // one launch, on the BF16 matrix cores (three-way split arithmetic)
    if (h->net.hex)
      hipLaunchKernelGGL(fused16_net_kernel<true>, dim3(h->fused16_grid), dim3(FUSED_THREADS), h->fused16_lds_bytes, s,
                         h->fused16_dev, h->buffers[0], h->inp, n_dev, n, logits_dev, probs_dev, value_dev);
    else
      hipLaunchKernelGGL(fused16_net_kernel<false>, dim3(h->fused16_grid), dim3(FUSED_THREADS), h->fused16_lds_bytes, s,
                         h->fused16_dev, h->buffers[0], h->inp, n_dev, n, logits_dev, probs_dev, value_dev);
    B_HIP(h, hipGetLastError());
    return NZ_OK;
  }
  if (h->fused_dev != nullptr && h->use_fused) {    // every layer, the softmax and the value in one launch (activations in LDS)
    if (h->net.hex)
      hipLaunchKernelGGL(fused_net_kernel<true>, dim3(h->fused_grid), dim3(FUSED_THREADS), h->fused_lds_bytes, s, h->fused_dev,
                         h->buffers[0], h->inp, n_dev, n, logits_dev, probs_dev, value_dev);
    else
      hipLaunchKernelGGL(fused_net_kernel<false>, dim3(h->fused_grid), dim3(FUSED_THREADS), h->fused_lds_bytes, s, h->fused_dev,
                         h->buffers[0], h->inp, n_dev, n, logits_dev, probs_dev, value_dev);
    B_HIP(h, hipGetLastError());
    return NZ_OK;
  }
  for (const ConvOp& op : h->ops) {
    const PackedConv& pc = h->convs[&op - h->ops.data()];
    ConvArgs a;
    a.src0 = h->buffers[op.src0];
    a.src1 = op.src1 >= 0 ? h->buffers[op.src1] : nullptr;
    a.w = pc.dev;
    a.ws = pc.dev_split;
    a.res = op.res >= 0 ? h->buffers[op.res] : nullptr;
    a.dst = h->buffers[op.dst];
    a.n_dev = n_dev; a.n_host = n; a.hw = h->hw; a.h = h->rows; a.wd = h->cols;
    a.c0 = pc.c0p; a.c1 = pc.c1p;
    a.s0 = h->buffer_channels[op.src0];
    a.s1 = op.src1 >= 0 ? h->buffer_channels[op.src1] : 0;
    a.cd = h->buffer_channels[op.dst];
    a.act = op.act;
    a.hex = h->net.hex ? 1 : 0;
    if (a.c0 > a.s0 || a.c1 > a.s1 || pc.coutp > a.cd) return bfail(h, NZ_ERR_STATE, "internal: layer shapes disagree");
    dispatch_conv(a, pc.coutp / 16, s);
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(n), dim3(64), 0, s, h->buffers[h->policy_buf], h->buffer_channels[h->policy_buf],
                     h->net.policy_channels, h->buffers[h->value_buf], h->buffer_channels[h->value_buf], h->hw, n_dev, n,
                     logits_dev, probs_dev, value_dev);
  B_HIP(h, hipGetLastError());
  return NZ_OK;
}

nz_status nz_boardnet_forward(nz_boardnet* h, const float* images_dev, int32_t n, const int32_t* n_dev,
                              float* logits_dev, float* probs_dev, float* value_dev, void* stream) {
  if (!images_dev) return NZ_ERR_ARG;
  return forward_impl(h, images_dev, n, n_dev, logits_dev, probs_dev, value_dev, stream);
}

nz_status nz_boardnet_input_rows(nz_boardnet* h, float** rows_dev, int32_t* row_stride) {
  if (!h || !rows_dev || !row_stride) return NZ_ERR_ARG;
  *rows_dev = h->buffers[0];
  *row_stride = h->inp;
  return NZ_OK;
}

nz_status nz_boardnet_forward_rows(nz_boardnet* h, int32_t n, const int32_t* n_dev, float* logits_dev, float* probs_dev,
                                   float* value_dev, void* stream) {
  return forward_impl(h, nullptr, n, n_dev, logits_dev, probs_dev, value_dev, stream);
}

}  // extern "C"
