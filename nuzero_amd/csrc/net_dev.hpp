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
// Work split.  The host compiles the network into one job list per wave
// (engine.hip build_program): a job is one (layer, 16-channel output tile,
// output-cell group).  64-channel layers give each of the 4 waves one full
// tile; the narrow head layers are cut by output-cell group as well so that all
// four matrix pipes stay busy, and the policy and value heads run side by side.
// Per 16-channel K group a wave reads at most 9 activation vectors (one
// ds_read_b128 per input cell) and 9 weight vectors (one global_load_dwordx4 per
// tap) and issues up to 196 MFMAs from them; the weights of the NEXT K group --
// also across jobs and barriers -- are already in flight while it does so.
//
// LDS layout: act[buf][cell][pos][64 ch], the 16-byte slot index XOR-ed with
// the position so that ds_read_b128 (A operands) and ds_write_b32 (epilogue) are
// bank-conflict free.
#pragma once
#include <type_traits>

#include "engine.h"

namespace nz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int POS = 16;            // positions per workgroup (MFMA M)
constexpr int CELLS = 9;
constexpr int ROW = 64;            // channels per (cell, pos) row
constexpr int NET_WAVES = 4;
constexpr int NET_THREADS = NET_WAVES * 64;
constexpr int NET_BUFFERS = 3;
constexpr int ACT_FLOATS = CELLS * POS * ROW;
constexpr int INP_FLOATS = CELLS * POS * 4;
constexpr int NET_LDS_FLOATS = NET_BUFFERS * ACT_FLOATS + INP_FLOATS;
constexpr int W_KG_FLOATS = NET_KG_DWORDS; // one K group of one n-tile: [tap][lane][4] (FP32) / [tap][piece][lane][8 bf16]
static_assert(NET_WAVES == NET_WAVES_HOST, "job lists are per wave");

// output-cell groups: all nine | four quarters {4,0} {1,3} {5,7} {2,6,8} (taps: 49 | 13, 12, 12, 12)
__host__ __device__ constexpr int og_mask(int og) {
  return og == 0 ? 0x1FF : og == 1 ? 0x011 : og == 2 ? 0x00A : og == 3 ? 0x0A0 : og == 4 ? 0x144 : 0;
}

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
template <int OMASK, int I, int TAP>
constexpr bool pair_used() { return TapMap<I, TAP>::valid && ((OMASK >> TapMap<I, TAP>::o) & 1); }
template <int OMASK, int I>
constexpr bool input_used() {
  return pair_used<OMASK, I, 0>() || pair_used<OMASK, I, 1>() || pair_used<OMASK, I, 2>() ||
         pair_used<OMASK, I, 3>() || pair_used<OMASK, I, 4>() || pair_used<OMASK, I, 5>() ||
         pair_used<OMASK, I, 6>() || pair_used<OMASK, I, 7>() || pair_used<OMASK, I, 8>();
}

struct Frag {          // operands of one 16-channel K group
  f32x4 a[CELLS];      // activations per input cell: 4 consecutive channels of this lane's K slice
  f32x4 b[9];          // weights per tap
};

template <int OMASK, int I, int TAP>
__device__ __forceinline__ void mfma_pair(f32x4 (&acc)[CELLS], const Frag& f) {
  if constexpr (pair_used<OMASK, I, TAP>()) {
    constexpr int o = TapMap<I, TAP>::o;
    // the four K sub-steps of one (input cell, tap) pair back to back: a single accumulator
    // chain issues at the full rate (scripts/microbench/mfma_rate.hip)
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[I][0], f.b[TAP][0], acc[o], 0, 0, 0);
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[I][1], f.b[TAP][1], acc[o], 0, 0, 0);
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[I][2], f.b[TAP][2], acc[o], 0, 0, 0);
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[I][3], f.b[TAP][3], acc[o], 0, 0, 0);
  }
}
// input-cell major: the MFMAs of cell I only need the I-th LDS read to have landed
template <int OMASK, int I>
__device__ __forceinline__ void mfma_cell(f32x4 (&acc)[CELLS], const Frag& f) {
  mfma_pair<OMASK, I, 0>(acc, f); mfma_pair<OMASK, I, 1>(acc, f); mfma_pair<OMASK, I, 2>(acc, f);
  mfma_pair<OMASK, I, 3>(acc, f); mfma_pair<OMASK, I, 4>(acc, f); mfma_pair<OMASK, I, 5>(acc, f);
  mfma_pair<OMASK, I, 6>(acc, f); mfma_pair<OMASK, I, 7>(acc, f); mfma_pair<OMASK, I, 8>(acc, f);
}
template <int OMASK>
__device__ __forceinline__ void mfma_group(f32x4 (&acc)[CELLS], const Frag& f) {
  mfma_cell<OMASK, 0>(acc, f); mfma_cell<OMASK, 1>(acc, f); mfma_cell<OMASK, 2>(acc, f);
  mfma_cell<OMASK, 3>(acc, f); mfma_cell<OMASK, 4>(acc, f); mfma_cell<OMASK, 5>(acc, f);
  mfma_cell<OMASK, 6>(acc, f); mfma_cell<OMASK, 7>(acc, f); mfma_cell<OMASK, 8>(acc, f);
}

template <int OMASK, int I>
__device__ __forceinline__ void load_a1(Frag& f, const float* __restrict__ src, int a0) {
#ifdef NZ_ABLATE_A   // timing-only build: no LDS reads of the A operands (outputs are wrong)
  if constexpr (input_used<OMASK, I>()) { const float v = (float)(a0 + I); f.a[I] = f32x4{v, v + 1.f, v + 2.f, v + 3.f}; asm volatile("" :: "v"(src)); }
#else
  if constexpr (input_used<OMASK, I>()) f.a[I] = *reinterpret_cast<const f32x4*>(src + a0 + I * (POS * ROW));
#endif
}
// this lane's activation operands of K group kg (address of cell 0; cell I is I*1024 floats further)
template <int OMASK>
__device__ __forceinline__ void load_a(Frag& f, const float* __restrict__ src, int pos, int quad, int kg) {
  const int a0 = act_addr(0, pos, kg * 16 + quad * 4);
  load_a1<OMASK, 0>(f, src, a0); load_a1<OMASK, 1>(f, src, a0); load_a1<OMASK, 2>(f, src, a0);
  load_a1<OMASK, 3>(f, src, a0); load_a1<OMASK, 4>(f, src, a0); load_a1<OMASK, 5>(f, src, a0);
  load_a1<OMASK, 6>(f, src, a0); load_a1<OMASK, 7>(f, src, a0); load_a1<OMASK, 8>(f, src, a0);
}
__device__ __forceinline__ void load_b(Frag& f, const float* __restrict__ w, int lane) {
#ifdef NZ_ABLATE_B   // timing-only build: no global loads of the B operands (outputs are wrong)
  const float v = (float)lane * 1e-3f;
#pragma unroll
  for (int t = 0; t < 9; ++t) f.b[t] = f32x4{v, v + (float)t, v, v};
  asm volatile("" :: "v"(w));
#else
  const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(w) + lane;
#pragma unroll
  for (int t = 0; t < 9; ++t) f.b[t] = p[t * 64];
#endif
}

// All K groups of one job.  On entry f0.b holds the weights of the job's first K group; on
// exit it holds those of the next job's first K group (`w_after`, may be null): the weight
// stream is always one K group ahead, also across jobs and stage barriers.  Unrolled by two
// so that the two operand sets swap roles without register copies.
template <int OMASK>
__device__ __forceinline__ void job_kloop(f32x4 (&acc)[CELLS], Frag& f0, Frag& f1, const float* __restrict__ src,
                                          const float* __restrict__ w, int kgroups,
                                          const float* __restrict__ w_after, int lane) {
  const int pos = lane & 15, quad = lane >> 4;
  int kg = 0;
  for (; kg + 2 <= kgroups; kg += 2) {
    load_b(f1, w + (kg + 1) * W_KG_FLOATS, lane);
    load_a<OMASK>(f0, src, pos, quad, kg);
    load_a<OMASK>(f1, src, pos, quad, kg + 1);
    mfma_group<OMASK>(acc, f0);
    const float* w_nxt = (kg + 2 < kgroups) ? w + (kg + 2) * W_KG_FLOATS : w_after;
    if (w_nxt != nullptr) load_b(f0, w_nxt, lane);
    mfma_group<OMASK>(acc, f1);
  }
  if (kg < kgroups) {                 // odd tail
    if (w_after != nullptr) load_b(f1, w_after, lane);
    load_a<OMASK>(f0, src, pos, quad, kg);
    mfma_group<OMASK>(acc, f0);
#pragma unroll
    for (int t = 0; t < 9; ++t) f0.b[t] = f1.b[t];
  }
}

// ---- split form: float32 products from six bf16 MFMAs (engine.h NET_SPLIT) ----------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TAP_DWORDS = 3 * 64 * 4;      // one tap of one K group: [piece][lane][8 bf16]
constexpr int W_RING = 3;                   // taps in flight; 9 taps per K group keep the slots aligned
struct TapB {             // one tap's weights: the three pieces of this lane's 8 channels
  u32x4 p[3];
};
struct FragS {            // the wavefront's weight stream, W_RING taps ahead of the MFMAs
  TapB rb[W_RING];
};
struct SplitA {           // one input cell's activations, this lane's 8 channels, as three bf16 pieces
  u32x4 p[3];
};

__device__ __forceinline__ uint32_t trunc_bf16(float x) { return __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u; }
// (x0, x1) -> the two upper halves in one register: bf16(x0) | bf16(x1) << 16 (truncating)
__device__ __forceinline__ uint32_t pack_hi16(float x0, float x1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, x0), 0x07060302u);
}
// exact: x = p0 + p1 + p2 with 8 significant bits each; element j of the result = pair j's word
template <int J>
__device__ __forceinline__ void split_pair(float x0, float x1, SplitA& out) {
  out.p[0][J] = pack_hi16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, trunc_bf16(x0));
  const float r1 = x1 - __builtin_bit_cast(float, trunc_bf16(x1));
  out.p[1][J] = pack_hi16(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, trunc_bf16(r0));
  const float s1 = r1 - __builtin_bit_cast(float, trunc_bf16(r1));
  out.p[2][J] = pack_hi16(s0, s1);
}
__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, SplitA& out) {
  split_pair<0>(lo[0], lo[1], out);
  split_pair<1>(lo[2], lo[3], out);
  split_pair<2>(hi[0], hi[1], out);
  split_pair<3>(hi[2], hi[3], out);
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int OMASK, int I, int TAP>
__device__ __forceinline__ void split_pair_mfma(f32x4 (&acc)[CELLS], const SplitA (&a)[CELLS], const TapB& b) {
  if constexpr (pair_used<OMASK, I, TAP>()) {
    constexpr int o = TapMap<I, TAP>::o;
    // small terms first; one accumulator chain issues at the full rate
    acc[o] = mfma_bf16(a[I].p[1], b.p[1], acc[o]);
    acc[o] = mfma_bf16(a[I].p[2], b.p[0], acc[o]);
    acc[o] = mfma_bf16(a[I].p[0], b.p[2], acc[o]);
    acc[o] = mfma_bf16(a[I].p[1], b.p[0], acc[o]);
    acc[o] = mfma_bf16(a[I].p[0], b.p[1], acc[o]);
    acc[o] = mfma_bf16(a[I].p[0], b.p[0], acc[o]);
  }
}
// all (input cell, TAP) pairs of the job's output cells: independent accumulators
template <int OMASK, int TAP>
__device__ __forceinline__ void split_tap(f32x4 (&acc)[CELLS], const SplitA (&a)[CELLS], const TapB& b) {
  split_pair_mfma<OMASK, 0, TAP>(acc, a, b); split_pair_mfma<OMASK, 1, TAP>(acc, a, b); split_pair_mfma<OMASK, 2, TAP>(acc, a, b);
  split_pair_mfma<OMASK, 3, TAP>(acc, a, b); split_pair_mfma<OMASK, 4, TAP>(acc, a, b); split_pair_mfma<OMASK, 5, TAP>(acc, a, b);
  split_pair_mfma<OMASK, 6, TAP>(acc, a, b); split_pair_mfma<OMASK, 7, TAP>(acc, a, b); split_pair_mfma<OMASK, 8, TAP>(acc, a, b);
}
template <int OMASK, int I>
__device__ __forceinline__ void split_cell(SplitA (&a)[CELLS], const float* __restrict__ src, int a0, int a1) {
  if constexpr (input_used<OMASK, I>()) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src + a0 + I * (POS * ROW));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(src + a1 + I * (POS * ROW));
    split8(lo, hi, a[I]);
  }
}
__device__ __forceinline__ void load_tap(TapB& t, const float* __restrict__ w, int lane) {
  const u32x4* __restrict__ p = reinterpret_cast<const u32x4*>(w) + lane;
  t.p[0] = p[0];
  t.p[1] = p[64];
  t.p[2] = p[128];
}
__device__ __forceinline__ void load_b(FragS& f, const float* __restrict__ w, int lane) {   // the first W_RING taps
#pragma unroll
  for (int i = 0; i < W_RING; ++i) load_tap(f.rb[i], w + i * TAP_DWORDS, lane);
}
// All K groups (32 channels each) of one job, tap-major.  On entry the ring holds the job's first
// W_RING taps; on exit those of the next job that reads weights (`w_after`, may be null): the
// weight stream stays W_RING taps ahead, also across jobs and stage barriers.  Per K group the
// lane reads its 8 channels of every used input cell from LDS (2 x ds_read_b128) and splits them.
template <int OMASK>
__device__ __forceinline__ void job_kloop(f32x4 (&acc)[CELLS], FragS& f, FragS&, const float* __restrict__ src,
                                          const float* __restrict__ w, int kgroups,
                                          const float* __restrict__ w_after, int lane) {
  const int pos = lane & 15, quad = lane >> 4;
  const int n_taps = kgroups * 9;
  auto refill = [&](TapB& slot, int t) {          // t = stream index of the tap to fetch
    if (t < n_taps) load_tap(slot, w + t * TAP_DWORDS, lane);
    else if (w_after != nullptr) load_tap(slot, w_after + (t - n_taps) * TAP_DWORDS, lane);
  };
  for (int kg = 0; kg < kgroups; ++kg) {
    const int a0 = act_addr(0, pos, kg * 32 + quad * 8), a1 = act_addr(0, pos, kg * 32 + quad * 8 + 4);
    SplitA a[CELLS];
    split_cell<OMASK, 0>(a, src, a0, a1); split_cell<OMASK, 1>(a, src, a0, a1); split_cell<OMASK, 2>(a, src, a0, a1);
    split_cell<OMASK, 3>(a, src, a0, a1); split_cell<OMASK, 4>(a, src, a0, a1); split_cell<OMASK, 5>(a, src, a0, a1);
    split_cell<OMASK, 6>(a, src, a0, a1); split_cell<OMASK, 7>(a, src, a0, a1); split_cell<OMASK, 8>(a, src, a0, a1);
    const int t0 = kg * 9 + W_RING;
    split_tap<OMASK, 0>(acc, a, f.rb[0]); refill(f.rb[0], t0 + 0);
    split_tap<OMASK, 1>(acc, a, f.rb[1]); refill(f.rb[1], t0 + 1);
    split_tap<OMASK, 2>(acc, a, f.rb[2]); refill(f.rb[2], t0 + 2);
    split_tap<OMASK, 3>(acc, a, f.rb[0]); refill(f.rb[0], t0 + 3);
    split_tap<OMASK, 4>(acc, a, f.rb[1]); refill(f.rb[1], t0 + 4);
    split_tap<OMASK, 5>(acc, a, f.rb[2]); refill(f.rb[2], t0 + 5);
    split_tap<OMASK, 6>(acc, a, f.rb[0]); refill(f.rb[0], t0 + 6);
    split_tap<OMASK, 7>(acc, a, f.rb[1]); refill(f.rb[1], t0 + 7);
    split_tap<OMASK, 8>(acc, a, f.rb[2]); refill(f.rb[2], t0 + 8);
  }
}

// tanh for the value head: 1 - 2 / (exp(2x) + 1) on the hardware exp; absolute error < 3e-7
// (the parity tolerance on values is 1e-5), exact limits at +-inf
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

template <int OMASK, int TAP, int I>
__device__ __forceinline__ void mfma_extra_pair(f32x4 (&acc)[CELLS], const float (&ax)[CELLS], const float (&bx)[9]) {
  if constexpr (pair_used<OMASK, I, TAP>()) {
    constexpr int o = TapMap<I, TAP>::o;
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[I], bx[TAP], acc[o], 0, 0, 0);
  }
}
template <int OMASK, int TAP>
__device__ __forceinline__ void mfma_extra_tap(f32x4 (&acc)[CELLS], const float (&ax)[CELLS], const float (&bx)[9]) {
  mfma_extra_pair<OMASK, TAP, 0>(acc, ax, bx); mfma_extra_pair<OMASK, TAP, 1>(acc, ax, bx);
  mfma_extra_pair<OMASK, TAP, 2>(acc, ax, bx); mfma_extra_pair<OMASK, TAP, 3>(acc, ax, bx);
  mfma_extra_pair<OMASK, TAP, 4>(acc, ax, bx); mfma_extra_pair<OMASK, TAP, 5>(acc, ax, bx);
  mfma_extra_pair<OMASK, TAP, 6>(acc, ax, bx); mfma_extra_pair<OMASK, TAP, 7>(acc, ax, bx);
  mfma_extra_pair<OMASK, TAP, 8>(acc, ax, bx);
}
// the (<= 4) raw input planes as one extra K step (projection / recall conv)
template <int OMASK>
__device__ __forceinline__ void extra_planes(f32x4 (&acc)[CELLS], const float* __restrict__ wx,
                                             const float* __restrict__ inp, int lane) {
  float ax[CELLS], bx[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) bx[t] = wx[t * 64 + lane];
#pragma unroll
  for (int i = 0; i < CELLS; ++i) ax[i] = inp[(i * POS + (lane & 15)) * 4 + (lane >> 4)];
  mfma_extra_tap<OMASK, 0>(acc, ax, bx); mfma_extra_tap<OMASK, 1>(acc, ax, bx); mfma_extra_tap<OMASK, 2>(acc, ax, bx);
  mfma_extra_tap<OMASK, 3>(acc, ax, bx); mfma_extra_tap<OMASK, 4>(acc, ax, bx); mfma_extra_tap<OMASK, 5>(acc, ax, bx);
  mfma_extra_tap<OMASK, 6>(acc, ax, bx); mfma_extra_tap<OMASK, 7>(acc, ax, bx); mfma_extra_tap<OMASK, 8>(acc, ax, bx);
}

// ---- epilogues: lane holds C[pos = 4*quad + r][cout = 16*nt + (lane & 15)] for the job's cells ----
// ACT: 0 none, 1 relu, 2 tanh.  Output cell o lives o*1024 floats after cell 0, so the four
// per-lane offsets are computed once and every access is base + constant.
template <int OMASK, int ACT, bool RES>
__device__ __forceinline__ void epilogue_lds(const f32x4 (&acc)[CELLS], float* __restrict__ dst,
                                             const float* res, int lane, int nt) {
  const int quad = lane >> 4, cout = nt * 16 + (lane & 15);
  int off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) off[r] = act_addr(0, quad * 4 + r, cout);
  float v[CELLS][4];
#pragma unroll
  for (int o = 0; o < CELLS; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!((OMASK >> o) & 1)) continue;
      v[o][r] = acc[o][r];
      if constexpr (RES) v[o][r] += res[o * (POS * ROW) + off[r]];   // read before any write below
    }
#pragma unroll
  for (int o = 0; o < CELLS; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!((OMASK >> o) & 1)) continue;
      float x = v[o][r];
      if constexpr (ACT == 1) x = fmaxf(x, 0.0f);
      if constexpr (ACT == 2) x = tanh_fast(x);
      if constexpr (ACT == 3) x = x > 0.0f ? x : expm1f(x);      // nn.ELU(), alpha = 1
      dst[o * (POS * ROW) + off[r]] = x;
    }
}
template <int OMASK>
__device__ __forceinline__ void epilogue(const f32x4 (&acc)[CELLS], const NetJob& job, float* __restrict__ lds,
                                         int lane, int policy_channels, int n_valid, float* logits, float* value) {
  const int quad = lane >> 4;
  const int cout = job.nt * 16 + (lane & 15);
  if (job.dst < NET_BUFFERS) {
    float* dst = lds + job.dst * ACT_FLOATS;
    if (job.res >= 0) {
      if constexpr (OMASK == 0x1FF) epilogue_lds<OMASK, 1, true>(acc, dst, lds + job.res * ACT_FLOATS, lane, job.nt);
    } else if (job.act == 1) {
      epilogue_lds<OMASK, 1, false>(acc, dst, nullptr, lane, job.nt);
    } else if (job.act == 2) {
      epilogue_lds<OMASK, 2, false>(acc, dst, nullptr, lane, job.nt);
    } else if (job.act == 3) {
      if constexpr (OMASK == 0x1FF) epilogue_lds<OMASK, 3, false>(acc, dst, nullptr, lane, job.nt);
    } else {
      epilogue_lds<OMASK, 0, false>(acc, dst, nullptr, lane, job.nt);
    }
  } else if (job.dst == NET_BUFFERS) {        // policy logits [pos][P][9]
    if (cout < policy_channels) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gp = quad * 4 + r;
        if (gp < n_valid) {
#pragma unroll
          for (int o = 0; o < CELLS; ++o)
            if ((OMASK >> o) & 1) logits[((size_t)gp * policy_channels + cout) * CELLS + o] = acc[o][r];
        }
      }
    }
  } else {                                     // value: mean over (C=1,H,W), tanh (blocks.py:82-84)
    if constexpr (OMASK == 0x1FF) {
      if (cout == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gp = quad * 4 + r;
          if (gp < n_valid) {
            float s = 0.0f;
#pragma unroll
            for (int o = 0; o < CELLS; ++o) s += acc[o][r];
            value[gp] = tanh_fast(s / 9.0f);
          }
        }
      }
    }
  }
}

// one job: K loop, input-plane step, epilogue
using NetOperands = typename std::conditional<NET_SPLIT, FragS, Frag>::type;
__device__ __forceinline__ void zero_b(Frag& f) {
#pragma unroll
  for (int t = 0; t < 9; ++t) f.b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void zero_b(FragS& f) {
#pragma unroll
  for (int t = 0; t < W_RING; ++t)
#pragma unroll
    for (int piece = 0; piece < 3; ++piece) f.rb[t].p[piece] = u32x4{0u, 0u, 0u, 0u};
}

template <int OMASK, typename Stamp>
__device__ __forceinline__ void run_job(const NetJob& job, NetOperands& f0, NetOperands& f1, const float* __restrict__ W,
                                        const float* w_after, float* __restrict__ lds,
                                        const float* __restrict__ inp, int lane, int policy_channels, int n_valid,
                                        float* logits, float* value, Stamp&& stamp) {
  f32x4 acc[CELLS];
#pragma unroll
  for (int o = 0; o < CELLS; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  job_kloop<OMASK>(acc, f0, f1, lds + job.src * ACT_FLOATS, W + job.w_off, job.kgroups, w_after, lane);
  stamp(0);
  if (job.extra) extra_planes<OMASK>(acc, W + job.wx_off, inp, lane);
  stamp(1);
  epilogue<OMASK>(acc, job, lds, lane, policy_channels, n_valid, logits, value);
  stamp(2);
}

// Run the compiled network on the 16 positions whose input planes are in `inp`
// ([cell][pos][4 planes]); `lds` holds the activation buffers.  Every thread of
// the 256-thread workgroup must call it; it ends with a workgroup barrier.
// Outputs: logits [pos][policy_channels][9] and value [pos] for pos < n_valid
// (any address space).
// STAMPS (diagnostic build only): wave 0 adds up the shader-clock ticks it spends in the K
// loops, the input-plane step, the epilogues and at the stage barriers into stamps[0..3].
template <bool STAMPS = false>
__device__ __forceinline__ void net_tile(const NetProgram* __restrict__ prog, const float* __restrict__ W,
                                         float* __restrict__ lds, const float* __restrict__ inp,
                                         int policy_channels, int n_valid, float* logits, float* value,
                                         unsigned long long* stamps = nullptr) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const NetJob* __restrict__ jobs = prog->jobs[wave];
  const int n_jobs = prog->n_jobs[wave];

  NetOperands f0, f1;
  zero_b(f0);
  if (prog->first_w_off[wave] >= 0) load_b(f0, W + prog->first_w_off[wave], lane);

  unsigned long long tk[4] = {0, 0, 0, 0}, ts = 0;
  auto stamp = [&](int slot) {
    if constexpr (STAMPS) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tk[slot] += now - ts;
      ts = now;
    }
  };
  if constexpr (STAMPS) ts = __builtin_amdgcn_s_memtime();

  for (int j = 0; j < n_jobs; ++j) {
    const NetJob job = jobs[j];
    if (job.og != OG_NONE) {
      const float* w_after = (job.kgroups > 0 && job.next_w_off >= 0) ? W + job.next_w_off : nullptr;
      switch (job.og) {
        case 0: run_job<og_mask(0)>(job, f0, f1, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, value, stamp); break;
        case 1: run_job<og_mask(1)>(job, f0, f1, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, value, stamp); break;
        case 2: run_job<og_mask(2)>(job, f0, f1, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, value, stamp); break;
        case 3: run_job<og_mask(3)>(job, f0, f1, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, value, stamp); break;
        default: run_job<og_mask(4)>(job, f0, f1, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, value, stamp); break;
      }
    }
    stamp(2);
    if (job.stage_end) __syncthreads();
    stamp(3);
  }
  if constexpr (STAMPS) {
    if (tid == 0)
      for (int i = 0; i < 4; ++i) stamps[i] = tk[i];
  }
}

}  // namespace nz
