// Fused policy/value network for one tile of 16 positions (gfx950 matrix cores).
// Device function shared by the stand-alone network kernel (net.hip) and the
// persistent self-play kernel (selfplay.hip).
//
// Computes Network_Manager.inference (Neural_Networks/Network_Manager.py:46-64)
// for the square-conv RecurrentNet / ResNet / ConvNet (Neural_Networks/Architectures/
// RecurrentNet.py:82-99; BasicBlock blocks.py:37-41; Reduce_PolicyHead
// blocks.py:130-170; Reduce_ValueHead blocks.py:46-92) on a batch of 3x3 boards:
// every conv is 3x3, stride 1, zero 'same' padding, bias-free.
//
// One workgroup runs the WHOLE network for 16 positions; activations never
// leave LDS and the weights stream from L2 straight into registers.
//
// Mapping.  On a 3x3 board with a 3x3 kernel every output cell o sees input
// cell i through exactly one tap (tap = i - o + centre) when |dy|,|dx| <= 1, so
// a conv layer is, per output cell, a dense [C_out] x [C_in * n_valid(o)] x
// [16 positions] product.  The zero-padding taps are never multiplied: 49 of the 81
// (cell, tap) pairs exist.
//
// Arithmetic.  Every float32 product a * w is formed on the BF16 matrix cores from exact
// three-way splits a = a0 + a1 + a2, w = w0 + w1 + w2 (each piece the next 8 significant
// bits, a bf16 number) as the six terms a1w1 + a2w0 + a0w2 + a1w0 + a0w1 + a0w0 accumulated
// in float32; the dropped terms are below 2^-23 |a w|, the rounding of the float32 product
// itself.  v_mfma_f32_16x16x32_bf16 runs at 16x the FP32 MFMA rate, so the six of them per
// 32 channels take 96 cycles where eight v_mfma_f32_16x16x4_f32 take 256.  Measured against
// float64-accumulated convolutions the result is closer than a plain float32 convolution
// (DESIGN.md section 4).  The weights are split once on the host (engine.hip pack_conv), the
// activations once per element in the epilogue that produces them: LDS holds the three bf16
// pieces, the K loop reads them and issues MFMAs, nothing else.  The (<= 4) raw input planes of
// the projection / recall convs go through one v_mfma_f32_16x16x4_f32 per pair.
//
// Orientation.  The weights are the MFMA's A operand (rows = 16 output channels), the
// activations its B operand (columns = 16 positions): lane l supplies W[cout = l & 15]
// [k = 8 (l >> 4) + j] and X[k = 8 (l >> 4) + j][pos = l & 15], j = 0..7, and receives
// C[cout = 4 (l >> 4) + r][pos = l & 15] in register r -- four consecutive channels of one
// position, which the epilogue splits and writes as one 8-byte LDS store per piece.  Each
// position's column is independent of the others, so a position's outputs do not depend on
// which batch slot it occupies.
//
// Work split.  The host compiles the network into one job list per wave
// (engine.hip): a job is one (layer, 16-channel output tile, output-cell group).
// 64-channel layers give each of the 4 waves one full tile; the narrow head layers are cut
// by output-cell group as well so that all four matrix pipes stay busy.  Per 32-channel K
// group a wave reads 3 x 16 bytes per used input cell from LDS (refilled in place for the next
// K group as soon as the last tap that needs a cell has issued) and streams the weights tap by
// tap from L2, one whole K group ahead of the MFMAs -- also across jobs and stage barriers.
//
// LDS layout: act[buffer][piece][cell][pos][64 ch] bf16, the 16-byte slot index XOR-ed with
// (pos >> 1) & 7 so that ds_read_b128 (operands) and ds_write_b64 (epilogue) are
// bank-conflict free.  Two buffers (ping-pong; residual blocks update in place).
#pragma once
#include "engine.h"

namespace nz {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int POS = 16;            // positions per workgroup (MFMA N)
constexpr int CELLS = 9;
constexpr int ROW = 64;            // channels per (cell, pos) row
constexpr int NET_WAVES = 8;            // two wavefronts per SIMD: one's epilogue / operand loads run under the other's MFMAs
constexpr int NET_THREADS = NET_WAVES * 64;
constexpr int NET_BUFFERS = NET_ACT_BUFFERS;
constexpr int PIECE_BYTES = CELLS * POS * ROW * 2;          // one bf16 piece of one buffer
constexpr int ACT_BYTES = 3 * PIECE_BYTES;
constexpr int ACT_FLOATS = ACT_BYTES / 4;                   // LDS is declared as float[] by the kernels
constexpr int INP_FLOATS = CELLS * POS * 4;
constexpr int VAL_FLOATS = CELLS * POS;                      // the value head's last layer, per cell, before the mean
constexpr int NET_LDS_FLOATS = NET_BUFFERS * ACT_FLOATS + INP_FLOATS + VAL_FLOATS;
constexpr int TAP_DWORDS = 3 * 64 * 4;                      // one tap of one K group: [piece][lane][8 bf16]
constexpr int W_RING = 3;                                   // taps in flight: the weight stream runs three taps ahead of the MFMAs
static_assert(NET_WAVES == NET_WAVES_HOST, "job lists are per wave");
static_assert(NET_KG_DWORDS == 9 * TAP_DWORDS && NET_KG_CHANNELS == 32, "host packing (engine.hip) and kernel agree");

// output-cell groups: all nine | four quarters {4,0} {1,3} {5,7} {2,6,8} | two halves {0,1,2,3,5} {4,6,7,8}
// ((input cell, tap) pairs: 49 | 13, 12, 12, 12 | 26, 23).  The halves of a tile go to the two waves of a SIMD: the older
// wave wins the matrix pipe, so its epilogue runs under the other's MFMAs and one epilogue is left when those end.
// (An uneven 7 + 2 split leaves a shorter epilogue but measured the same and spilled four registers in the large job.)
__host__ __device__ constexpr int og_mask(int og) {
  return og == 0 ? 0x1FF : og == 1 ? 0x011 : og == 2 ? 0x00A : og == 3 ? 0x0A0 : og == 4 ? 0x144 :
         og == 5 ? 0x02F : og == 6 ? 0x1D0 : 0;
}

// byte offset of the 16-byte slot holding channels 8 s .. 8 s + 7 of (cell, pos) inside one piece
__device__ __forceinline__ int slot_addr(int cell, int pos, int s) {
  return ((cell * POS + pos) << 7) + (((s ^ (pos >> 1)) & 7) << 4);
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

// first / last tap (0..8) that reads input cell I for this output-cell group
template <int OMASK, int I>
constexpr int first_use() {
  return pair_used<OMASK, I, 0>() ? 0 : pair_used<OMASK, I, 1>() ? 1 : pair_used<OMASK, I, 2>() ? 2 :
         pair_used<OMASK, I, 3>() ? 3 : pair_used<OMASK, I, 4>() ? 4 : pair_used<OMASK, I, 5>() ? 5 :
         pair_used<OMASK, I, 6>() ? 6 : pair_used<OMASK, I, 7>() ? 7 : 8;
}
template <int OMASK, int I>
constexpr int last_use() {
  return pair_used<OMASK, I, 8>() ? 8 : pair_used<OMASK, I, 7>() ? 7 : pair_used<OMASK, I, 6>() ? 6 :
         pair_used<OMASK, I, 5>() ? 5 : pair_used<OMASK, I, 4>() ? 4 : pair_used<OMASK, I, 3>() ? 3 :
         pair_used<OMASK, I, 2>() ? 2 : pair_used<OMASK, I, 1>() ? 1 : 0;
}
// When cell I's operands of the SECOND K group are read, on a timeline of 18 steps (step s < 9: after tap s of the first
// K group; step 9 + t: before tap t of the second): not before the first group is done with the registers, and not
// earlier than REFILL_AHEAD taps before the second group needs them -- operands that sit in registers for a whole K
// group cost the large jobs their last free registers (a four-register spill around the K loop otherwise).
constexpr int REFILL_AHEAD = 3;
template <int OMASK, int I>
constexpr int refill_step() {
  constexpr int earliest = last_use<OMASK, I>();                       // after that tap of group one
  constexpr int wanted = 9 + first_use<OMASK, I>() - REFILL_AHEAD;      // before that tap of group two
  return wanted > earliest ? wanted : earliest;
}

struct Pieces {           // the three bf16 pieces of 8 channels (one lane's share of a 32-channel K group)
  u32x4 p[3];
};
struct FragS {            // the wavefront's weight stream, W_RING taps ahead of the MFMAs
  Pieces rb[W_RING];
};

// ---- exact three-way split of float32 into bf16 pieces -------------------------------------------
__device__ __forceinline__ uint32_t trunc_bf16(float x) { return __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u; }
// (x0, x1) -> the two upper halves in one register: bf16(x0) | bf16(x1) << 16 (truncating)
__device__ __forceinline__ uint32_t pack_hi16(float x0, float x1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, x0), 0x07060302u);
}
// x = p0 + p1 + p2 exactly, 8 significant bits each
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = pack_hi16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, trunc_bf16(x0));
  const float r1 = x1 - __builtin_bit_cast(float, trunc_bf16(x1));
  p1 = pack_hi16(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, trunc_bf16(r0));
  const float s1 = r1 - __builtin_bit_cast(float, trunc_bf16(r1));
  p2 = pack_hi16(s0, s1);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& w, const u32x4& x, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
template <int OMASK, int I, int TAP>
__device__ __forceinline__ void pair_mfma(f32x4 (&acc)[CELLS], const Pieces (&x)[CELLS], const Pieces& w) {
  if constexpr (pair_used<OMASK, I, TAP>()) {
    constexpr int o = TapMap<I, TAP>::o;
    // small terms first; one accumulator chain issues at the full rate
    acc[o] = mfma_bf16(w.p[1], x[I].p[1], acc[o]);
    acc[o] = mfma_bf16(w.p[0], x[I].p[2], acc[o]);
    acc[o] = mfma_bf16(w.p[2], x[I].p[0], acc[o]);
    acc[o] = mfma_bf16(w.p[0], x[I].p[1], acc[o]);
    acc[o] = mfma_bf16(w.p[1], x[I].p[0], acc[o]);
    acc[o] = mfma_bf16(w.p[0], x[I].p[0], acc[o]);
    // keep the chain together: the scheduler would interleave it with the other pairs' chains, and
    // round-robin over accumulators issues at 21 cycles per MFMA instead of 16
    // (scripts/microbench/mfma_bf16_rate.hip)
    __builtin_amdgcn_sched_barrier(0);
  }
}
// all (input cell, TAP) pairs of the job's output cells: independent accumulators
template <int OMASK, int TAP>
__device__ __forceinline__ void tap_mfma(f32x4 (&acc)[CELLS], const Pieces (&x)[CELLS], const Pieces& w) {
  pair_mfma<OMASK, 0, TAP>(acc, x, w); pair_mfma<OMASK, 1, TAP>(acc, x, w); pair_mfma<OMASK, 2, TAP>(acc, x, w);
  pair_mfma<OMASK, 3, TAP>(acc, x, w); pair_mfma<OMASK, 4, TAP>(acc, x, w); pair_mfma<OMASK, 5, TAP>(acc, x, w);
  pair_mfma<OMASK, 6, TAP>(acc, x, w); pair_mfma<OMASK, 7, TAP>(acc, x, w); pair_mfma<OMASK, 8, TAP>(acc, x, w);
}

template <int OMASK, int I>
__device__ __forceinline__ void load_cell(Pieces (&x)[CELLS], const unsigned char* __restrict__ src, int a0) {
#ifdef NZ_ABLATE_A   // timing-only build: no LDS reads of the activation operands (outputs are wrong)
  if constexpr (input_used<OMASK, I>()) { const uint32_t v = 0x3C003C00u + a0 + I; x[I].p[0] = x[I].p[1] = x[I].p[2] = u32x4{v, v, v, v}; asm volatile("" :: "v"(src)); }
#else
  if constexpr (input_used<OMASK, I>()) {
#pragma unroll
    for (int piece = 0; piece < 3; ++piece)
      x[I].p[piece] = *reinterpret_cast<const u32x4*>(src + piece * PIECE_BYTES + I * (POS * 128) + a0);
  }
#endif
}
// this lane's activation operands of K group kg: channels 32 kg + 8 quad .. + 7 of every used input cell
template <int OMASK>
__device__ __forceinline__ void load_cells(Pieces (&x)[CELLS], const unsigned char* __restrict__ src, int pos, int quad,
                                           int kg) {
  const int a0 = slot_addr(0, pos, kg * 4 + quad);
  load_cell<OMASK, 0>(x, src, a0); load_cell<OMASK, 1>(x, src, a0); load_cell<OMASK, 2>(x, src, a0);
  load_cell<OMASK, 3>(x, src, a0); load_cell<OMASK, 4>(x, src, a0); load_cell<OMASK, 5>(x, src, a0);
  load_cell<OMASK, 6>(x, src, a0); load_cell<OMASK, 7>(x, src, a0); load_cell<OMASK, 8>(x, src, a0);
}
__device__ __forceinline__ void load_tap(Pieces& t, const float* __restrict__ w, int lane) {
#ifdef NZ_ABLATE_B   // timing-only build: no global loads of the weight operands (outputs are wrong)
  const uint32_t v = 0x3C003C00u + lane;
  t.p[0] = t.p[1] = t.p[2] = u32x4{v, v, v, v};
  asm volatile("" :: "v"(w));
#else
  // uniform base + unsigned 32-bit lane offset: the loads take their base from scalar registers (no 64-bit address per
  // tap).  The offset is opaque to the optimiser, which would otherwise fold it into a per-lane base outside the job loop.
  unsigned off = (unsigned)lane * 16u;
  asm volatile("" : "+v"(off));
  typedef const __attribute__((address_space(1))) u32x4* gptr;   // global (not flat) loads: vmcnt only
  unsigned long long b = reinterpret_cast<unsigned long long>(w);
  asm volatile("" : "+s"(b));                              // the tap's base stays a scalar: base + lane offset + immediate
  t.p[0] = *reinterpret_cast<gptr>(b + off);
  t.p[1] = *reinterpret_cast<gptr>(b + off + 1024);
  t.p[2] = *reinterpret_cast<gptr>(b + off + 2048);
#endif
}
__device__ __forceinline__ void load_b(FragS& f, const float* __restrict__ w, int lane) {   // a whole K group
#pragma unroll
  for (int i = 0; i < W_RING; ++i) load_tap(f.rb[i], w + i * TAP_DWORDS, lane);
}
__device__ __forceinline__ void zero_b(FragS& f) {
#pragma unroll
  for (int t = 0; t < W_RING; ++t)
#pragma unroll
    for (int piece = 0; piece < 3; ++piece) f.rb[t].p[piece] = u32x4{0u, 0u, 0u, 0u};
}

// the second K group's operands of every cell whose refill_step is STEP (slot address a1)
template <int OMASK, int STEP>
__device__ __forceinline__ void refill(Pieces (&x)[CELLS], const unsigned char* __restrict__ src, int a1) {
  if constexpr (refill_step<OMASK, 0>() == STEP) load_cell<OMASK, 0>(x, src, a1);
  if constexpr (refill_step<OMASK, 1>() == STEP) load_cell<OMASK, 1>(x, src, a1);
  if constexpr (refill_step<OMASK, 2>() == STEP) load_cell<OMASK, 2>(x, src, a1);
  if constexpr (refill_step<OMASK, 3>() == STEP) load_cell<OMASK, 3>(x, src, a1);
  if constexpr (refill_step<OMASK, 4>() == STEP) load_cell<OMASK, 4>(x, src, a1);
  if constexpr (refill_step<OMASK, 5>() == STEP) load_cell<OMASK, 5>(x, src, a1);
  if constexpr (refill_step<OMASK, 6>() == STEP) load_cell<OMASK, 6>(x, src, a1);
  if constexpr (refill_step<OMASK, 7>() == STEP) load_cell<OMASK, 7>(x, src, a1);
  if constexpr (refill_step<OMASK, 8>() == STEP) load_cell<OMASK, 8>(x, src, a1);
}
// One K group (32 channels), tap-major.  The weight ring holds three taps: tap t's operands sit in slot t % 3 and, once
// its MFMAs are issued, the slot is refilled with tap t + 3 -- of this K group (`wc`) or, for the last three taps, with
// taps 0-2 of the NEXT K group (`wn`; when nothing follows, any readable weights: the loads are unconditional) -- so the
// weight stream runs three taps ahead, also across jobs and stage barriers.  PART 1 / 2: the first / second K group of a
// two-group job -- the activation operands of the second group are read in place (LDS slot address a1) at their
// refill_step, some after a tap of the first group, the others before a tap of the second; PART 0: a one-group job.
template <int OMASK, int PART>
__device__ __forceinline__ void kgroup(f32x4 (&acc)[CELLS], Pieces (&x)[CELLS], FragS& f,
                                       const unsigned char* __restrict__ src, int a1,
                                       const float* __restrict__ wc, const float* __restrict__ wn, int lane) {
  constexpr bool A = PART == 1, B = PART == 2;
  if constexpr (B) refill<OMASK, 9>(x, src, a1);
  tap_mfma<OMASK, 0>(acc, x, f.rb[0]); load_tap(f.rb[0], wc + 3 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 0>(x, src, a1);
  if constexpr (B) refill<OMASK, 10>(x, src, a1);
  tap_mfma<OMASK, 1>(acc, x, f.rb[1]); load_tap(f.rb[1], wc + 4 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 1>(x, src, a1);
  if constexpr (B) refill<OMASK, 11>(x, src, a1);
  tap_mfma<OMASK, 2>(acc, x, f.rb[2]); load_tap(f.rb[2], wc + 5 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 2>(x, src, a1);
  if constexpr (B) refill<OMASK, 12>(x, src, a1);
  tap_mfma<OMASK, 3>(acc, x, f.rb[0]); load_tap(f.rb[0], wc + 6 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 3>(x, src, a1);
  if constexpr (B) refill<OMASK, 13>(x, src, a1);
  tap_mfma<OMASK, 4>(acc, x, f.rb[1]); load_tap(f.rb[1], wc + 7 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 4>(x, src, a1);
  if constexpr (B) refill<OMASK, 14>(x, src, a1);
  tap_mfma<OMASK, 5>(acc, x, f.rb[2]); load_tap(f.rb[2], wc + 8 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 5>(x, src, a1);
  if constexpr (B) refill<OMASK, 15>(x, src, a1);
  tap_mfma<OMASK, 6>(acc, x, f.rb[0]); load_tap(f.rb[0], wn + 0 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 6>(x, src, a1);
  if constexpr (B) refill<OMASK, 16>(x, src, a1);
  tap_mfma<OMASK, 7>(acc, x, f.rb[1]); load_tap(f.rb[1], wn + 1 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 7>(x, src, a1);
  if constexpr (B) refill<OMASK, 17>(x, src, a1);
  tap_mfma<OMASK, 8>(acc, x, f.rb[2]); load_tap(f.rb[2], wn + 2 * TAP_DWORDS, lane);
  if constexpr (A) refill<OMASK, 8>(x, src, a1);
}
// All K groups of one job as straight-line code (KG = 1 or 2: layers are at most 64 channels wide;
// a loop would carry the operand registers around its back edge through copies).  On entry the
// ring holds the first three taps of the job's first K group; on exit those of the next job that reads
// weights (`w_after`; the stream's start when there is none).
template <int OMASK, int KG>
__device__ __forceinline__ void job_kloop(f32x4 (&acc)[CELLS], FragS& f, const unsigned char* __restrict__ src,
                                          const float* __restrict__ w, const float* __restrict__ w_after, int lane) {
  const int pos = lane & 15, quad = lane >> 4;
  Pieces x[CELLS];
  load_cells<OMASK>(x, src, pos, quad, 0);
  if constexpr (KG == 2) {
    const int a1 = slot_addr(0, pos, 4 + quad);
    kgroup<OMASK, 1>(acc, x, f, src, a1, w, w + NET_KG_DWORDS, lane);
    kgroup<OMASK, 2>(acc, x, f, src, a1, w + NET_KG_DWORDS, w_after, lane);
  } else {
    kgroup<OMASK, 0>(acc, x, f, src, 0, w, w_after, lane);
  }
}

// tanh for the value head: 1 - 2 / (exp(2x) + 1) on the hardware exp; absolute error < 3e-7
// (the parity tolerance on values is 1e-5), exact limits at +-inf
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

template <int OMASK, int TAP, int I>
__device__ __forceinline__ void mfma_extra_pair(f32x4 (&acc)[CELLS], const float (&ax)[CELLS], const float (&bx)[9]) {
  if constexpr (pair_used<OMASK, I, TAP>()) {
    constexpr int o = TapMap<I, TAP>::o;
    acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(bx[TAP], ax[I], acc[o], 0, 0, 0);
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
// the (<= 4) raw input planes as one extra K step in plain float32 (projection / recall conv):
// lane l supplies W[cout = l & 15][plane = l >> 4] and X[plane = l >> 4][pos = l & 15]
template <int OMASK>
__device__ __forceinline__ void extra_planes(f32x4 (&acc)[CELLS], const float* __restrict__ wx,
                                             const float* __restrict__ inp, int lane) {
  float ax[CELLS], bx[9];
  // (explicit address spaces: a FLAT load in flight makes every later wait in the job a full vmcnt(0) / lgkmcnt(0))
  const __attribute__((address_space(1))) float* wxg = (const __attribute__((address_space(1))) float*)wx;
  const __attribute__((address_space(3))) float* inl = (const __attribute__((address_space(3))) float*)inp;
#pragma unroll
  for (int t = 0; t < 9; ++t) bx[t] = wxg[t * 64 + lane];
#pragma unroll
  for (int i = 0; i < CELLS; ++i) ax[i] = inl[(i * POS + (lane & 15)) * 4 + (lane >> 4)];
  mfma_extra_tap<OMASK, 0>(acc, ax, bx); mfma_extra_tap<OMASK, 1>(acc, ax, bx); mfma_extra_tap<OMASK, 2>(acc, ax, bx);
  mfma_extra_tap<OMASK, 3>(acc, ax, bx); mfma_extra_tap<OMASK, 4>(acc, ax, bx); mfma_extra_tap<OMASK, 5>(acc, ax, bx);
  mfma_extra_tap<OMASK, 6>(acc, ax, bx); mfma_extra_tap<OMASK, 7>(acc, ax, bx); mfma_extra_tap<OMASK, 8>(acc, ax, bx);
}

// ---- epilogues: lane holds C[cout = 16 nt + 4 quad + r][pos = lane & 15] for the job's cells ------
// ACT: 0 none, 1 relu, 2 tanh, 3 elu.  The four channels of a lane are half a 16-byte slot: one
// ds_write_b64 per piece (and one ds_read_b64 per piece for the residual, which the three pieces
// reproduce exactly).  Output cell o lives o * 2048 bytes after cell 0.
template <int OMASK, int ACT, bool RES>
__device__ __forceinline__ void epilogue_lds(const f32x4 (&acc)[CELLS], unsigned char* __restrict__ dst,
                                             const unsigned char* res, int lane, int nt) {
  const int pos = lane & 15, quad = lane >> 4;
  const int off = slot_addr(0, pos, nt * 2 + (quad >> 1)) + (quad & 1) * 8;
#ifdef NZ_ABLATE_EPI   // timing-only build: one store per job instead of the epilogue (outputs are wrong)
  {
    float t = 0.f;
#pragma unroll
    for (int o = 0; o < CELLS; ++o) if ((OMASK >> o) & 1) t += acc[o][0] + acc[o][1] + acc[o][2] + acc[o][3];
    *reinterpret_cast<float*>(dst + off) = t;
    return;
  }
#endif
  u32x2 q[CELLS][3];
  if constexpr (RES) {                                     // all residual reads in flight at once, before any write
#pragma unroll
    for (int o = 0; o < CELLS; ++o)
#pragma unroll
      for (int piece = 0; piece < 3; ++piece)
        if ((OMASK >> o) & 1) q[o][piece] = *reinterpret_cast<const u32x2*>(res + piece * PIECE_BYTES + o * (POS * 128) + off);
  }
#pragma unroll
  for (int o = 0; o < CELLS; ++o) {
    if (!((OMASK >> o) & 1)) continue;
    f32x4 v = acc[o];
    if constexpr (RES) {
      v[0] += (bf16_lo(q[o][0][0]) + bf16_lo(q[o][1][0])) + bf16_lo(q[o][2][0]);
      v[1] += (bf16_hi(q[o][0][0]) + bf16_hi(q[o][1][0])) + bf16_hi(q[o][2][0]);
      v[2] += (bf16_lo(q[o][0][1]) + bf16_lo(q[o][1][1])) + bf16_lo(q[o][2][1]);
      v[3] += (bf16_hi(q[o][0][1]) + bf16_hi(q[o][1][1])) + bf16_hi(q[o][2][1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = v[r];
      if constexpr (ACT == 1) x = fmaxf(x, 0.0f);
      if constexpr (ACT == 2) x = tanh_fast(x);
      if constexpr (ACT == 3) x = x > 0.0f ? x : expm1f(x);      // nn.ELU(), alpha = 1
      v[r] = x;
    }
    uint32_t a0, a1, a2, b0, b1, b2;
    split_pair(v[0], v[1], a0, a1, a2);
    split_pair(v[2], v[3], b0, b1, b2);
    *reinterpret_cast<u32x2*>(dst + 0 * PIECE_BYTES + o * (POS * 128) + off) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(dst + 1 * PIECE_BYTES + o * (POS * 128) + off) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(dst + 2 * PIECE_BYTES + o * (POS * 128) + off) = u32x2{a2, b2};
  }
}
template <int OMASK>
__device__ __forceinline__ void epilogue(const f32x4 (&acc)[CELLS], const NetJob& job, unsigned char* __restrict__ lds,
                                         int lane, int policy_channels, int n_valid, float* logits, float* vcells) {
  if (job.dst < NET_BUFFERS) {
    unsigned char* dst = lds + job.dst * ACT_BYTES;
    if (job.res >= 0) {
      epilogue_lds<OMASK, 1, true>(acc, dst, lds + job.res * ACT_BYTES, lane, job.nt);
    } else if (job.act == 1) {
      epilogue_lds<OMASK, 1, false>(acc, dst, nullptr, lane, job.nt);
    } else if (job.act == 2) {
      epilogue_lds<OMASK, 2, false>(acc, dst, nullptr, lane, job.nt);
    } else if (job.act == 3) {
      epilogue_lds<OMASK, 3, false>(acc, dst, nullptr, lane, job.nt);
    } else {
      epilogue_lds<OMASK, 0, false>(acc, dst, nullptr, lane, job.nt);
    }
  } else if (job.dst == NET_DST_POLICY) {     // policy logits [pos][P][9]
    // (addresses re-derived from an opaque copy of the lane id: hoisted out of the job loop they would sit in
    // registers -- in the persistent kernel: in scratch memory -- for the whole network)
    int l2 = lane;
    asm volatile("" : "+v"(l2));
    const int pos = l2 & 15, quad = l2 >> 4;
    if (pos < n_valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cout = job.nt * 16 + quad * 4 + r;
        if (cout < policy_channels) {
#pragma unroll
          for (int o = 0; o < CELLS; ++o)
            if ((OMASK >> o) & 1) logits[((size_t)pos * policy_channels + cout) * CELLS + o] = acc[o][r];
        }
      }
    }
  } else {                                     // value: channel 0 of every cell; net_tile takes the mean when all are in
    int l2 = lane;                             // as for the policy: no address hoisted out of the job loop
    asm volatile("" : "+v"(l2));
    const int pos = l2 & 15, quad = l2 >> 4;
    if (job.nt == 0 && quad == 0) {
#pragma unroll
      for (int o = 0; o < CELLS; ++o)
        if ((OMASK >> o) & 1) vcells[o * POS + pos] = acc[o][0];
    }
  }
}

// one job: K loop, input-plane step, epilogue
template <int OMASK, typename Stamp, typename Fetch>
__device__ __forceinline__ void run_job(const NetJob& job, FragS& f, const float* __restrict__ W,
                                        const float* w_after, unsigned char* __restrict__ lds,
                                        const float* __restrict__ inp, int lane, int policy_channels, int n_valid,
                                        float* logits, float* value, Stamp&& stamp, Fetch&& fetch_next) {
  // per-lane LDS addresses are derived inside the job from an opaque copy of the lane id: hoisted out of the job loop
  // they are spilled and reloaded in the middle of the K loop, behind a wait for the whole weight stream
  asm volatile("" : "+v"(lane));
  f32x4 acc[CELLS];
#pragma unroll
  for (int o = 0; o < CELLS; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (job.kgroups == 2) job_kloop<OMASK, 2>(acc, f, lds + job.src * ACT_BYTES, W + job.w_off, w_after, lane);
  else if (job.kgroups == 1) job_kloop<OMASK, 1>(acc, f, lds + job.src * ACT_BYTES, W + job.w_off, w_after, lane);
  stamp(0);
  if (job.extra) extra_planes<OMASK>(acc, W + job.wx_off, inp, lane);
  stamp(1);
  fetch_next();          // the next job's descriptor: in flight under the epilogue, in no register during the K loop
  epilogue<OMASK>(acc, job, lds, lane, policy_channels, n_valid, logits, value);   // `value`: the per-cell staging area
  stamp(2);
}

// Run the compiled network on the 16 positions whose input planes are in `inp`
// ([cell][pos][4 planes]); `lds_f` holds the activation buffers (NET_BUFFERS * ACT_FLOATS floats,
// no NaN bit patterns: the callers zero it once), then `inp`'s INP_FLOATS, then VAL_FLOATS of staging.  Every thread of the 256-thread workgroup must
// call it; it ends with a workgroup barrier.
// Outputs: logits [pos][policy_channels][9] and value [pos] for pos < n_valid
// (any address space).
// STAMPS (diagnostic build only): wave 0 adds up the shader-clock ticks it spends in the K
// loops, the input-plane step, the epilogues and at the stage barriers into stamps[0..3].
template <bool STAMPS = false>
__device__ __forceinline__ void net_tile(const NetProgram* __restrict__ prog, const float* __restrict__ W,
                                         float* __restrict__ lds_f, const float* __restrict__ inp,
                                         int policy_channels, int n_valid, float* logits, float* value,
                                         unsigned long long* stamps = nullptr) {
  unsigned char* __restrict__ lds = reinterpret_cast<unsigned char*>(lds_f);
  float* const vcells = lds_f + NET_BUFFERS * ACT_FLOATS + INP_FLOATS;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                     // per-lane addresses are derived here, not hoisted out of the caller's loop
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const NetJob* __restrict__ jobs = prog->jobs[wave];
  // (uniform by construction; said explicitly for callers that pass `prog` as a generic pointer)
  const int n_jobs = __builtin_amdgcn_readfirstlane(prog->n_jobs[wave]);
  const int first_w = __builtin_amdgcn_readfirstlane(prog->first_w_off[wave]);

  FragS f;
  zero_b(f);
  if (first_w >= 0) load_b(f, W + first_w, lane);

  unsigned long long tk[4] = {0, 0, 0, 0}, ts = 0;
#ifdef NZ_STAGE_STAMPS
  unsigned long long pk[3] = {0, 0, 0}, st_rec[24][3];
  int n_st = 0;
#endif
  auto stamp = [&](int slot) {
    if constexpr (STAMPS) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tk[slot] += now - ts;
      ts = now;
    }
  };
  if constexpr (STAMPS) ts = __builtin_amdgcn_s_memtime();

  // The next job's descriptor is fetched while this job's epilogue runs -- with VECTOR loads (every lane reads the same six
  // words, v_readfirstlane makes them scalars again).  As a scalar load it would share lgkmcnt with the K loop's LDS
  // operand reads and return out of order with them; that variant computes wrong outputs.
  static_assert(sizeof(NetJob) == 24, "six dwords");
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));                  // a zero the compiler must keep in a vector register
  const __attribute__((address_space(1))) uint32_t* jw =
      (const __attribute__((address_space(1))) uint32_t*)reinterpret_cast<const uint32_t*>(jobs) + vzero;   // global, not FLAT
  uint32_t nw[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) nw[i] = jw[i];
  for (int j = 0; j < n_jobs; ++j) {
    NetJob job;
    {
      uint32_t cw[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) cw[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)nw[i]);
      __builtin_memcpy(&job, cw, sizeof(job));
    }
    // Every vector load still in flight here was issued a whole epilogue ago (this job's first three taps): an explicit
    // wait costs nothing and gives the K loop exact counts.  Left to itself the compiler, which cannot count loads across
    // the loop's back edges, makes the first wait inside the job a vmcnt(0) -- behind the taps issued just before it.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
    auto fetch_next = [&]() {
      if (j + 1 < n_jobs) {
#pragma unroll
        for (int i = 0; i < 6; ++i) nw[i] = jw[(j + 1) * 6 + i];
      }
    };
    if (job.og == OG_NONE) fetch_next();
    if (job.og != OG_NONE) {
      const float* w_after = W + (job.next_w_off >= 0 ? job.next_w_off : 0);   // never null: straight-line K loops
      switch (job.og) {
        // (group 0, all nine cells in one job, is never scheduled: layers are at most four tiles wide and eight waves
        // want a unit each -- engine.hip add_stage)
        case 1: run_job<og_mask(1)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
        case 2: run_job<og_mask(2)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
        case 3: run_job<og_mask(3)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
        case 4: run_job<og_mask(4)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
        case 5: run_job<og_mask(5)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
        default: run_job<og_mask(6)>(job, f, W, w_after, lds, inp, lane, policy_channels, n_valid, logits, vcells, stamp, fetch_next); break;
      }
    }
    stamp(2);
    if (job.stage_end) __syncthreads();
    stamp(3);
#ifdef NZ_STAGE_STAMPS   // diagnostic build: per-stage ticks of waves 0 and 4 of workgroup 0 (K loops + planes, epilogue, barrier)
    if constexpr (STAMPS) {
      if (job.stage_end && blockIdx.x == 0 && (tid == 0 || tid == 256) && n_st < 24) {
        st_rec[n_st][0] = tk[0] + tk[1] - pk[0]; st_rec[n_st][1] = tk[2] - pk[1]; st_rec[n_st][2] = tk[3] - pk[2];
        ++n_st;
        pk[0] = tk[0] + tk[1]; pk[1] = tk[2]; pk[2] = tk[3];
      }
    }
#endif
  }
  // value = tanh(mean over (C=1,H,W)) (blocks.py:82-84); the last stage's jobs left channel 0 of each cell in `vcells`
  // (the program's last stage ends with a barrier).  Cells are added in index order, whichever wave produced them.
  if (tid < POS && tid < n_valid) {
    float sum = 0.0f;
#pragma unroll
    for (int o = 0; o < CELLS; ++o) sum += vcells[o * POS + tid];
    value[tid] = tanh_fast(sum / 9.0f);
  }
  __syncthreads();
  if constexpr (STAMPS) {
    if (tid == 0)
      for (int i = 0; i < 4; ++i) stamps[i] = tk[i];
#ifdef NZ_STAGE_STAMPS
    if (blockIdx.x == 0 && (tid == 0 || tid == 256))
      for (int i = 0; i < n_st; ++i)
        printf("wave %d stage %d: k %llu epi %llu bar %llu\n", wave, i, st_rec[i][0], st_rec[i][1], st_rec[i][2]);
#endif
  }
}

}  // namespace nz
