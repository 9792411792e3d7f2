// Device pieces of the one-launch BF16 network (boardnet.hip fused16_net_kernel) that the persistent SCS self-play
// kernel (scs_search.hip) runs per wavefront too: the split-bf16 arithmetic -- every float32 the exact sum of three bf16
// pieces, six of the nine piece products on v_mfma_f32_16x16x32_bf16 with float32 accumulation --
// and the layer program the host resolves (every LDS offset and stride).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nz {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t wide_pack_hi16(float x0, float x1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, x1), __builtin_bit_cast(uint32_t, x0), 0x07060302u);
}
__device__ __forceinline__ float wide_trunc(float x) {
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u);
}
// 8 float32 -> three bf16 pieces (exact: 8 + 8 + 8 significant bits)
__device__ __forceinline__ void wide_split8(const f32x4& lo, const f32x4& hi, u32x4& p0, u32x4& p1, u32x4& p2) {
  const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  uint32_t a[4], b[4], c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v[2 * j], x1 = v[2 * j + 1];
    a[j] = wide_pack_hi16(x0, x1);
    const float r0 = x0 - wide_trunc(x0), r1 = x1 - wide_trunc(x1);
    b[j] = wide_pack_hi16(r0, r1);
    c[j] = wide_pack_hi16(r0 - wide_trunc(r0), r1 - wide_trunc(r1));
  }
  p0 = u32x4{a[0], a[1], a[2], a[3]};
  p1 = u32x4{b[0], b[1], b[2], b[3]};
  p2 = u32x4{c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ f32x4 wide_mfma(const u32x4& w, const u32x4& x, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

constexpr int FUSED_MAX_OPS = 176;
struct Fused16Op {
  const uint32_t* w;                       // [col tile][tap][32-channel group][piece][lane][4 dwords = 8 bf16]
  int32_t off0, cs0, ps0, kg0;             // source 0: float offset, floats per row and per piece, piece stride, K groups
  int32_t off1, cs1, ps1, kg1;             // source 1 (off1 = -1: none)
  int32_t offd, csd, psd;                  // destination (psd = 0: float32 rows [row][csd])
  int32_t offr, csr, psr;                  // residual (offr = -1: none)
  int32_t ntiles, act;
  int32_t w_lds, w_chunks;                 // 1: weights staged in LDS; 16-byte chunks of one column tile
  int32_t w_slot, w_after_barrier;
  int32_t pad;
};
struct Fused16Program {
  int32_t n_ops, hw, h, wd, planes, hex;
  int32_t zrow_index, lds_floats;           // every pieces buffer has a row of zeros at this index (never written)
  int32_t clear_from, pad2;                 // LDS floats [clear_from, lds_floats) start as zeros: the activation buffers
  int32_t wbuf_off[2];
  int32_t in_off, in_cs, in_ps, pol_off, pol_cs, val_off, val_cs, pad;
  int32_t zero_at_op, n_zero, zero_off[6], zero_len;   // rows of zeros to (re)make before that layer: buffers that take over a weight buffer's space
  // the per-wavefront-pair program (boardnet_wave_program) only: the layers from solo_at on are two independent chains
  // (the policy head's solo_pol layers, then the value head's) that the pair's two wavefronts run side by side; the first
  // chain's hidden buffer has its row of zeros at float offset solo_zero_off (solo_zero_len floats), remade every pass
  int32_t solo_at, solo_pol, solo_zero_off, solo_zero_len;
  Fused16Op ops[FUSED_MAX_OPS];
};
// one K step (32 channels of one tap): the six piece products, small terms first (net_dev.hpp pair_mfma)
// The weights go in as the MFMA's first operand, so the output tile comes out transposed: lane l holds ROW l & 15 and the
// four consecutive channels 4 (l >> 4) .. + 3 -- one address, one bounds check and three 8-byte stores per lane in the
// epilogue where the other orientation (a column and four rows per lane) needs four of each and twelve 2-byte stores.
__device__ __forceinline__ void step16(f32x4& acc, const u32x4 (&a)[3], const u32x4 (&b)[3]) {
  acc = wide_mfma(b[1], a[1], acc);
  acc = wide_mfma(b[0], a[2], acc);
  acc = wide_mfma(b[2], a[0], acc);
  acc = wide_mfma(b[0], a[1], acc);
  acc = wide_mfma(b[1], a[0], acc);
  acc = wide_mfma(b[0], a[0], acc);
}
// exp(x) - 1 for x <= 0 to an absolute error of ~1e-7 (v_exp_f32; libm's expm1f keeps the RELATIVE error small near zero,
// thirty instructions the activations' 1e-5 tolerance has no use for)
__device__ __forceinline__ float fast_expm1(float x) { return __expf(x) - 1.0f; }
// exact three-way split of one value; piece i as the bf16 bit pattern
__device__ __forceinline__ void split3_bits(float v, uint16_t (&h)[3]) {
  float r = v;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint32_t bits = __builtin_bit_cast(uint32_t, r) & 0xFFFF0000u;
    h[i] = (uint16_t)(bits >> 16);
    r = r - __builtin_bit_cast(float, bits);
  }
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }


}  // namespace nz
