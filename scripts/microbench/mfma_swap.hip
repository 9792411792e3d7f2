// Is v_mfma_f32_16x16x32_bf16 with its operands swapped the transposed product, bit for bit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const u32x4* a, const u32x4* b, f32x4* d1, f32x4* d2) {
  const int l = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  d1[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[l]), __builtin_bit_cast(bf16x8, b[l]), z, 0, 0, 0);
  d2[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[l]), __builtin_bit_cast(bf16x8, a[l]), z, 0, 0, 0);
}
int main() {
  uint32_t ha[64 * 4], hb[64 * 4];
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; const uint32_t e = 120 + (s >> 28), m = (s >> 8) & 0x7f; return (uint32_t)(((s >> 3) & 1) << 15 | e << 7 | m); };
  for (int i = 0; i < 256; ++i) { ha[i] = rnd() | (rnd() << 16); hb[i] = rnd() | (rnd() << 16); }
  u32x4 *a, *b; f32x4 *d1, *d2;
  hipMalloc(&a, 1024); hipMalloc(&b, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024);
  hipMemcpy(a, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(b, hb, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d1, d2);
  float h1[256], h2[256];
  hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
  // d1: lane l reg r = D[i = 4 (l / 16) + r][j = l % 16];  d2 should be D'[j][i]: lane l reg r = D[i = l % 16][j = 4 (l / 16) + r]
  int bad = 0; double maxd = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int i = l % 16, j = 4 * (l / 16) + r;             // element of D that d2[l][r] should hold
    const int l1 = 16 * (i / 4) + j, r1 = i % 4;            // where d1 holds D[i][j]
    const float x = h2[l * 4 + r], y = h1[l1 * 4 + r1];
    if (x != y) { ++bad; double d = x - y; if (d < 0) d = -d; if (d > maxd) maxd = d; }
  }
  printf("mismatches %d of 256, max |diff| %g (sample d1[0]=%g d2[0]=%g)\n", bad, maxd, h1[0], h2[0]);
  return 0;
}
