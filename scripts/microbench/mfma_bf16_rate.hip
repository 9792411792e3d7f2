// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 in the patterns of net_dev.hpp (gfx950, one wave
// per SIMD): chains of six dependent MFMAs per accumulator (the six split terms of one pair), with
// constant operands (zeros-like) and with operands that change every step (random-like bits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mf(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void mf_agpr(f32x4& c, const u32x4& a, const u32x4& b) {   // accumulator and A operand in AGPRs
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mf_acc_agpr(f32x4& c, const u32x4& a, const u32x4& b) {   // only the accumulator in AGPRs
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// MODE 0: 9 accumulators round-robin, same operands; 1: chains of 6 on one accumulator, 3+3 operand pieces;
// 2: as 1 with accumulators and the A operand in AGPRs; 3: as 1 with only the accumulators in AGPRs
template <int MODE>
__global__ __launch_bounds__(256) void kern(float* out, unsigned long long* ticks, int iters, uint32_t seed) {
  f32x4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = f32x4{0, 0, 0, 0};
  u32x4 w[3], x[9][3];
  uint32_t s = seed * 2654435761u + threadIdx.x * 40503u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return seed ? ((s & 0x807F807Fu) | 0x3F003F00u) : 0x3F803F80u; };
  for (int p = 0; p < 3; ++p) w[p] = u32x4{rnd(), rnd(), rnd(), rnd()};
  for (int i = 0; i < 9; ++i) for (int p = 0; p < 3; ++p) x[i][p] = u32x4{rnd(), rnd(), rnd(), rnd()};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = mf(w[r % 3], x[i][r / 2], acc[i]);
    } else if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (MODE == 2) {
          mf_agpr(acc[i], w[1], x[i][1]); mf_agpr(acc[i], w[0], x[i][2]); mf_agpr(acc[i], w[2], x[i][0]);
          mf_agpr(acc[i], w[0], x[i][1]); mf_agpr(acc[i], w[1], x[i][0]); mf_agpr(acc[i], w[0], x[i][0]);
        } else {
          mf_acc_agpr(acc[i], w[1], x[i][1]); mf_acc_agpr(acc[i], w[0], x[i][2]); mf_acc_agpr(acc[i], w[2], x[i][0]);
          mf_acc_agpr(acc[i], w[0], x[i][1]); mf_acc_agpr(acc[i], w[1], x[i][0]); mf_acc_agpr(acc[i], w[0], x[i][0]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        acc[i] = mf(w[1], x[i][1], acc[i]);
        acc[i] = mf(w[0], x[i][2], acc[i]);
        acc[i] = mf(w[2], x[i][0], acc[i]);
        acc[i] = mf(w[0], x[i][1], acc[i]);
        acc[i] = mf(w[1], x[i][0], acc[i]);
        acc[i] = mf(w[0], x[i][0], acc[i]);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0;
  for (int i = 0; i < 9; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks, uint32_t seed) {
  float* out; unsigned long long* ticks;
  const int threads = 256, iters = 2000;
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters, seed);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters, seed);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h; hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 54;
  printf("%-34s blocks %4d: %.2f ticks/MFMA/wave, %.1f us -> %.2f ns/MFMA/wave, %.0f TFLOP/s bf16, clock %.2f GHz if 1 tick = 1 cycle\n",
         name, blocks, h / n, ms * 1e3, ms * 1e6 / n, n * 4 * blocks * 16384.0 / (ms * 1e-3) / 1e12, h / (ms * 1e6));
  hipFree(out); hipFree(ticks);
}
int main() {
  for (int blocks : {1, 256}) {
    run<0>("round-robin 9 acc, constant ops", blocks, 0);
    run<0>("round-robin 9 acc, random ops", blocks, 7);
    run<1>("chains of 6, constant ops", blocks, 0);
    run<1>("chains of 6, random ops", blocks, 7);
    run<2>("chains of 6, acc + A in AGPRs", blocks, 7);
    run<3>("chains of 6, acc in AGPRs", blocks, 7);
  }
  return 0;
}
