// Micro-benchmark: issue rate of the f32-input MFMA shapes on gfx950, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, unsigned long long* ticks, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* ticks, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <typename F>
void run(const char* name, F f, int nacc, int blocks, int threads, double flop_per_mfma) {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipMalloc(&ticks, blocks * sizeof(unsigned long long));
  const int iters = 200;
  f(out, ticks, iters, blocks, threads);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  f(out, ticks, iters, blocks, threads);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h; hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double n = (double)iters * 8 * nacc;
  printf("%-22s blocks %4d waves/blk %d: %.1f ticks/MFMA/wave, %.1f us, %.1f TFLOP/s\n", name, blocks, threads / 64,
         h / n, ms * 1e3, n * (threads / 64) * blocks * flop_per_mfma / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(ticks);
}
int main() {
  for (int blocks : {1, 256}) {
    run("16x16x4 f32, 9 acc", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k16<9>, dim3(b), dim3(th), 0, 0, o, t, it); }, 9, blocks, 256, 2048.0);
    run("16x16x4 f32, 2 acc", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k16<2>, dim3(b), dim3(th), 0, 0, o, t, it); }, 2, blocks, 256, 2048.0);
    run("16x16x4 f32, 1 acc", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k16<1>, dim3(b), dim3(th), 0, 0, o, t, it); }, 1, blocks, 256, 2048.0);
    run("32x32x2 f32, 4 acc", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(th), 0, 0, o, t, it); }, 4, blocks, 256, 4096.0);
    run("32x32x2 f32, 1 acc", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(th), 0, 0, o, t, it); }, 1, blocks, 256, 4096.0);
    run("16x16x4 f32, 9 acc x8w", [](float* o, unsigned long long* t, int it, int b, int th) { hipLaunchKernelGGL(k16<9>, dim3(b), dim3(th), 0, 0, o, t, it); }, 9, blocks, 512, 2048.0);
  }
  return 0;
}
