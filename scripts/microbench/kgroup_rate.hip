// Micro-benchmark: the network kernel's K-group MFMA block (net_dev.hpp mfma_group<FULL>) with operands
// in registers, no memory traffic: is the block itself issued at 32 cycles per MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../nuzero_amd/csrc/net_dev.hpp"
using namespace nz;

template <int OMASK>
__global__ void kg(float* out, unsigned long long* ticks, int iters) {
  Frag f0, f1;
  for (int i = 0; i < 9; ++i) {
    const float v = threadIdx.x * 0.001f + i;
    f0.a[i] = f32x4{v, v + 1, v + 2, v + 3};
    f0.b[i] = f32x4{v * 0.5f, v, v, v};
    f1.a[i] = f32x4{v + 0.25f, v + 1, v + 2, v + 3};
    f1.b[i] = f32x4{v * 0.75f, v, v, v};
  }
  f32x4 acc[CELLS];
  for (int o = 0; o < CELLS; ++o) acc[o] = f32x4{0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    mfma_group<OMASK>(acc, f0);
    mfma_group<OMASK>(acc, f1);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int o = 0; o < CELLS; ++o) s += acc[o][0] + acc[o][1] + acc[o][2] + acc[o][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* ticks;
  (void)hipMalloc(&out, 256 * 256 * sizeof(float));
  (void)hipMalloc(&ticks, 256 * sizeof(unsigned long long));
  const int iters = 50;
  for (int blocks : {1, 256}) {
    hipLaunchKernelGGL(kg<0x1FF>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    printf("FULL group   blocks %3d: %.2f ticks/MFMA\n", blocks, h / (double)(iters * 2 * 196));
    hipLaunchKernelGGL(kg<0x011>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    printf("quarter {4,0} blocks %3d: %.2f ticks/MFMA\n", blocks, h / (double)(iters * 2 * 52));
  }
  return 0;
}
