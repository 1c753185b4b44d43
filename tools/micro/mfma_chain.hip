// Dev micro-benchmark: fp32 MFMA throughput of one MI355X as a function of (a) independent accumulator tiles per wave
// (1 = every v_mfma_f32_32x32x2_f32 depends on the previous one, what a 32 x 32 wave tile does) and (b) waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/micro/mfma_chain.hip ; run: /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  extern __shared__ float pad[];      // dynamic LDS only to set the number of resident blocks per CU
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a0 = a + threadIdx.x * 0.01f, b0 = b;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (a == 12345.f) pad[threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu * 4, iters = 4096;
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : blocks_per_cu == 2 ? 60 * 1024 : 30 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), lds, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), lds, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("accumulators per wave %d, waves per SIMD %d: %8.3f ms  %7.1f TF/s\n", NACC, blocks_per_cu, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int bpc : {1, 2, 4}) { run<1>(bpc); run<2>(bpc); run<4>(bpc); }
  return 0;
}
