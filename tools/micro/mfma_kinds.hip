// Dev micro-benchmark: how much fp32 MFMA throughput do VALU instructions issued between MFMA groups cost?
// (4 waves per block, 4 blocks per CU, 2x2 tiles of v_mfma_f32_32x32x2_f32 per wave = the conv engine's shape.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NVALU, int KIND>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, int seed) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a0 = lane * 0.01f, a1 = lane * 0.02f, b0 = 1.f, b1 = 2.f;
  int v[8]; int sreg = seed; long long v64[4] = {tid, tid + 1, tid + 2, tid + 3}; long long sbase = (long long)out;
  for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + tid;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int n = 0; n < NVALU; ++n) { if (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(v[(n + 3) & 7]));
        if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
        if (KIND == 2) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(v[n & 7]));
        if (KIND == 3) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(v64[n & 3]) : "s"(sbase)); }
      __builtin_amdgcn_sched_barrier(0);
      
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  int q = 0;
  for (int i = 0; i < 8; ++i) q ^= v[i]; q ^= sreg; for (int i = 0; i < 4; ++i) q ^= (int)v64[i];
  out[blockIdx.x * 256 + tid] = s + (float)q;
}

template <int NV, int KIND>
void run(int blocks, int iters) {
  float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 3);
  (void)hipEventRecord(e0);
  const int reps = 3;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flop = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
  printf("instr/kstep %3d kind %d blocks %5d  %8.3f ms  %7.1f TF/s\n", NV, KIND, blocks, ms, flop / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  const int blocks = 1024, iters = 4096;
  run<0, 0>(blocks, iters);
  run<16, 0>(blocks, iters);
  run<16, 1>(blocks, iters);
  run<32, 1>(blocks, iters);
  run<16, 2>(blocks, iters);
  run<16, 3>(blocks, iters);
  run<8, 3>(blocks, iters);
  return 0;
}
