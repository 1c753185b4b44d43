// Dev micro-benchmark: the fp32 MFMA issue ceiling of one MI355X with the conv engine's wave shape
// (4 waves per block, 2x2 tiles of v_mfma_f32_32x32x2_f32 per wave), with and without the LDS fragment reads.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREAD>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters) {
  __shared__ float sm[2 * 16 * 132];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 16 * 132; i += 256) sm[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a0 = lane * 0.01f, a1 = lane * 0.02f, b0 = 1.f, b1 = 2.f;
  const float* As = sm + (lane & 31);
  const float* Bs = sm + 16 * 132 + (lane & 31);
  const int lhi = lane >> 5;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (LDSREAD) {
        const int kk = ks * 2 + lhi;
        a0 = As[kk * 132]; a1 = As[kk * 132 + 32];
        b0 = Bs[kk * 132]; b1 = Bs[kk * 132 + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (LDSREAD == 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int L>
void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flop = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
  printf("%-28s blocks %5d  %8.3f ms  %7.1f TF/s\n", name, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1) {   // short-lived blocks, like the conv engine's (blocks, K-tiles per block)
    const int cfg[5][2] = {{3072, 72}, {1536, 144}, {768, 288}, {1024, 216}, {30720, 72}};
    for (auto& c : cfg) {
      run<1>("mfma + lds frag reads", c[0], c[1]);
      run<2>("mfma + lds reads + barrier", c[0], c[1]);
    }
    return 0;
  }
  for (int blocks : {256, 512, 1024, 4096}) {
    run<0>("mfma only", blocks, 4096 * 256 / blocks * 4);
    run<1>("mfma + lds frag reads", blocks, 4096 * 256 / blocks * 4);
    run<2>("mfma + lds reads + barrier", blocks, 4096 * 256 / blocks * 4);
  }
  return 0;
}
