// Shared device/host helpers for libcontrad_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CONTRAD_WAVE 64

#define CONTRAD_CHECK_LAUNCH()                       \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

#define CONTRAD_ARG(cond)                            \
  do {                                               \
    if (!(cond)) return -22; /* -EINVAL */           \
  } while (0)

// Development switches (tile modes, split-K on/off, forced tiles ...; tools/README.md) exist only in the variant built with
// -DCONTRAD_DEV_SWITCHES (libcontrad_hip_dev.so, contrad_amd/build.py; tools/build_variant.sh): the shipped library never
// reads its environment, so an integrator's environment cannot change a launch plan (no hidden state).
#ifdef CONTRAD_DEV_SWITCHES
#include <stdlib.h>
static inline const char* contrad_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* contrad_dev_env(const char*) { return nullptr; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// MI355X: 256 CUs in 8 XCDs; the dispatcher is observed to place block b on XCD b % 8.  Remap the
// linear block id so that each XCD (own 4 MiB L2) works on a contiguous run of tiles.  Bijective for
// any block count (cdna_hip_programming.md 5: "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int xcd = b & 7;
  const int q = nb >> 3, r = nb & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
