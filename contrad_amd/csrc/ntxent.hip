// Fused contrastive losses for gfx950: NT-Xent (training/criterion.py:24-45) and the fake-anchored
// supervised-contrastive term (training/gan/contrad.py:8-32), forward and backward, without ever
// materialising the R x R similarity matrix in HBM.
//
//   Z [R x D]  L2-normalised embeddings (R = 2N for NT-Xent, 3N for SupCon)
//   S = Z Z^T * inv_temp, diagonal forced to -5e4 (the reference's fill_diagonal_)
//   forward : lse_i = logsumexp_j S_ij  (online max/sum kept per lane, merged with wave shuffles),
//             t_i   = sum_j T_ij S_ij   (T = one-hot positive / uniform over other fakes),
//             loss  = sum_{anchors} c * (lse_i - t_i)
//   backward: dZ_i  = inv_temp * sum_j (G_ij + G_ji) z_j,  G_ij = c * (exp(S_ij - lse_i) - T_ij) for anchor i
//             (S is recomputed tile by tile; the weight tile goes through LDS into a second MFMA GEMM).
//
// Both GEMMs run on v_mfma_f32_32x32x2_f32.  A block owns 64 rows and every S-th 64-column tile (grid = row tiles x
// S column splits, S chosen so that ~256 blocks exist: 2N = 1024 rows alone would occupy 16 of the 256 CUs).  The
// per-split partial (max, sum, target) triples / partial dZ slabs go to a workspace and are merged in a fixed order.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

constexpr int RT = 64;  // row tile
constexpr int CT = 64;  // column tile
constexpr float DIAG = -5e4f;

struct ContrastArgs {
  const float* z;
  int R, D, N, mode;  // mode 0: NT-Xent (R = 2N), 1: SupCon on fakes (R = 3N)
  float inv_temp;
};

__device__ __forceinline__ bool is_anchor(const ContrastArgs& a, int i) {
  return a.mode == 0 ? (i < a.R) : (i >= 2 * a.N && i < a.R);
}
// target weight T_ij for anchor i
__device__ __forceinline__ float target_w(const ContrastArgs& a, int i, int j, float sup_w) {
  if (a.mode == 0) {
    const int pos = (i < a.N) ? i + a.N : i - a.N;
    return j == pos ? 1.f : 0.f;
  }
  return (j >= 2 * a.N && j != i) ? sup_w : 0.f;
}

// Stage rows [r0, r0+64) of z into LDS as dst[row][DP+1] (zero padded in both dims).
template <int DP>
__device__ __forceinline__ void stage_rows(float* dst, const float* z, int r0, int R, int D) {
  constexpr int LD = DP + 1;
  for (int e = threadIdx.x; e < 64 * (DP / 4); e += blockDim.x) {
    const int row = e / (DP / 4), d4 = (e % (DP / 4)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int r = r0 + row;
    if (r < R) {
      if (d4 + 3 < D && (D & 3) == 0) {
        const float4 t = *reinterpret_cast<const float4*>(z + (size_t)r * D + d4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (d4 + j < D) v[j] = z[(size_t)r * D + d4 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[row * LD + d4 + j] = v[j];
  }
}

// One 32x32 tile of S = Ar(rows) * Zr(cols)^T over the padded feature dim.
template <int DP>
__device__ __forceinline__ f32x16 s_tile(const float* Ar, const float* Zr, int rbase, int cbase, int lane) {
  constexpr int LD = DP + 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll 8
  for (int ks = 0; ks < DP / 2; ++ks) {
    const int k = ks * 2 + lhi;
    const float a = Ar[(rbase + l31) * LD + k];
    const float b = Zr[(cbase + l31) * LD + k];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}

template <int DP>
__global__ __launch_bounds__(256) void contrast_fwd_kernel(ContrastArgs a, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DP + 1;
  float* Ar = smem;
  float* Zr = Ar + RT * LD;
  float* red = Zr + CT * LD;  // [2 col-waves][64 rows][3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * RT;
  const float sup_w = (a.mode == 1) ? 1.f / (float)(a.N - 1) : 0.f;

  stage_rows<DP>(Ar, a.z, r0, a.R, a.D);

  float m[16], s[16], tsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { m[r] = -INFINITY; s[r] = 0.f; tsum[r] = 0.f; }

  for (int c0 = blockIdx.y * CT; c0 < a.R; c0 += gridDim.y * CT) {
    __syncthreads();
    stage_rows<DP>(Zr, a.z, c0, a.R, a.D);
    __syncthreads();
    const f32x16 acc = s_tile<DP>(Ar, Zr, wr * 32, wc * 32, lane);
    const int j = c0 + wc * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = r0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (j < a.R && i < a.R) {
        float v = acc[r] * a.inv_temp;
        if (i == j) v = DIAG;
        const float mn = fmaxf(m[r], v);
        s[r] = s[r] * __expf(m[r] - mn) + __expf(v - mn);
        m[r] = mn;
        tsum[r] += target_w(a, i, j, sup_w) * v;
      }
    }
  }
  // merge the per-lane (m, s, t) over the 32 lanes that share a row, then over the two column waves
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float mo = __shfl_xor(m[r], o, 64), so = __shfl_xor(s[r], o, 64), to = __shfl_xor(tsum[r], o, 64);
      const float mn = fmaxf(m[r], mo);
      const float sa = (m[r] == -INFINITY) ? 0.f : s[r] * __expf(m[r] - mn);
      const float sb = (mo == -INFINITY) ? 0.f : so * __expf(mo - mn);
      s[r] = sa + sb;
      m[r] = mn;
      tsum[r] += to;
    }
  }
  __syncthreads();
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float* q = red + (wc * RT + row) * 3;
      q[0] = m[r]; q[1] = s[r]; q[2] = tsum[r];
    }
  }
  __syncthreads();
  if (tid < RT) {
    const int i = r0 + tid;
    if (i < a.R) {
      const float* q0 = red + tid * 3;
      const float* q1 = red + (RT + tid) * 3;
      const float mn = fmaxf(q0[0], q1[0]);
      const float sa = (q0[0] == -INFINITY) ? 0.f : q0[1] * __expf(q0[0] - mn);
      const float sb = (q1[0] == -INFINITY) ? 0.f : q1[1] * __expf(q1[0] - mn);
      float* o = part + ((size_t)blockIdx.y * a.R + i) * 3;   // this split's (max, sum exp, target) of row i
      o[0] = mn; o[1] = sa + sb; o[2] = q0[2] + q1[2];
    }
  }
}

// Merge the S column splits of every row (fixed order), emit lse / rowloss, and reduce
// loss = c * sum_i rowloss[i] (single block -> one fixed summation order).
// (one block of 1024 threads: one or two rows per thread.  With 256 threads every thread walked 4 - 6 rows x 2 S dependent
// load groups one after the other: 30 us for 200 KB, pure latency.)
__global__ __launch_bounds__(1024) void contrast_loss_reduce_kernel(ContrastArgs a, const float* __restrict__ part, int S, float c,
                                            float* __restrict__ lse_out, float* __restrict__ rowloss_out,
                                            float* __restrict__ loss_out) {
  __shared__ float red[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < a.R; i += blockDim.x) {
    float mn = -INFINITY;
    for (int s = 0; s < S; ++s) mn = fmaxf(mn, part[((size_t)s * a.R + i) * 3]);
    float sum = 0.f, t = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* q = part + ((size_t)s * a.R + i) * 3;
      if (q[0] != -INFINITY) sum += q[1] * __expf(q[0] - mn);
      t += q[2];
    }
    const float lse = mn + __logf(sum);
    lse_out[i] = lse;
    const float rl = is_anchor(a, i) ? (lse - t) : 0.f;
    rowloss_out[i] = rl;
    v += rl;
  }
  v = block_sum(v, red);
  if (threadIdx.x == 0) loss_out[0] = v * c;
}

template <int DP>
__global__ __launch_bounds__(256) void contrast_bwd_kernel(ContrastArgs a, const float* __restrict__ lse,
                                                           float coef, const float* __restrict__ gscale,
                                                           float* __restrict__ dz, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DP + 1;
  constexpr int LDW = RT + 1;
  constexpr int NT = DP / 64;  // 32-wide d-tiles per wave in the second GEMM
  float* Ar = smem;
  float* Zr = Ar + RT * LD;
  float* Ws = Zr + CT * LD;    // [CT cols][RT rows + 1]
  float* lse_r = Ws + CT * LDW;  // [64]
  float* lse_c = lse_r + RT;     // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int r0 = blockIdx.x * RT;
  const float sup_w = (a.mode == 1) ? 1.f / (float)(a.N - 1) : 0.f;
  const float scale = coef * a.inv_temp * (gscale ? gscale[0] : 1.f);

  stage_rows<DP>(Ar, a.z, r0, a.R, a.D);
  if (tid < RT) lse_r[tid] = (r0 + tid < a.R) ? lse[r0 + tid] : 0.f;

  f32x16 dacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[t][r] = 0.f;

  for (int c0 = blockIdx.y * CT; c0 < a.R; c0 += gridDim.y * CT) {
    __syncthreads();
    stage_rows<DP>(Zr, a.z, c0, a.R, a.D);
    if (tid < CT) lse_c[tid] = (c0 + tid < a.R) ? lse[c0 + tid] : 0.f;
    __syncthreads();
    const f32x16 acc = s_tile<DP>(Ar, Zr, wr * 32, wc * 32, lane);
    const int jl = wc * 32 + l31, j = c0 + jl;
    const bool janchor = is_anchor(a, j);
    const float lj = lse_c[jl];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, i = r0 + il;
      float w = 0.f;
      if (i < a.R && j < a.R && i != j) {
        const float v = acc[r] * a.inv_temp;
        if (is_anchor(a, i)) w += __expf(v - lse_r[il]) - target_w(a, i, j, sup_w);
        if (janchor) w += __expf(v - lj) - target_w(a, j, i, sup_w);
      }
      Ws[jl * LDW + il] = w;
    }
    __syncthreads();
    // dZ[rows wr*32.., d-tiles] += W[rows, cols] * Zr[cols, d]
#pragma unroll 4
    for (int ks = 0; ks < CT / 2; ++ks) {
      const int k = ks * 2 + lhi;
      const float av = Ws[k * LDW + wr * 32 + l31];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = Zr[k * LD + (wc + 2 * t) * 32 + l31];
        dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dacc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int dcol = (wc + 2 * t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = r0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (i < a.R && dcol < a.D) {
        if (gridDim.y == 1) dz[(size_t)i * a.D + dcol] = dacc[t][r] * scale;
        else part[((size_t)blockIdx.y * a.R + i) * a.D + dcol] = dacc[t][r];   // unscaled slab of this split
      }
    }
  }
}

// dz = scale * sum_s part[s]  (fixed order)
__global__ void contrast_bwd_reduce_kernel(const float* __restrict__ part, int S, long long n, float coef_temp,
                                           const float* __restrict__ gscale, float* __restrict__ dz) {
  const float scale = coef_temp * (gscale ? gscale[0] : 1.f);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += part[(size_t)s * n + e];
    dz[e] = v * scale;
  }
}

// ---- row L2 normalisation (F.normalize, contrad.py:43,48): one wave per row ----
__global__ void l2norm_fwd_kernel(const float* __restrict__ u, int ldu, float* __restrict__ z,
                                  float* __restrict__ invn, int R, int D, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = u[(size_t)row * ldu + d]; ss += v * v; }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  for (int d = lane; d < D; d += 64) z[(size_t)row * D + d] = u[(size_t)row * ldu + d] * inv;
  if (lane == 0) invn[row] = inv;
}

// du = (dz - z * <z, dz>) * inv_norm   (rows whose norm was clamped by eps are not on this path)
__global__ void l2norm_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                  const float* __restrict__ invn, float* __restrict__ du, int ldu, int R,
                                  int D, int accumulate) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += dz[(size_t)row * D + d] * z[(size_t)row * D + d];
  dot = wave_sum(dot);
  const float inv = invn[row];
  for (int d = lane; d < D; d += 64) {
    const float g = (dz[(size_t)row * D + d] - z[(size_t)row * D + d] * dot) * inv;
    float* o = du + (size_t)row * ldu + d;
    *o = accumulate ? (*o + g) : g;
  }
}

template <int DP>
size_t fwd_smem() { return (size_t)(2 * 64 * (DP + 1) + 2 * 64 * 3) * sizeof(float); }
template <int DP>
size_t bwd_smem() { return (size_t)(2 * 64 * (DP + 1) + 64 * 65 + 128) * sizeof(float); }

template <int DP>
int launch_fwd(const ContrastArgs& a, int S, float* part, hipStream_t s) {
  static bool set = false;
  if (!set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&contrast_fwd_kernel<DP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_smem<DP>());
    if (e != hipSuccess) return (int)e;
    set = true;
  }
  hipLaunchKernelGGL((contrast_fwd_kernel<DP>), dim3(cdiv(a.R, RT), S), dim3(256), fwd_smem<DP>(), s, a, part);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
template <int DP>
int launch_bwd(const ContrastArgs& a, int S, const float* lse, float coef, const float* gscale, float* dz,
               float* part, hipStream_t s) {
  static bool set = false;
  if (!set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&contrast_bwd_kernel<DP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_smem<DP>());
    if (e != hipSuccess) return (int)e;
    set = true;
  }
  hipLaunchKernelGGL((contrast_bwd_kernel<DP>), dim3(cdiv(a.R, RT), S), dim3(256), bwd_smem<DP>(), s, a, lse,
                     coef, gscale, dz, part);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

int check(int R, int D, int N, int mode) {
  CONTRAD_ARG(R > 0 && D > 0 && D <= 256 && N > 0);
  CONTRAD_ARG(mode == 0 || mode == 1);
  CONTRAD_ARG(mode == 0 ? (R == 2 * N) : (R == 3 * N));
  return 0;
}
float anchor_coef(int N, int mode) { return mode == 0 ? 1.f / (2.f * N) : 1.f / (float)N; }
// column splits: ~256 blocks in total, never more splits than column tiles
int splits(int R) {
  const int tiles = cdiv(R, RT);
  int S = 256 / tiles;
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return S;
}

}  // namespace

extern "C" long long contrad_contrast_workspace_bytes(int R, int D) {
  if (R <= 0 || D <= 0) return -22;
  const long long per = (long long)R * (D > 3 ? D : 3);
  return (long long)splits(R) * per * (long long)sizeof(float);
}

extern "C" int contrad_contrast_fwd(const float* z, int R, int D, int N, int mode, float inv_temp,
                                    float* lse, float* rowloss, float* loss, float* workspace,
                                    long long workspace_bytes, contrad_stream_t stream) {
  int rc = check(R, D, N, mode);
  if (rc) return rc;
  CONTRAD_ARG(z && lse && rowloss && loss && workspace);
  CONTRAD_ARG(workspace_bytes >= contrad_contrast_workspace_bytes(R, D));
  ContrastArgs a{z, R, D, N, mode, inv_temp};
  hipStream_t s = (hipStream_t)stream;
  const int S = splits(R);
  if (D <= 64) rc = launch_fwd<64>(a, S, workspace, s);
  else if (D <= 128) rc = launch_fwd<128>(a, S, workspace, s);
  else rc = launch_fwd<256>(a, S, workspace, s);
  if (rc) return rc;
  hipLaunchKernelGGL(contrast_loss_reduce_kernel, dim3(1), dim3(1024), 0, s, a, workspace, S, anchor_coef(N, mode),
                     lse, rowloss, loss);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_contrast_bwd(const float* z, const float* lse, int R, int D, int N, int mode,
                                    float inv_temp, const float* grad_scale, float* dz, float* workspace,
                                    long long workspace_bytes, contrad_stream_t stream) {
  int rc = check(R, D, N, mode);
  if (rc) return rc;
  CONTRAD_ARG(z && lse && dz && workspace);
  CONTRAD_ARG(workspace_bytes >= contrad_contrast_workspace_bytes(R, D));
  ContrastArgs a{z, R, D, N, mode, inv_temp};
  hipStream_t s = (hipStream_t)stream;
  const float c = anchor_coef(N, mode);
  const int S = splits(R);
  if (D <= 64) rc = launch_bwd<64>(a, S, lse, c, grad_scale, dz, workspace, s);
  else if (D <= 128) rc = launch_bwd<128>(a, S, lse, c, grad_scale, dz, workspace, s);
  else rc = launch_bwd<256>(a, S, lse, c, grad_scale, dz, workspace, s);
  if (rc || S == 1) return rc;
  const long long n = (long long)R * D;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(contrast_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, s, workspace, S, n, c * inv_temp,
                     grad_scale, dz);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_l2norm_fwd(const float* u, int ldu, float* z, float* inv_norm, int R, int D,
                                  float eps, contrad_stream_t stream) {
  CONTRAD_ARG(u && z && inv_norm && R > 0 && D > 0 && ldu >= D);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, u, ldu, z,
                     inv_norm, R, D, eps);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_l2norm_bwd(const float* dz, const float* z, const float* inv_norm, float* du,
                                  int ldu, int R, int D, int accumulate, contrad_stream_t stream) {
  CONTRAD_ARG(dz && z && inv_norm && du && R > 0 && D > 0 && ldu >= D);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dz, z,
                     inv_norm, du, ldu, R, D, accumulate);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
