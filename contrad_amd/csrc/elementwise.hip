// HBM-bound helpers of the discriminator step: column statistics (bias gradients, BatchNorm batch statistics),
// BatchNorm normalise+ReLU for the generator forward, the GAN logit losses with their gradients, fused Adam.
// All reductions are two-stage with a fixed summation order (deterministic); wave-level sums use shuffles.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

// ---- column sums of a row-major [M][ld] matrix (K columns): per-block partials, then fixed-order reduce ----
// thread (tx = tid & 63, ty = tid >> 6): columns tx + 64 j, rows r0 + ty, r0 + ty + 4, ...
constexpr int COLS_PER_THREAD = 8;  // K <= 512 per pass (column blocks over blockIdx.y)

template <bool SQ>
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ x, long long M, int K,
                                                               int ld, int rows_per_block,
                                                               float* __restrict__ partial) {
  __shared__ float red[4][64 * COLS_PER_THREAD + 1];
  __shared__ float red2[4][64 * COLS_PER_THREAD + 1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int cbase = blockIdx.y * 64 * COLS_PER_THREAD;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float s[COLS_PER_THREAD], q[COLS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < COLS_PER_THREAD; ++j) { s[j] = 0.f; q[j] = 0.f; }
  for (long long r = r0 + ty; r < r1; r += 4) {
    const float* row = x + r * ld;
#pragma unroll
    for (int j = 0; j < COLS_PER_THREAD; ++j) {
      const int c = cbase + tx + 64 * j;
      if (c < K) {
        const float v = row[c];
        s[j] += v;
        if (SQ) q[j] = fmaf(v, v, q[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COLS_PER_THREAD; ++j) {
    red[ty][tx + 64 * j] = s[j];
    if (SQ) red2[ty][tx + 64 * j] = q[j];
  }
  __syncthreads();
  // partial layout: [gridDim.x][SQ ? 2 : 1][K]
  const int nstat = SQ ? 2 : 1;
  for (int e = threadIdx.x; e < 64 * COLS_PER_THREAD; e += blockDim.x) {
    const int c = cbase + e;
    if (c < K) {
      partial[((size_t)blockIdx.x * nstat + 0) * K + c] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
      if (SQ) partial[((size_t)blockIdx.x * nstat + 1) * K + c] = red2[0][e] + red2[1][e] + red2[2][e] + red2[3][e];
    }
  }
}

// float4 variant (K % 4 == 0, ld % 4 == 0): CQB column quads x (256 / CQB) rows per pass, 4 independent row loads in
// flight per thread -- the scalar kernel above walks its rows one dependent 4-byte load at a time and was latency-bound
// (91 us for the 134 MB of G_SNDCGAN's last BatchNorm input; this one streams it).  Same partial layout.
template <bool SQ, int CQB>
__global__ __launch_bounds__(256) void colstats_partial_vec_kernel(const float* __restrict__ x, long long M, int K,
                                                                   int ld, int rows_per_block,
                                                                   float* __restrict__ partial) {
  constexpr int RPP = 256 / CQB;
  __shared__ float4 red[RPP][CQB];
  __shared__ float4 red2[SQ ? RPP : 1][CQB];
  const int cq = threadIdx.x % CQB, rr = threadIdx.x / CQB;
  const int c = (blockIdx.y * CQB + cq) * 4;
  const bool cok = c < K;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  auto add = [&](const float4& v) {
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    if (SQ) { q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w); }
  };
  if (cok) {
    long long r = r0 + rr;
    for (; r + 3 * RPP < r1; r += 4 * RPP) {
      const float4 v0 = *reinterpret_cast<const float4*>(x + r * ld + c);
      const float4 v1 = *reinterpret_cast<const float4*>(x + (r + RPP) * ld + c);
      const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * RPP) * ld + c);
      const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * RPP) * ld + c);
      add(v0); add(v1); add(v2); add(v3);
    }
    for (; r < r1; r += RPP) add(*reinterpret_cast<const float4*>(x + r * ld + c));
  }
  red[rr][cq] = s;
  if (SQ) red2[rr][cq] = q;
  __syncthreads();
  const int nstat = SQ ? 2 : 1;
  if (rr == 0 && cok) {   // fixed order over the RPP row groups
    float4 t = red[0][cq];
    for (int g = 1; g < RPP; ++g) { const float4 u = red[g][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * nstat + 0) * K + c) = t;
    if (SQ) {
      float4 t2 = red2[0][cq];
      for (int g = 1; g < RPP; ++g) { const float4 u = red2[g][cq]; t2.x += u.x; t2.y += u.y; t2.z += u.z; t2.w += u.w; }
      *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * nstat + 1) * K + c) = t2;
    }
  }
}

// out[s][c] = sum_b partial[b][s][c]  (optionally accumulated onto out).  Block = 16 entries x 16 partial lanes: with up
// to 512 partial rows and only 2K <= 1024 entries the launch is pure latency -- 64 entries x 4 lanes walked 128 strided
// loads per thread (14 us, six times per step); 16 lanes walk 32, four independent chains each.  Fixed summation order.
constexpr int CSR_E = 16, CSR_P = 16;
__global__ __launch_bounds__(256) void colstats_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                              int nstat, int K, float* __restrict__ out,
                                                              int accumulate) {
  __shared__ float red[CSR_P][CSR_E + 1];
  const int ex = threadIdx.x % CSR_E, py = threadIdx.x / CSR_E;
  const int e = blockIdx.x * CSR_E + ex;
  const int E = nstat * K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < E) {
    int b = py;
    for (; b + 3 * CSR_P < nblocks; b += 4 * CSR_P) {
      s0 += partial[(size_t)b * E + e];
      s1 += partial[(size_t)(b + CSR_P) * E + e];
      s2 += partial[(size_t)(b + 2 * CSR_P) * E + e];
      s3 += partial[(size_t)(b + 3 * CSR_P) * E + e];
    }
    for (; b < nblocks; b += CSR_P) s0 += partial[(size_t)b * E + e];
  }
  red[py][ex] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (py == 0 && e < E) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < CSR_P; ++r) s += red[r][ex];
    out[e] = accumulate ? out[e] + s : s;
  }
}

// ---- BatchNorm (training mode) finalise + apply + ReLU ----
// stats[0][c] = sum x, stats[1][c] = sum x^2 over `count` rows (already reduced over ranks for SyncBN).
// conv_bias (may be NULL) is the bias of the producing layer: BN(x + b) == BN(x) up to the running mean,
// so it is folded into the running_mean update only.  perm_hw > 1 writes column c*perm_hw + hw of the input
// to NHWC position (hw, c)  (G_SNDCGAN's linear -> norm_init -> view(-1, 512, 4, 4), sndcgan.py:42-45).
__global__ void bn_relu_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long M, int K,
                                     int ldx, int ldy, const float* __restrict__ stats, float count,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, int perm_hw) {
  const long long total = M * K;
  const float inv_n = 1.f / count;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / K;
    const int c = (int)(e - r * K);
    const float mean = stats[c] * inv_n;
    const float var = fmaxf(stats[K + c] * inv_n - mean * mean, 0.f);
    const float v = (x[r * ldx + c] - mean) * rsqrtf(var + eps) * gamma[c] + beta[c];
    const float o = v > 0.f ? v : 0.f;
    if (perm_hw > 1) {
      const int ch = c / perm_hw, hw = c - ch * perm_hw;
      y[r * ldy + (size_t)hw * (K / perm_hw) + ch] = o;
    } else {
      y[r * ldy + c] = o;
    }
  }
}

// ---- BatchNorm(train) + ReLU backward ----
// pass 1 (two-stage, deterministic): per channel S1 = sum dyM, S2 = sum dyM * xhat, dyM = dy * [y > 0]
// thread (tx, ty) layout as colstats_partial_kernel
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ x, long long M, int K,
                                                                int ld, const float* __restrict__ stats, float count,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[4][64 * COLS_PER_THREAD + 1];
  __shared__ float red2[4][64 * COLS_PER_THREAD + 1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int cbase = blockIdx.y * 64 * COLS_PER_THREAD;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float s[COLS_PER_THREAD], q[COLS_PER_THREAD], mu[COLS_PER_THREAD], rs[COLS_PER_THREAD], ga[COLS_PER_THREAD],
      be[COLS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < COLS_PER_THREAD; ++j) {
    s[j] = 0.f; q[j] = 0.f;
    const int c = cbase + tx + 64 * j;
    if (c < K) {
      mu[j] = stats[c] / count;
      rs[j] = rsqrtf(fmaxf(stats[K + c] / count - mu[j] * mu[j], 0.f) + eps);
      ga[j] = gamma[c]; be[j] = beta[c];
    } else { mu[j] = 0.f; rs[j] = 0.f; ga[j] = 0.f; be[j] = 0.f; }
  }
  for (long long r = r0 + ty; r < r1; r += 4) {
#pragma unroll
    for (int j = 0; j < COLS_PER_THREAD; ++j) {
      const int c = cbase + tx + 64 * j;
      if (c < K) {
        const float xh = (x[r * ld + c] - mu[j]) * rs[j];
        const float g = (xh * ga[j] + be[j] > 0.f) ? dy[r * ld + c] : 0.f;
        s[j] += g;
        q[j] = fmaf(g, xh, q[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COLS_PER_THREAD; ++j) { red[ty][tx + 64 * j] = s[j]; red2[ty][tx + 64 * j] = q[j]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * COLS_PER_THREAD; e += blockDim.x) {
    const int c = cbase + e;
    if (c < K) {
      partial[((size_t)blockIdx.x * 2 + 0) * K + c] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
      partial[((size_t)blockIdx.x * 2 + 1) * K + c] = red2[0][e] + red2[1][e] + red2[2][e] + red2[3][e];
    }
  }
}

// pass 2: dx = gamma * rstd * (dyM - S1/n - xhat * S2/n)
__global__ void bn_relu_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                         float* __restrict__ dx, long long M, int K, int ld,
                                         const float* __restrict__ stats, float count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                         const float* __restrict__ bstats) {
  const long long total = M * K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / K;
    const int c = (int)(e - r * K);
    const float mean = stats[c] / count;
    const float rstd = rsqrtf(fmaxf(stats[K + c] / count - mean * mean, 0.f) + eps);
    const float xh = (x[r * ld + c] - mean) * rstd;
    const float g = (xh * gamma[c] + beta[c] > 0.f) ? dy[r * ld + c] : 0.f;
    dx[r * ld + c] = gamma[c] * rstd * (g - bstats[c] / count - xh * bstats[K + c] / count);
  }
}

__global__ void bn_running_update_kernel(const float* __restrict__ stats, float count, int K,
                                         const float* __restrict__ conv_bias, float momentum,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;      // (nn.BatchNorm2d's counter: was an ATen launch per layer)
  if (c >= K) return;
  const float mean = stats[c] / count;
  const float var = fmaxf(stats[K + c] / count - mean * mean, 0.f);
  const float unbiased = (count > 1.f) ? var * (count / (count - 1.f)) : var;
  const float m = mean + (conv_bias ? conv_bias[c] : 0.f);
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
}

// ---- discriminator GAN losses on the logits d[3N] (rows [0,N) real view 1, [2N,3N) fakes) ----
// contrad.loss_D_fn, training/gan/contrad.py:51-64.  out[0]=loss, out[1]=mean d_real, out[2]=mean d_gen;
// grad[3N] = d loss / d logits (view-2 logits get 0).  Single block; N is a per-rank batch (<= a few 1000).
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void gan_d_loss_kernel(const float* __restrict__ d, int ldd, int N, int kind,
                                  float* __restrict__ out, float* __restrict__ grad) {
  __shared__ float red[16];
  float l = 0.f, sr = 0.f, sg = 0.f;
  const float invN = 1.f / (float)N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float r = d[(size_t)i * ldd], g = d[(size_t)(2 * N + i) * ldd];
    float gr, gg;
    if (kind == 0) {          // nonsat: softplus(d_gen) + softplus(-d_real)
      l += softplus_f(g) + softplus_f(-r);
      gg = sigmoid_f(g); gr = -sigmoid_f(-r);
    } else if (kind == 1) {   // wgan
      l += g - r; gg = 1.f; gr = -1.f;
    } else if (kind == 2) {   // hinge
      l += fmaxf(1.f + g, 0.f) + fmaxf(1.f - r, 0.f);
      gg = (1.f + g > 0.f) ? 1.f : 0.f; gr = (1.f - r > 0.f) ? -1.f : 0.f;
    } else {                  // lsgan
      l += 0.5f * ((r - 1.f) * (r - 1.f) + g * g);
      gr = (r - 1.f); gg = g;
    }
    sr += r; sg += g;
    grad[i] = gr * invN;
    grad[N + i] = 0.f;
    grad[2 * N + i] = gg * invN;
  }
  l = block_sum(l, red);
  sr = block_sum(sr, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) { out[0] = l * invN; out[1] = sr * invN; out[2] = sg * invN; }
}

// generator-side loss on d[N] (contrad.loss_G_fn, contrad.py:73-82): kind 0 nonsat softplus(-d), 3 lsgan, else -d
__global__ void gan_g_loss_kernel(const float* __restrict__ d, int ldd, int N, int kind,
                                  float* __restrict__ out, float* __restrict__ grad) {
  __shared__ float red[16];
  float l = 0.f;
  const float invN = 1.f / (float)N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float g = d[(size_t)i * ldd];
    float gg;
    if (kind == 0) { l += softplus_f(-g); gg = -sigmoid_f(-g); }
    else if (kind == 3) { l += 0.5f * (g - 1.f) * (g - 1.f); gg = (g - 1.f); }
    else { l += -g; gg = -1.f; }
    grad[i] = gg * invN;
  }
  l = block_sum(l, red);
  if (threadIdx.x == 0) out[0] = l * invN;
}

// ---- fused multi-tensor Adam (torch.optim.Adam semantics: no weight decay, no amsgrad) ----
__global__ void adam_kernel(contrad_adam_batch b, float step_size, float beta1, float beta2,
                            float inv_sqrt_bc2, float eps, float grad_scale, const float* __restrict__ hyper) {
  if (hyper) {   // step-dependent scalars from device memory: a captured hipGraph replays with each step's values
    step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; grad_scale = hyper[2];
  }
  // block -> (tensor, chunk)
  int t = 0;
  while (t + 1 < b.n && (int)blockIdx.x >= b.block_start[t + 1]) ++t;
  const contrad_adam_tensor& T = b.t[t];
  const long long chunk = (long long)(blockIdx.x - b.block_start[t]) * CONTRAD_ADAM_CHUNK;
  const long long end = (chunk + CONTRAD_ADAM_CHUNK < T.numel) ? chunk + CONTRAD_ADAM_CHUNK : T.numel;
  for (long long i = chunk + threadIdx.x; i < end; i += blockDim.x) {
    const float g = T.g[i] * grad_scale;
    const float m = beta1 * T.m[i] + (1.f - beta1) * g;
    const float v = beta2 * T.v[i] + (1.f - beta2) * g * g;
    T.m[i] = m;
    T.v[i] = v;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    T.p[i] = T.p[i] - step_size * (m / denom);
  }
}

__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, long long n, float a, float bcoef) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = a * y[i] + bcoef * x[i];
}

// dst[i] = src[i]; src may be PINNED HOST memory (device-mapped): the per-step random draws are pulled over PCIe by
// this kernel instead of a runtime H2D copy (see contrad_amd/hostio.py for why)
__global__ void pull_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

int colstats_plan(long long M, int* rows_per_block, int* nblocks) {
  long long nb = (M + 31) / 32;    // >= 32 rows per block (a 512-row BatchNorm1d still spreads over 16 row blocks)
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  *rows_per_block = (int)((M + nb - 1) / nb);
  *nblocks = (int)((M + *rows_per_block - 1) / *rows_per_block);
  return 0;
}

}  // namespace

extern "C" long long contrad_colstats_workspace_bytes(long long M, int K, int with_sq) {
  int rpb, nb;
  colstats_plan(M, &rpb, &nb);
  return (long long)nb * (with_sq ? 2 : 1) * K * (long long)sizeof(float);
}

extern "C" int contrad_colstats(const float* x, long long M, int K, int ld, int with_sq, float* out,
                                int accumulate, float* workspace, long long workspace_bytes,
                                contrad_stream_t stream) {
  CONTRAD_ARG(x && out && workspace && M > 0 && K > 0 && ld >= K);
  CONTRAD_ARG(workspace_bytes >= contrad_colstats_workspace_bytes(M, K, with_sq));
  int rpb, nb;
  colstats_plan(M, &rpb, &nb);
  dim3 grid(nb, cdiv(K, 64 * COLS_PER_THREAD));
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (K & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)workspace & 15) == 0;
  if (vec) {
    const int cq = K / 4;
#define CONTRAD_COLSTATS_VEC(CQB)                                                                                   \
    do {                                                                                                            \
      dim3 g(nb, cdiv(cq, CQB));                                                                                    \
      if (with_sq) hipLaunchKernelGGL((colstats_partial_vec_kernel<true, CQB>), g, dim3(256), 0, s, x, M, K, ld, rpb, workspace); \
      else hipLaunchKernelGGL((colstats_partial_vec_kernel<false, CQB>), g, dim3(256), 0, s, x, M, K, ld, rpb, workspace);        \
    } while (0)
    if (cq <= 16) CONTRAD_COLSTATS_VEC(16);
    else if (cq <= 32) CONTRAD_COLSTATS_VEC(32);
    else if (cq <= 64) CONTRAD_COLSTATS_VEC(64);
    else if (cq <= 128) CONTRAD_COLSTATS_VEC(128);
    else CONTRAD_COLSTATS_VEC(256);
#undef CONTRAD_COLSTATS_VEC
  } else if (with_sq) {
    hipLaunchKernelGGL(colstats_partial_kernel<true>, grid, dim3(256), 0, s, x, M, K, ld, rpb, workspace);
  } else {
    hipLaunchKernelGGL(colstats_partial_kernel<false>, grid, dim3(256), 0, s, x, M, K, ld, rpb, workspace);
  }
  CONTRAD_CHECK_LAUNCH();
  const int nstat = with_sq ? 2 : 1;
  hipLaunchKernelGGL(colstats_reduce_kernel, dim3(cdiv(nstat * K, CSR_E)), dim3(256), 0, s, workspace, nb, nstat, K,
                     out, accumulate);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_bn_relu_apply(const float* x, float* y, long long M, int K, int ldx, int ldy,
                                     const float* stats, float count, const float* gamma, const float* beta,
                                     float eps, int perm_hw, contrad_stream_t stream) {
  CONTRAD_ARG(x && y && stats && gamma && beta && M > 0 && K > 0 && ldx >= K && ldy >= K && count >= 1.f);
  CONTRAD_ARG(perm_hw >= 1 && K % perm_hw == 0);
  long long grid = (M * K + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, y, M, K, ldx,
                     ldy, stats, count, gamma, beta, eps, perm_hw);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_bn_relu_bwd_stats(const float* dy, const float* x, long long M, int K, int ld,
                                         const float* stats, float count, const float* gamma, const float* beta,
                                         float eps, float* out2k, float* workspace, long long workspace_bytes,
                                         contrad_stream_t stream) {
  CONTRAD_ARG(dy && x && stats && gamma && beta && out2k && workspace && M > 0 && K > 0 && ld >= K);
  CONTRAD_ARG(workspace_bytes >= contrad_colstats_workspace_bytes(M, K, 1));
  int rpb, nb;
  colstats_plan(M, &rpb, &nb);
  dim3 grid(nb, cdiv(K, 64 * COLS_PER_THREAD));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_relu_bwd_stats_kernel, grid, dim3(256), 0, s, dy, x, M, K, ld, stats, count, gamma, beta,
                     eps, rpb, workspace);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(colstats_reduce_kernel, dim3(cdiv(2 * K, CSR_E)), dim3(256), 0, s, workspace, nb, 2, K, out2k, 0);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_bn_relu_bwd_apply(const float* dy, const float* x, float* dx, long long M, int K, int ld,
                                         const float* stats, float count, const float* gamma, const float* beta,
                                         float eps, const float* bstats, contrad_stream_t stream) {
  CONTRAD_ARG(dy && x && dx && stats && gamma && beta && bstats && M > 0 && K > 0 && ld >= K);
  long long grid = (M * K + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, dy, x, dx, M, K,
                     ld, stats, count, gamma, beta, eps, bstats);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_bn_running_update(const float* stats, float count, int K, const float* conv_bias,
                                         float momentum, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, contrad_stream_t stream) {
  CONTRAD_ARG(stats && running_mean && running_var && K > 0 && count >= 1.f);
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, stats,
                     count, K, conv_bias, momentum, running_mean, running_var, num_batches_tracked);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_gan_d_loss(const float* logits, int ld, int N, int kind, float* out3, float* grad,
                                  contrad_stream_t stream) {
  CONTRAD_ARG(logits && out3 && grad && N > 0 && ld >= 1 && kind >= 0 && kind <= 3);
  hipLaunchKernelGGL(gan_d_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, N, kind, out3, grad);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_gan_g_loss(const float* logits, int ld, int N, int kind, float* out1, float* grad,
                                  contrad_stream_t stream) {
  CONTRAD_ARG(logits && out1 && grad && N > 0 && ld >= 1 && kind >= 0 && kind <= 3);
  hipLaunchKernelGGL(gan_g_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, N, kind, out1, grad);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_adam_step(const contrad_adam_batch* b, int step, float lr, float beta1, float beta2,
                                 float eps, float grad_scale, contrad_stream_t stream) {
  CONTRAD_ARG(b && b->n > 0 && b->n <= CONTRAD_ADAM_MAX_TENSORS && step >= 1);
  contrad_adam_batch bb = *b;
  int blocks = 0;
  for (int i = 0; i < bb.n; ++i) {
    CONTRAD_ARG(bb.t[i].p && bb.t[i].g && bb.t[i].m && bb.t[i].v && bb.t[i].numel > 0);
    bb.block_start[i] = blocks;
    blocks += (int)((bb.t[i].numel + CONTRAD_ADAM_CHUNK - 1) / CONTRAD_ADAM_CHUNK);
  }
  bb.block_start[bb.n] = blocks;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bb, (float)(lr / bc1), beta1,
                     beta2, (float)(1.0 / sqrt(bc2)), eps, grad_scale, (const float*)nullptr);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_adam_step_dev(const contrad_adam_batch* b, const float* hyper_dev, float beta1, float beta2,
                                     float eps, contrad_stream_t stream) {
  CONTRAD_ARG(b && hyper_dev && b->n > 0 && b->n <= CONTRAD_ADAM_MAX_TENSORS);
  contrad_adam_batch bb = *b;
  int blocks = 0;
  for (int i = 0; i < bb.n; ++i) {
    CONTRAD_ARG(bb.t[i].p && bb.t[i].g && bb.t[i].m && bb.t[i].v && bb.t[i].numel > 0);
    bb.block_start[i] = blocks;
    blocks += (int)((bb.t[i].numel + CONTRAD_ADAM_CHUNK - 1) / CONTRAD_ADAM_CHUNK);
  }
  bb.block_start[bb.n] = blocks;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bb, 0.f, beta1, beta2, 1.f, eps, 1.f,
                     hyper_dev);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_axpby(float* y, const float* x, long long n, float a, float b, contrad_stream_t stream) {
  CONTRAD_ARG(y && x && n > 0);
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, y, x, n, a, b);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_pull_host(const float* host_pinned, float* dst, long long n, contrad_stream_t stream) {
  CONTRAD_ARG(host_pinned && dst && n > 0);
  long long grid = (n + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(pull_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, host_pinned, dst, n);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
