// Weight preparation for the conv engine, batched over all layers of a network in a handful of launches:
//
//   forward : spectral normalisation exactly as torch.nn.utils.spectral_norm's pre-forward hook (applied to
//             every Conv2d/Linear of D at models/gan/sndcgan.py:111-118): ONE power iteration per call in
//             training mode      v <- normalize(W^T u),  u <- normalize(W v),  sigma = u^T W v,
//             u/v updated in place, and the effective weight W/sigma written directly in the packed GEMM
//             layout Wp[(tap*C + c)][k] the implicit-GEMM kernels consume (OIHW -> "HWIO" transpose through
//             LDS).  Layers with fixed_scale > 0 skip the iteration and use W * fixed_scale instead
//             (StyleGAN2's EqualConv2d / EqualLinear runtime scale, stylegan2/layers.py:104,117,141).
//   backward: given G = dL/dWp (packed), dL/dW_orig = (G - <G, Wp> u v^T) / sigma   (u, v constants in the
//             graph, as in torch), transposed back to OIHW.
//
// Everything is deterministic: cross-block reductions go through per-block partial slots summed in a fixed
// order by the consumer.  W is [K][IN] row-major with IN = C*T, column index i = c*T + tap.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

constexpr int SN_THREADS = 256;
constexpr int MAXP = CONTRAD_SN_MAX_PARTIALS;  // partial slots per layer per reduction

struct BlockMap {
  int start[CONTRAD_SN_MAX_LAYERS + 1];
};

__device__ __forceinline__ int find_layer(const BlockMap& m, int n, int b) {
  int l = 0;
  while (l + 1 < n && b >= m.start[l + 1]) ++l;
  return l;
}

// scratch layout (floats) per layer l at base l*SCR_STRIDE:
//   [0, MAXP)        partial sums of |W^T u|^2        (phase 1)
//   [MAXP, 2 MAXP)   partial sums of |W v|^2          (phase 2)
//   [2 MAXP, 3 MAXP) partial sums of <G, Wp>          (backward)
//   [3 MAXP, +IN)    vt = W^T u (unnormalised)
//   then             t  = W v   (K floats)
__device__ __forceinline__ size_t scr_base(const contrad_sn_batch& b, int l) { return (size_t)b.scratch_off[l]; }

// ---- phase 1: vt[i] = sum_k W[k][i] u[k]; partial |vt|^2 per block ----
// Block = 1024 threads = 64 column quads (256 columns, one partial slot) x 16 row slices: every thread streams float4s
// of rows ks, ks+16, ... with 4 independent loads in flight, the 16 slices are summed through LDS in a fixed order.
// (One thread per column walking all K rows with dependent 4-byte loads took 190 us for 74 MB of weights.)
constexpr int P1_SLICES = 16;
__global__ __launch_bounds__(1024) void sn_phase1_kernel(contrad_sn_batch b, BlockMap map, float* __restrict__ scratch) {
  __shared__ float4 part[P1_SLICES][64];
  __shared__ float red[16];
  const int l = find_layer(map, b.n, blockIdx.x);
  const contrad_sn_layer& L = b.layers[l];
  const int chunk = blockIdx.x - map.start[l];
  const int IN = L.C * L.T;
  float* scr = scratch + scr_base(b, l);
  const int cq = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int i = chunk * SN_THREADS + cq * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((IN & 3) == 0 && ((uintptr_t)L.w & 15) == 0) {
    if (i < IN) {
      const float* w = L.w + i;
      int k = ks;
      for (; k + 3 * P1_SLICES < L.K; k += 4 * P1_SLICES) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)k * IN);
        const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)(k + P1_SLICES) * IN);
        const float4 w2 = *reinterpret_cast<const float4*>(w + (size_t)(k + 2 * P1_SLICES) * IN);
        const float4 w3 = *reinterpret_cast<const float4*>(w + (size_t)(k + 3 * P1_SLICES) * IN);
        const float u0 = L.u[k], u1 = L.u[k + P1_SLICES], u2 = L.u[k + 2 * P1_SLICES], u3 = L.u[k + 3 * P1_SLICES];
        acc.x += w0.x * u0; acc.y += w0.y * u0; acc.z += w0.z * u0; acc.w += w0.w * u0;
        acc.x += w1.x * u1; acc.y += w1.y * u1; acc.z += w1.z * u1; acc.w += w1.w * u1;
        acc.x += w2.x * u2; acc.y += w2.y * u2; acc.z += w2.z * u2; acc.w += w2.w * u2;
        acc.x += w3.x * u3; acc.y += w3.y * u3; acc.z += w3.z * u3; acc.w += w3.w * u3;
      }
      for (; k < L.K; k += P1_SLICES) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)k * IN);
        const float u0 = L.u[k];
        acc.x += w0.x * u0; acc.y += w0.y * u0; acc.z += w0.z * u0; acc.w += w0.w * u0;
      }
    }
  } else {
    float* a = reinterpret_cast<float*>(&acc);
    for (int k = ks; k < L.K; k += P1_SLICES) {
      const float uk = L.u[k];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < IN) a[j] += L.w[(size_t)k * IN + i + j] * uk;
    }
  }
  part[ks][cq] = acc;
  __syncthreads();
  float ss = 0.f;
  if (ks == 0) {
    float4 t = part[0][cq];
    for (int q = 1; q < P1_SLICES; ++q) { const float4 v = part[q][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const float* a = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j < IN) { scr[3 * MAXP + i + j] = a[j]; ss += a[j] * a[j]; }
  }
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) scr[chunk] = ss;
}

// Sum of n per-block partial slots, by the whole block (fixed tree -> every block, and every kernel that needs the same
// sum, gets the bit-identical value).  One thread walking up to 384 dependent loads cost more than the block's real work.
__device__ __forceinline__ float sum_partials(const float* p, int n, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  return block_sum(s, red);
}

// ---- phase 2: t[k] = sum_i W[k][i] vhat[i]; one wave per row; partial |t|^2 per block ----
__global__ void sn_phase2_kernel(contrad_sn_batch b, BlockMap map, int training, float eps,
                                 float* __restrict__ scratch) {
  __shared__ float red[16];
  const int l = find_layer(map, b.n, blockIdx.x);
  const contrad_sn_layer& L = b.layers[l];
  const int chunk = blockIdx.x - map.start[l];
  const int IN = L.C * L.T;
  float* scr = scratch + scr_base(b, l);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = chunk * 4 + wave;
  float inv = 1.f;
  const float* vsrc;
  if (training) {
    const int np1 = cdiv_dev(IN, SN_THREADS);
    inv = 1.f / fmaxf(sqrtf(sum_partials(scr, np1, red)), eps);
    vsrc = scr + 3 * MAXP;
  } else {
    vsrc = L.v;
  }
  float acc = 0.f;
  if (k < L.K) {
    const float* w = L.w + (size_t)k * IN;
    if ((IN & 3) == 0 && ((uintptr_t)L.w & 15) == 0 && ((uintptr_t)vsrc & 15) == 0) {
      float a0 = 0.f, a1 = 0.f;
      int i = lane * 4;
      for (; i + 256 < IN; i += 512) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + i), v0 = *reinterpret_cast<const float4*>(vsrc + i);
        const float4 w1 = *reinterpret_cast<const float4*>(w + i + 256), v1 = *reinterpret_cast<const float4*>(vsrc + i + 256);
        a0 += w0.x * v0.x + w0.y * v0.y + w0.z * v0.z + w0.w * v0.w;
        a1 += w1.x * v1.x + w1.y * v1.y + w1.z * v1.z + w1.w * v1.w;
      }
      for (; i < IN; i += 256) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + i), v0 = *reinterpret_cast<const float4*>(vsrc + i);
        a0 += w0.x * v0.x + w0.y * v0.y + w0.z * v0.z + w0.w * v0.w;
      }
      acc = a0 + a1;
    } else {
      for (int i = lane; i < IN; i += 64) acc += w[i] * vsrc[i];
    }
    acc = wave_sum(acc) * inv;
    if (lane == 0) scr[3 * MAXP + IN + k] = acc;
  }
  const float mine = (lane == 0 && k < L.K) ? acc * acc : 0.f;
  const float ss = block_sum(mine, red);
  if (threadIdx.x == 0) scr[MAXP + chunk] = ss;
}

// ---- phase 3: finalise u, v, sigma; write Wp = W / sigma in packed layout (LDS transpose) ----
// block tile: 32 rows (k) x CB channels (all T taps)
__global__ void sn_phase3_kernel(contrad_sn_batch b, BlockMap map, int training, float eps,
                                 float* __restrict__ scratch, float* __restrict__ sigma_out) {
  __shared__ float tile[32][257];
  __shared__ float red[16];
  const int l = find_layer(map, b.n, blockIdx.x);
  const contrad_sn_layer& L = b.layers[l];
  const int chunk = blockIdx.x - map.start[l];
  const int IN = L.C * L.T;
  float* scr = scratch + scr_base(b, l);
  const int CB = max(1, 256 / L.T);
  const int cchunks = cdiv_dev(L.C, CB);
  const int kt = chunk / cchunks, cc = chunk % cchunks;
  const int k0 = kt * 32, c0 = cc * CB;
  const int cn = min(CB, L.C - c0);
  const int width = cn * L.T;  // contiguous floats per row in this tile

  float inv_sigma;
  if (L.fixed_scale > 0.f) {
    inv_sigma = L.fixed_scale;
    if (chunk == 0 && threadIdx.x == 0) sigma_out[l] = 1.f / L.fixed_scale;
  } else {
    const float* t = scr + 3 * MAXP + IN;
    float sigma;
    if (training) {
      const int np2 = cdiv_dev(L.K, 4);
      const float nt2 = sum_partials(scr + MAXP, np2, red);
      const float inv_t = 1.f / fmaxf(sqrtf(nt2), eps);
      sigma = nt2 * inv_t;
      if (chunk == 0) {   // (uniform per block)
        const int np1 = cdiv_dev(IN, SN_THREADS);
        const float inv_v = 1.f / fmaxf(sqrtf(sum_partials(scr, np1, red)), eps);
        for (int k = threadIdx.x; k < L.K; k += blockDim.x) {
          const float uk = t[k] * inv_t;
          L.u[k] = uk;
          if (L.u_snap) L.u_snap[k] = uk;
        }
        for (int i = threadIdx.x; i < IN; i += blockDim.x) {
          const float vi = scr[3 * MAXP + i] * inv_v;
          L.v[i] = vi;
          if (L.v_snap) L.v_snap[i] = vi;
        }
      }
    } else {
      // sigma = u . (W v): the same fixed-tree dot in every block
      float part = 0.f;
      for (int k = threadIdx.x; k < L.K; k += blockDim.x) part += L.u[k] * t[k];
      sigma = block_sum(part, red);
      if (chunk == 0) {
        if (L.u_snap) for (int k = threadIdx.x; k < L.K; k += blockDim.x) L.u_snap[k] = L.u[k];
        if (L.v_snap) for (int i = threadIdx.x; i < IN; i += blockDim.x) L.v_snap[i] = L.v[i];
      }
    }
    inv_sigma = 1.f / sigma;
    if (chunk == 0 && threadIdx.x == 0) sigma_out[l] = sigma;
  }

  // load 32 x width (coalesced along the row), scaled.  (No per-element integer divisions in these loops: with 32
  // elements per thread they were ~1/3 of the kernel's time.)  16-byte accesses on both sides when the tile allows it
  // (every layer of the networks here): 8 float4 loads + 8 float4 stores per thread instead of 32 + 32 dword ones --
  // the pass ran at 2.0 - 2.5 TB/s on instruction count, not on bytes.
  const bool vec = ((width & 3) == 0) && ((IN & 3) == 0) && (((c0 * L.T) & 3) == 0) && ((L.ldw & 3) == 0) &&
                   ((((uintptr_t)L.w | (uintptr_t)L.wp) & 15) == 0);
  if (vec) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int w4 = width >> 2;
    for (int r = ty; r < 32; r += 4) {
      const int k = k0 + r;
      const float4* src = reinterpret_cast<const float4*>(L.w + (size_t)k * IN + (size_t)c0 * L.T);
      for (int j = tx; j < w4; j += 64) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < L.K) q = src[j];
        tile[r][4 * j + 0] = q.x * inv_sigma; tile[r][4 * j + 1] = q.y * inv_sigma;
        tile[r][4 * j + 2] = q.z * inv_sigma; tile[r][4 * j + 3] = q.w * inv_sigma;
      }
    }
    __syncthreads();
    // store: packed row (tap*C + c), 32 consecutive k as 8 float4; 32 rows per pass
    const int kq = threadIdx.x & 7;
    int rr = threadIdx.x >> 3;
    int tap = rr / cn, c = rr - tap * cn;
    for (; rr < width; rr += 32) {
      const int col = c * L.T + tap;
      const int k = k0 + 4 * kq;
      float* dst = L.wp + ((size_t)tap * L.C + c0 + c) * L.ldw + k;
      if (k + 3 < L.K) {
        *reinterpret_cast<float4*>(dst) = make_float4(tile[4 * kq][col], tile[4 * kq + 1][col], tile[4 * kq + 2][col],
                                                      tile[4 * kq + 3][col]);
      } else {
        for (int i = 0; i < 4; ++i)
          if (k + i < L.K) dst[i] = tile[4 * kq + i][col];
      }
      c += 32;
      while (c >= cn) { c -= cn; ++tap; }
    }
    return;
  }
  {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 32; r += 4) {
      const int k = k0 + r;
      const float* src = L.w + (size_t)k * IN + (size_t)c0 * L.T;
      for (int j = tx; j < width; j += 64) tile[r][j] = (k < L.K) ? src[j] * inv_sigma : 0.f;
    }
  }
  __syncthreads();
  // store: packed row (tap*C + c), 32 consecutive k; rr walks tap-major so consecutive rr are consecutive rows
  const int kk = threadIdx.x & 31;
  int rr = threadIdx.x >> 5;
  int tap = rr / cn, c = rr - tap * cn;
  for (; rr < width; rr += 8) {
    if (k0 + kk < L.K)
      L.wp[((size_t)tap * L.C + c0 + c) * L.ldw + k0 + kk] = tile[kk][c * L.T + tap];
    c += 8;
    while (c >= cn) { c -= cn; ++tap; }
  }
}

// ---- backward 1: partial <G, Wp> per block (grid-stride over packed rows) ----
__global__ void sn_bwd_dot_kernel(contrad_sn_batch b, BlockMap map, float* __restrict__ scratch) {
  __shared__ float red[16];
  const int l = find_layer(map, b.n, blockIdx.x);
  const contrad_sn_layer& L = b.layers[l];
  const int chunk = blockIdx.x - map.start[l];
  const int nchunks = map.start[l + 1] - map.start[l];
  float* scr = scratch + scr_base(b, l);
  float acc = 0.f;
  if (L.fixed_scale <= 0.f) {
    const long long total = (long long)L.C * L.T * L.K;
    if (L.ldw == L.K && (L.K & 3) == 0 && ((((uintptr_t)L.gwp | (uintptr_t)L.wp) & 15) == 0)) {
      // dense packed rows: two float4 streams, no index arithmetic (the per-element division + dword loads ran at 3 TB/s)
      const float4* g4 = reinterpret_cast<const float4*>(L.gwp);
      const float4* w4 = reinterpret_cast<const float4*>(L.wp);
      const long long n4 = total >> 2, stride = (long long)nchunks * blockDim.x;
      float a0 = 0.f, a1 = 0.f;
      long long q = (long long)chunk * blockDim.x + threadIdx.x;
      for (; q + stride < n4; q += 2 * stride) {          // two independent loads in flight per stream
        const float4 ga = g4[q], wa = w4[q], gb = g4[q + stride], wb = w4[q + stride];
        a0 += ga.x * wa.x + ga.y * wa.y + ga.z * wa.z + ga.w * wa.w;
        a1 += gb.x * wb.x + gb.y * wb.y + gb.z * wb.z + gb.w * wb.w;
      }
      if (q < n4) {
        const float4 ga = g4[q], wa = w4[q];
        a0 += ga.x * wa.x + ga.y * wa.y + ga.z * wa.z + ga.w * wa.w;
      }
      acc = a0 + a1;
    } else {
      for (long long e = (long long)chunk * blockDim.x + threadIdx.x; e < total;
           e += (long long)nchunks * blockDim.x) {
        const long long r = e / L.K;
        const int k = (int)(e - r * L.K);
        acc += L.gwp[r * L.ldw + k] * L.wp[r * L.ldw + k];
      }
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) scr[2 * MAXP + chunk] = s;
}

// ---- backward 2: gw[k][c*T+tap] = (G[(tap*C+c)][k] - dot * u[k] v[c*T+tap]) / sigma ----
__global__ void sn_bwd_write_kernel(contrad_sn_batch b, BlockMap map, BlockMap dotmap,
                                    const float* __restrict__ scratch, const float* __restrict__ sigma) {
  __shared__ float tile[32][257];
  __shared__ float red[16];
  const int l = find_layer(map, b.n, blockIdx.x);
  const contrad_sn_layer& L = b.layers[l];
  const int chunk = blockIdx.x - map.start[l];
  const int IN = L.C * L.T;
  const float* scr = scratch + scr_base(b, l);
  const int CB = max(1, 256 / L.T);
  const int cchunks = cdiv_dev(L.C, CB);
  const int kt = chunk / cchunks, cc = chunk % cchunks;
  const int k0 = kt * 32, c0 = cc * CB;
  const int cn = min(CB, L.C - c0);
  const int width = cn * L.T;
  float dot = 0.f, inv_sigma;
  if (L.fixed_scale > 0.f) {
    inv_sigma = L.fixed_scale;
  } else {
    dot = sum_partials(scr + 2 * MAXP, dotmap.start[l + 1] - dotmap.start[l], red);
    inv_sigma = 1.f / sigma[l];
  }
  const float* uu = L.u_snap ? L.u_snap : L.u;
  const float* vv = L.v_snap ? L.v_snap : L.v;
  const bool sn = L.fixed_scale <= 0.f;
  const bool vec = ((width & 3) == 0) && ((IN & 3) == 0) && (((c0 * L.T) & 3) == 0) && ((L.ldw & 3) == 0) &&
                   ((((uintptr_t)L.gw | (uintptr_t)L.gwp | (uintptr_t)vv) & 15) == 0);
  if (vec) {   // 16-byte accesses on both sides (see sn_phase3_kernel)
    const int kq = threadIdx.x & 7;
    {
      int rr = threadIdx.x >> 3;
      int tap = rr / cn, c = rr - tap * cn;
      for (; rr < width; rr += 32) {
        const int col = c * L.T + tap;
        const int k = k0 + 4 * kq;
        const float* src = L.gwp + ((size_t)tap * L.C + c0 + c) * L.ldw + k;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 3 < L.K) {
          q = *reinterpret_cast<const float4*>(src);
        } else {
          if (k < L.K) q.x = src[0];
          if (k + 1 < L.K) q.y = src[1];
          if (k + 2 < L.K) q.z = src[2];
        }
        tile[4 * kq][col] = q.x; tile[4 * kq + 1][col] = q.y; tile[4 * kq + 2][col] = q.z; tile[4 * kq + 3][col] = q.w;
        c += 32;
        while (c >= cn) { c -= cn; ++tap; }
      }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int w4 = width >> 2;
    for (int r = ty; r < 32; r += 4) {
      const int k = k0 + r;
      if (k >= L.K) break;
      const float du = sn ? dot * uu[k] : 0.f;
      float4* dst = reinterpret_cast<float4*>(L.gw + (size_t)k * IN + (size_t)c0 * L.T);
      const float4* v4 = reinterpret_cast<const float4*>(vv + (size_t)c0 * L.T);
      for (int j = tx; j < w4; j += 64) {
        float4 g = make_float4(tile[r][4 * j], tile[r][4 * j + 1], tile[r][4 * j + 2], tile[r][4 * j + 3]);
        if (sn) {
          const float4 v = v4[j];
          g.x -= du * v.x; g.y -= du * v.y; g.z -= du * v.z; g.w -= du * v.w;
        }
        g.x *= inv_sigma; g.y *= inv_sigma; g.z *= inv_sigma; g.w *= inv_sigma;
        dst[j] = g;
      }
    }
    return;
  }
  const int kk = threadIdx.x & 31;
  {
    int rr = threadIdx.x >> 5;
    int tap = rr / cn, c = rr - tap * cn;
    for (; rr < width; rr += 8) {
      tile[kk][c * L.T + tap] =
          (k0 + kk < L.K) ? L.gwp[((size_t)tap * L.C + c0 + c) * L.ldw + k0 + kk] : 0.f;
      c += 8;
      while (c >= cn) { c -= cn; ++tap; }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 32; r += 4) {
    const int k = k0 + r;
    if (k >= L.K) break;
    const float du = sn ? dot * uu[k] : 0.f;
    float* dst = L.gw + (size_t)k * IN + (size_t)c0 * L.T;
    for (int j = tx; j < width; j += 64) {
      float g = tile[r][j];
      if (sn) g -= du * vv[(size_t)c0 * L.T + j];
      dst[j] = g * inv_sigma;
    }
  }
}

int check_batch(const contrad_sn_batch* b) {
  CONTRAD_ARG(b && b->n > 0 && b->n <= CONTRAD_SN_MAX_LAYERS);
  for (int l = 0; l < b->n; ++l) {
    const contrad_sn_layer& L = b->layers[l];
    CONTRAD_ARG(L.w && L.wp && L.K > 0 && L.C > 0 && L.T > 0 && L.T <= 256 && L.ldw >= L.K);
    if (L.fixed_scale <= 0.f) {
      CONTRAD_ARG(L.u && L.v);
      CONTRAD_ARG(cdiv(L.C * L.T, SN_THREADS) <= MAXP && cdiv(L.K, 4) <= MAXP);
    }
  }
  return 0;
}

int tiles_of(const contrad_sn_layer& L) {
  const int CB = (256 / L.T) > 1 ? (256 / L.T) : 1;
  return cdiv(L.K, 32) * cdiv(L.C, CB);
}

}  // namespace

extern "C" long long contrad_sn_scratch_floats(int K, int C, int T) {
  return (3ll * MAXP + (long long)C * T + K + 16 + 3) & ~3ll;   // multiple of 4 floats: every layer's slab stays 16-B aligned
}

extern "C" int contrad_sn_weight_prep(const contrad_sn_batch* b, int training, float eps, float* scratch,
                                      float* sigma_out, contrad_stream_t stream) {
  int rc = check_batch(b);
  if (rc) return rc;
  CONTRAD_ARG(scratch && sigma_out);
  hipStream_t s = (hipStream_t)stream;
  BlockMap m1{}, m2{}, m3{};
  bool any_sn = false;
  for (int l = 0; l < b->n; ++l) {
    const contrad_sn_layer& L = b->layers[l];
    const bool sn = L.fixed_scale <= 0.f;
    any_sn |= sn;
    m1.start[l + 1] = m1.start[l] + ((sn && training) ? cdiv(L.C * L.T, SN_THREADS) : 0);
    m2.start[l + 1] = m2.start[l] + (sn ? cdiv(L.K, 4) : 0);
    m3.start[l + 1] = m3.start[l] + tiles_of(L);
  }
  if (any_sn) {
    if (training && m1.start[b->n] > 0) {
      hipLaunchKernelGGL(sn_phase1_kernel, dim3(m1.start[b->n]), dim3(64 * P1_SLICES), 0, s, *b, m1, scratch);
      CONTRAD_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(sn_phase2_kernel, dim3(m2.start[b->n]), dim3(256), 0, s, *b, m2, training, eps, scratch);
    CONTRAD_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sn_phase3_kernel, dim3(m3.start[b->n]), dim3(256), 0, s, *b, m3, training, eps, scratch,
                     sigma_out);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_sn_weight_grad(const contrad_sn_batch* b, float* scratch, const float* sigma,
                                      contrad_stream_t stream) {
  int rc = check_batch(b);
  if (rc) return rc;
  CONTRAD_ARG(scratch && sigma);
  hipStream_t s = (hipStream_t)stream;
  BlockMap md{}, mw{};
  for (int l = 0; l < b->n; ++l) {
    const contrad_sn_layer& L = b->layers[l];
    CONTRAD_ARG(L.gwp && L.gw);
    const long long total = (long long)L.C * L.T * L.K;
    int nb = (int)((total + 256 * 16 - 1) / (256 * 16));
    if (nb > MAXP) nb = MAXP;
    if (nb < 1) nb = 1;
    md.start[l + 1] = md.start[l] + (L.fixed_scale <= 0.f ? nb : 0);
    mw.start[l + 1] = mw.start[l] + tiles_of(L);
  }
  if (md.start[b->n] > 0) {
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(md.start[b->n]), dim3(256), 0, s, *b, md, scratch);
    CONTRAD_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sn_bwd_write_kernel, dim3(mw.start[b->n]), dim3(256), 0, s, *b, mw, md, scratch, sigma);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
