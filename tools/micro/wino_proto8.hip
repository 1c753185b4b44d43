// Prototype (round 6): Winograd F(2x2, 3x3) in exact fp32 on v_mfma_f32_32x32x2_f32 for the 3x3 stride-1
// pad-1 layers, input and output transforms in registers / LDS (nothing transformed ever reaches HBM).
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        16 independent GEMMs  M_xi[tile][k] = sum_c V_xi[tile][c] U_xi[c][k]
//
// v3: block = 64 tiles x 64 couts x 16 xi on EIGHT waves (two per SIMD, so that one wave's waits and staging instructions are
// covered by its partner's MFMAs): wave w = (sub-block w & 3, xi half w >> 2) holds 8 accumulator tiles = 128 registers; the
// waves of the low half (xi rows a = 0, 1) do the input transform, those of the high half (a = 2, 3) move x and U from
// global memory into LDS; at the end the high half hands its part of A^T M A to the low half through LDS.
// (v1 / v2, profiles/r06_wino_proto_v[12].txt: 4 waves, one per SIMD, 256 accumulator registers:  The contraction runs in 8-channel chunks; per chunk a wave
// issues 64 MFMAs (4096 matrix-pipe cycles) and, between them, its share of the NEXT chunk's staging: 8 buffer loads of x
// + 8 of U, 64 VALU of input transform (column stage in registers, row stage with one DPP exchange between the two
// lanes that share a tile), 16 ds_write_b128, and the 32 ds_read_b128 of its own fragments.
//
// standalone: hipcc --offload-arch=gfx950 -O3 -o wino_proto wino_proto.hip && ./wino_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef WINO_KQS
#define WINO_KQS 264      // dwords per k-quad plane: 64 rows x 4 + pad; = 8 (mod 32)
#endif
#ifndef WINO_PL
#define WINO_PL 528       // dwords per xi plane (2 k-quads); = 16 (mod 32): the two lanes of a tile write planes an odd
#endif                    // number apart in one ds_write_b128 -> their 8-lane group covers all 32 banks once
constexpr int KQS = WINO_KQS, PL = WINO_PL;
constexpr int V_SZ = 16 * PL;          // V (transformed input) then U (transformed filter)
constexpr int BUF = 2 * V_SZ;          // one stage: 67 584 B; two stages 135 168 B
constexpr int RAW_SZ = 18 * 18 * 8;   // raw input patch of one chunk: [pixel][8 channels]
constexpr int RAW0 = 2 * BUF;
constexpr int LDS_DWORDS = 2 * BUF + 2 * RAW_SZ;   // 155 904 B
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)0x80000000u : 0, 0x00020000);
}
__device__ __forceinline__ float dpp_swap1(float v) {   // value of the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

struct WinoArgs {
  const float* x;    // [N][H][W][C]
  const float* U;    // [16][C/8][2][K][4]   (xi, chunk, k-quad, cout, 4 channels); xi = (a, 3) planes negated
  float* y;          // [N][H][W][K]
  int N, H, W, C, K;
  int PH, PW;        // 16 x 16 output patches per image
};


// v4: PERSISTENT blocks (one per CU): a block walks its list of (patch, cout block) items as one flat sequence of chunks,
// so the loads of the next item's first chunks are in flight while the current item finishes, and the fixed cost per
// item (v3: 6 us of launch + cold prologue + epilogue per 34 us of chunks at C = 128) shrinks to the output transform.
template <int ROLE>   // 0: transform wave (xi rows 0, 1), 1: mover wave (xi rows 2, 3)
__device__ __forceinline__ void wino_body(const WinoArgs& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // (index inside the role's 4 waves)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int NKB = p.K / 64;
  const int NCH = p.C / 8;
  const int ppi = p.PH * p.PW;
  // work list of this block: items w = slot, slot + nslots, ... of its XCD's list (item -> kb = w % NKB, patch = (w / NKB) * 8 + xcd)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int L = p.N * ppi * NKB / 8;
  int w_cur = slot;
  if (w_cur >= L) return;

  struct Item { int n, ph, pw, kb; };
  auto decode = [&](int w) -> Item {
    Item it;
    it.kb = w % NKB;
    const int patch = (w / NKB) * 8 + xcd;
    it.n = patch / ppi;
    const int pr = patch - it.n * ppi;
    it.ph = pr / p.PW; it.pw = pr - it.ph * p.PW;
    return it;
  };

  // ---- mover: raw patch items (pixel * 2 + kq): tid, tid + 256, tid + 512 (< 648); U: (cout, kq, xi group) 8 planes ----
  int rr[3], rc[3], rq[3];
  unsigned vu[8];
  const bool raw3 = tid + 512 < 648;
  const int ucout = tid & 63, ukq = (tid >> 6) & 1, uxg = tid >> 7;
  const unsigned u_step = (unsigned)(2 * p.K * 16);
  const int wrU = V_SZ + (uxg * 8) * PL + ukq * KQS + ucout * 4;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int item = tid + 256 * i, px = item >> 1;
      rq[i] = item & 1; rr[i] = px / 18; rc[i] = px - rr[i] * 18;
      if (item >= 648) rr[i] = 1 << 20;     // never valid
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) vu[i] = (unsigned)(((((size_t)(uxg * 8 + i) * NCH) * 2 + ukq) * p.K + ucout) * 16);
  }
  // state of the load streams: raw runs 3 chunks ahead, U 2 chunks ahead of the chunk being multiplied
  unsigned vraw[3];
  const float* xb_raw = nullptr;     // image base of the item the raw stream is in (nullptr: past the end)
  int t_raw = 0, w_raw = 0;          // its chunk / item
  unsigned u_koff = 0; int t_u = 0, w_u = 0; bool u_on = true;
  auto raw_item = [&](int w) {       // per-thread offsets of item w's patch
    if (w < L) {
      const Item it = decode(w);
      xb_raw = p.x + (size_t)it.n * p.H * p.W * p.C;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int hh = it.ph * 16 - 1 + rr[i], ww = it.pw * 16 - 1 + rc[i];
        vraw[i] = ((unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W) ? (unsigned)(((hh * p.W + ww) * p.C + rq[i] * 4) * 4) : OOB;
      }
    } else {
      xb_raw = nullptr;
    }
  };
  auto u_item = [&](int w) { u_on = w < L; u_koff = u_on ? (unsigned)((w % NKB) * 64 * 16) : 0u; };

  // ---- transform: (half, kq, tile) ----
  const int half = tid & 1, kq = (tid >> 1) & 1, tile = tid >> 2;
  const int ty = tile >> 3, tx = tile & 7;
  const int rdRawA = ((2 * ty) * 18 + 2 * tx + (half ? 2 : 1)) * 8 + kq * 4;
  const int rdRawB = ((2 * ty) * 18 + 2 * tx + (half ? 3 : 0)) * 8 + kq * 4;
  const float sgn = half ? -1.f : 1.f;
  const int wrV0 = (half ? 3 : 0) * PL + kq * KQS + tile * 4;
  const int wrV1 = (half ? 2 : 1) * PL + kq * KQS + tile * 4;

  // fragment reads: this wave's 8 planes start at xi = ROLE * 8
  const int rdA = ROLE * 8 * PL + lhi * KQS + (wm * 32 + l31) * 4;
  const int rdB = V_SZ + ROLE * 8 * PL + lhi * KQS + (wn * 32 + l31) * 4;

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 rraw[3], ru[8];
  float4 rxA[4], rxB[4];
  float tA[4][4], tB[4][4];

  auto load_raw3 = [&]() {     // the raw stream's next chunk into flight, then step the stream
    const __amdgpu_buffer_rsrc_t rs = rsrc(xb_raw, xb_raw != nullptr);
#pragma unroll
    for (int i = 0; i < 3; ++i) rraw[i] = bload4(rs, vraw[i], (unsigned)t_raw * 32u);
    if (++t_raw == NCH) { t_raw = 0; w_raw += nslots; raw_item(w_raw); }
  };
  auto store_raw = [&](int stage, int i) {
    if (i < 2 || raw3) *reinterpret_cast<float4*>(smem + RAW0 + stage * RAW_SZ + (tid + 256 * i) * 4) = rraw[i];
  };
  auto read_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rxA[i] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rdRawA + i * 18 * 8);
      rxB[i] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rdRawB + i * 18 * 8);
    }
  };
  auto load_u = [&](int i) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(p.U, u_on);
    ru[i] = bload4(rs, vu[i], (unsigned)t_u * u_step + u_koff);
  };
  auto step_u = [&]() { if (++t_u == NCH) { t_u = 0; w_u += nslots; u_item(w_u); } };
  auto col_stage = [&](int c) {
    const float a0 = ((const float*)&rxA[0])[c], a1 = ((const float*)&rxA[1])[c], a2 = ((const float*)&rxA[2])[c], a3 = ((const float*)&rxA[3])[c];
    const float b0 = ((const float*)&rxB[0])[c], b1 = ((const float*)&rxB[1])[c], b2 = ((const float*)&rxB[2])[c], b3 = ((const float*)&rxB[3])[c];
    tA[0][c] = a0 - a2; tA[1][c] = a1 + a2; tA[2][c] = a2 - a1; tA[3][c] = a1 - a3;
    tB[0][c] = b0 - b2; tB[1][c] = b1 + b2; tB[2][c] = b2 - b1; tB[3][c] = b1 - b3;
  };
  auto row_stage_store = [&](int bufoff, int i) {
    float4 o0, o1;
    float* q0 = (float*)&o0; float* q1 = (float*)&o1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float r = dpp_swap1(tA[i][c]);
      q0[c] = tB[i][c] - r;
      q1[c] = __builtin_fmaf(sgn, r, tA[i][c]);
    }
    *reinterpret_cast<float4*>(smem + bufoff + wrV0 + i * 4 * PL) = o0;
    *reinterpret_cast<float4*>(smem + bufoff + wrV1 + i * 4 * PL) = o1;
  };
  auto store_u = [&](int bufoff, int i) { *reinterpret_cast<float4*>(smem + bufoff + wrU + i * PL) = ru[i]; };

  // ---- prologue (first item of the block) ----
  if constexpr (ROLE == 1) {
    w_raw = w_cur; raw_item(w_raw);
    w_u = w_cur; u_item(w_u);
    load_raw3();                                     // raw 0
#pragma unroll
    for (int i = 0; i < 8; ++i) load_u(i);           // U 0
    step_u();
#pragma unroll
    for (int i = 0; i < 3; ++i) store_raw(0, i);
    load_raw3();                                     // raw 1
#pragma unroll
    for (int i = 0; i < 8; ++i) store_u(0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_u(i);           // U 1
    step_u();
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    read_raw(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) col_stage(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_stage_store(0, i);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) store_raw(1, i);
    load_raw3();                                     // raw 2
  }
  __syncthreads();

  // iteration g (chunk t of the current item): MFMAs on stage g & 1;  transform waves: raw (g + 1) -> V stage (g+1)&1;
  //   movers: raw (g + 2) registers -> raw stage g & 1, raw (g + 3) into flight, U (g + 1) registers -> U stage (g+1)&1,
  //   U (g + 2) into flight
#ifdef WAVE_REUSE
  // TIMING EXPERIMENT (results are wrong: the epilogue still assumes the (sub-block, xi half) layout): wave = (xi quarter,
  // cout half) with BOTH tile groups -- the B fragment of a plane feeds two MFMA tiles: 12 fragment reads per chunk, not 16
  auto chunk = [&](auto par) {
    constexpr int P = decltype(par)::value;
    constexpr int cur = P * BUF, nxt = BUF - cur;
    const int w8 = (ROLE * 4 + wave), q = w8 >> 1, cn = w8 & 1;
    const int ra0 = q * 4 * PL + lhi * KQS + l31 * 4, ra1 = ra0 + 32 * 4;
    const int rb = V_SZ + q * 4 * PL + lhi * KQS + (cn * 32 + l31) * 4;
    float4 fa0[2], fa1[2], fb[2];
    fa0[0] = *reinterpret_cast<const float4*>(smem + cur + ra0);
    fa1[0] = *reinterpret_cast<const float4*>(smem + cur + ra1);
    fb[0] = *reinterpret_cast<const float4*>(smem + cur + rb);
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {      // slot xi: plane xi >> 1, tile group xi & 1
      if ((xi & 1) == 0 && xi + 2 < 8) {
        const int pl = (xi >> 1) + 1, s2 = pl & 1;
        fa0[s2] = *reinterpret_cast<const float4*>(smem + cur + ra0 + pl * PL);
        fa1[s2] = *reinterpret_cast<const float4*>(smem + cur + ra1 + pl * PL);
        fb[s2] = *reinterpret_cast<const float4*>(smem + cur + rb + pl * PL);
      }
      if constexpr (ROLE == 0) {
        if (xi == 0) read_raw(1 - P);
        if (xi == 1) { col_stage(0); col_stage(1); }
        if (xi == 2) { col_stage(2); col_stage(3); }
        if (xi >= 3 && xi < 7) row_stage_store(nxt, xi - 3);
      } else {
        if (xi == 0) { store_raw(P, 0); store_raw(P, 1); store_raw(P, 2); }
        if (xi == 1) load_raw3();
        if (xi >= 2 && xi < 6) { store_u(nxt, 2 * (xi - 2)); store_u(nxt, 2 * (xi - 2) + 1); }
        if (xi >= 3 && xi < 7) { load_u(2 * (xi - 3)); load_u(2 * (xi - 3) + 1); }
        if (xi == 6) step_u();
      }
      __builtin_amdgcn_sched_barrier(0);
      const int s1 = (xi >> 1) & 1;
      const float* a = (xi & 1) ? (const float*)&fa1[s1] : (const float*)&fa0[s1];
      const float* b = (const float*)&fb[s1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
#else
  auto chunk = [&](auto par) {
    constexpr int P = decltype(par)::value;
    constexpr int cur = P * BUF, nxt = BUF - cur;
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(smem + cur + rdA);
    fb[0] = *reinterpret_cast<const float4*>(smem + cur + rdB);
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      if (xi + 1 < 8) {
        fa[(xi + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdA + (xi + 1) * PL);
        fb[(xi + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdB + (xi + 1) * PL);
      }
      if constexpr (ROLE == 0) {
        if (xi == 0) read_raw(1 - P);
        if (xi == 1) { col_stage(0); col_stage(1); }
        if (xi == 2) { col_stage(2); col_stage(3); }
        if (xi >= 3 && xi < 7) row_stage_store(nxt, xi - 3);
      } else {
        if (xi == 0) { store_raw(P, 0); store_raw(P, 1); store_raw(P, 2); }
        if (xi == 1) load_raw3();
        if (xi >= 2 && xi < 6) { store_u(nxt, 2 * (xi - 2)); store_u(nxt, 2 * (xi - 2) + 1); }
        if (xi >= 3 && xi < 7) { load_u(2 * (xi - 3)); load_u(2 * (xi - 3) + 1); }
        if (xi == 6) step_u();
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* a = (const float*)&fa[xi & 1];
      const float* b = (const float*)&fb[xi & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
#endif

  float4* xch = reinterpret_cast<float4*>(smem + BUF) + (wave * 16) * 64 + lane;   // stage 1 is the free one at an item's end
  for (; w_cur < L; w_cur += nslots) {
    for (int t = 0; t < NCH; t += 2) {
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});     // (C % 16 == 0)
    }
    // ---- output transform: Y[i][j] = sum_a AT[i][a] s_j[a],  s_0[a] = M[a][0] + M[a][1] + M[a][2],  s_1[a] = M[a][1] - M[a][2] - M[a][3]
    // low half (a = 0, 1): P0j = s_j[0] + s_j[1], P1j = s_j[1];  high half (a = 2, 3): Q0j = s_j[2], Q1j = -s_j[2] - s_j[3]
    // (the last chunk ran on stage 1 and its barrier has passed: stage 1 is free; stage 0 holds the next item's chunk 0)
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s0[2], s1[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float m0 = acc[a * 4 + 0][r], m1 = acc[a * 4 + 1][r], m2 = acc[a * 4 + 2][r], m3 = acc[a * 4 + 3][r];
          s0[a] = m0 + m1 + m2;
          s1[a] = m1 - m2 - m3;
        }
        xch[r * 64] = make_float4(s0[0], s1[0], -s0[0] - s0[1], -s1[0] - s1[1]);
      }
    }
    __syncthreads();
    if constexpr (ROLE == 0) {
      const Item it = decode(w_cur);
      const int cout = it.kb * 64 + wn * 32 + l31;
      float* ybase = p.y + (size_t)it.n * p.H * p.W * p.K + cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int oy = it.ph * 16 + (tl >> 3) * 2, ox = it.pw * 16 + (tl & 7) * 2;
        float s0[2], s1[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float m0 = acc[a * 4 + 0][r], m1 = acc[a * 4 + 1][r], m2 = acc[a * 4 + 2][r], m3 = acc[a * 4 + 3][r];
          s0[a] = m0 + m1 + m2;
          s1[a] = m1 - m2 - m3;
        }
        const float4 q4 = xch[r * 64];
        float* q = ybase + ((size_t)oy * p.W + ox) * p.K;
        q[0] = s0[0] + s0[1] + q4.x; q[p.K] = s1[0] + s1[1] + q4.y;
        q[(size_t)p.W * p.K] = s0[1] + q4.z; q[(size_t)p.W * p.K + p.K] = s1[1] + q4.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();     // the exchange area is stage 1: nobody may write the next chunk into it before it has been read
  }
}

__global__ __launch_bounds__(512, 2) void wino_fwd(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) wino_body<0>(p, smem); else wino_body<1>(p, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
static void make_U(const std::vector<float>& w, int C, int K, std::vector<float>& U) {   // w: [K][C][3][3]
  const float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
  const int NCH = C / 8;
  U.assign((size_t)16 * C * K, 0.f);
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < C; ++c) {
      const float* g = &w[((size_t)k * C + c) * 9];
      float t[4][3], u[4][4];
      for (int a = 0; a < 4; ++a) for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) u[a][b] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
        const int xi = a * 4 + b, ch = c / 8, kq = (c % 8) / 4, j = c % 4;
        U[((((size_t)xi * NCH + ch) * 2 + kq) * K + k) * 4 + j] = (b == 3) ? -u[a][b] : u[a][b];
      }
    }
}

static int g_normal = 0;    // 1: standard-normal x (as tools/bench_conv.py) instead of uniform +-0.5: the chip clocks to its power budget
static double run_case(int N, int H, int C, int K, int reps) {
  const int W = H;
  const size_t nx = (size_t)N * H * W * C, ny = (size_t)N * H * W * K;
  std::vector<float> hx(nx), hw((size_t)K * C * 9), hU;
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (auto& v : hx) v = g_normal ? (rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd() + rnd()) * 1.0f * 3.4641f / 3.4641f * 1.0f : rnd();
  for (auto& v : hw) v = rnd() * 0.1f;
  if (g_normal == 2) { for (auto& v : hx) v = 0.f; for (auto& v : hw) v = 0.f; }
  make_U(hw, C, K, hU);
  float *dx, *dU, *dy;
  (void)hipMalloc(&dx, nx * 4); (void)hipMalloc(&dU, hU.size() * 4); (void)hipMalloc(&dy, ny * 4);
  (void)hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dU, hU.data(), hU.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(dy, 0xFF, ny * 4);
  WinoArgs a{dx, dU, dy, N, H, W, C, K, H / 16, W / 16};
  const int patches = N * a.PH * a.PW, items = patches * (K / 64);
  const int blocks = items < 256 ? items : 256;
  const size_t lds = (size_t)LDS_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)wino_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(wino_fwd, dim3(blocks), dim3(512), lds, 0, a);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return -1; }
  std::vector<float> hy(ny);
  (void)hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost);
  // sampled check against a double-precision direct correlation
  double maxerr = 0, maxref = 0, sse = 0, ssr = 0;
  for (int it = 0; it < 4096; ++it) {
    s = s * 1664525u + 1013904223u; const int nn = (s >> 8) % N;
    s = s * 1664525u + 1013904223u; int hh = (s >> 8) % H;
    s = s * 1664525u + 1013904223u; int ww = (s >> 8) % W;
    s = s * 1664525u + 1013904223u; const int kk = (s >> 8) % K;
    if (it < 64) { hh = (it & 1) ? H - 1 : 0; ww = (it & 2) ? W - 1 : 0; }   // corners
    double ref = 0;
    for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
      const int ih = hh - 1 + kh, iw = ww - 1 + kw;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      const float* xp = &hx[(((size_t)nn * H + ih) * W + iw) * C];
      for (int c = 0; c < C; ++c) ref += (double)xp[c] * hw[((size_t)kk * C + c) * 9 + kh * 3 + kw];
    }
    const double got = hy[(((size_t)nn * H + hh) * W + ww) * K + kk];
    maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
    sse += (got - ref) * (got - ref); ssr += ref * ref;
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_fwd, dim3(blocks), dim3(512), lds, 0, a);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_fwd, dim3(blocks), dim3(512), lds, 0, a);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double dense = 2.0 * N * H * W * (double)K * C * 9, issued = dense / 2.25;
  printf("N%-5d H%-4d C%-4d K%-4d  blocks %6d  %8.1f us  nominal %6.1f TF/s  issued %6.1f TF/s (%.3f of 157.3)  max|err| %.3g (max|ref| %.3g)  rel-L2 %.3g\n",
         N, H, C, K, blocks, ms * 1e3, dense / ms / 1e9, issued / ms / 1e9, issued / ms / 1e9 / 157.3, maxerr, maxref, sqrt(sse / ssr));
  (void)hipFree(dx); (void)hipFree(dU); (void)hipFree(dy);
  return ms;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  if (argc > 2 && argv[2][0] == 'n') g_normal = 1;
  else if (argc > 2 && argv[2][0] == 'z') g_normal = 2;      // all-zero x and weights: the chip's clock under the same instruction stream without data toggling
  else if (argc > 2) { run_case(48, 128, 128, 128, reps); return 0; }   // one shape (counter runs)
#ifdef ABL_ANY
  run_case(1536, 16, 128, 128, reps); run_case(48, 128, 128, 128, reps); return 0;
#endif
  run_case(8, 16, 128, 128, 2);            // small: correctness first
  run_case(1536, 16, 128, 128, reps);      // headline layer (today FWD 863 us in the step, ~800 alone)
  run_case(48, 128, 128, 128, reps);       // StyleGAN2_512 conv1 at 128^2 (1863 in the step, 1620 alone)
  run_case(1536, 16, 128, 128, reps);
  run_case(48, 128, 128, 128, reps);
  run_case(48, 64, 256, 256, reps);
  run_case(48, 32, 512, 512, reps);
  return 0;
}
