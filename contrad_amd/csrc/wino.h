// Winograd F(2x2, 3x3) in exact fp32 on v_mfma_f32_32x32x2_f32 for the 3x3 stride-1 pad-1 layers (included by igemm.hip).
//
// Replaces F.conv2d forward / input gradient of those layers (reference: models/gan/sndcgan.py:91-109,
// models/gan/stylegan2/layers.py:95-123, discriminator.py:60-76), which the reference reaches through cuDNN's own
// Winograd-class fp32 kernels.  Round 6: the direct implicit-GEMM kernels sit at 0.90 - 0.93 of the fp32 MFMA peak when
// run alone, so the only lever left on these layers is issuing fewer multiply-adds:
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      16 independent GEMMs  M_xi[tile][k] = sum_c V_xi[tile][c] U_xi[c][k]
//
// = 16 multiply-adds per 2x2 output tile, channel pair instead of 36 (2.25x fewer).  Nothing transformed ever reaches HBM
// except U = G g G^T (written once per call by wino_filter_kernel from the packed weight: 16/9 of the weight's size).
// The data gradient of such a layer is the same correlation over gy with the filter mirrored and its channel roles
// swapped: the same kernel with another U.
//
// Block (512 threads, ONE per CU, persistent) = 64 tiles x 64 output channels x all 16 xi.  Eight waves, two per SIMD
// (so that one wave's waits and staging instructions are covered by its partner's MFMAs): wave = (sub-block of 32 tiles x
// 32 couts, xi half) holds 8 accumulator tiles = 128 registers.  The contraction runs in 8-channel chunks through a
// double-buffered LDS stage (V: 16 xi x 64 tiles x 8 ch, U: 16 xi x 8 ch x 64 couts, quad layout [k/4][row][4] as in
// igemm_lean.h); per chunk a wave issues 32 MFMAs and its role's share of the staging of the chunks ahead:
//   * "mover" waves (xi rows 2, 3): the raw input patch of the block's tiles -- every pixel ONCE per block (the tiles of a
//     patch overlap: fetched per tile it is 3.2x the bytes, prototype v1) -- and U, global -> registers -> LDS;
//   * "transform" waves (xi rows 0, 1): raw patch (LDS) -> B^T d B -> V (LDS): the column stage in registers, the row
//     stage with one DPP exchange between the two lanes that share a tile (each owns two of the four columns).
// At the end of an item the two xi halves exchange their halves of A^T M A through LDS (8 of the 16 accumulator rows
// each way) and both store.  A block walks its list of (patch, cout block) items as one flat sequence of chunks: the
// loads of the next item's first chunks are in flight while the current item finishes.
//
// Prototype history and measurements: tools/micro/wino_proto*.hip, profiles/r06_wino_proto_*.txt (v1 one wave per SIMD
// 0.51 of peak issued, v2 + raw patch through LDS 0.59, v3 eight waves 0.65, v4 persistent 0.66 - 0.75; matrix pipe busy
// 0.81 at the 2.0 - 2.1 GHz the chip holds under this kernel).  Error against fp64: rel-L2 3 - 6e-7 (fp32 round-off class).
#pragma once

namespace wino {

constexpr int KQS = 264;               // dwords per k-quad plane: 64 rows x 4 + 8: the two k-quads of a chunk 8 banks apart
constexpr int PL = 528;                // dwords per xi plane (2 k-quads); = 16 (mod 32): the two lanes of a tile write planes an
                                       // odd number apart in one ds_write_b128 -> an 8-lane store group covers all 32 banks once
constexpr int V_SZ = 16 * PL;          // V (transformed input) then U (transformed filter)
constexpr int BUF = 2 * V_SZ;          // one stage: 67 584 B
constexpr int RAW_PX = 324;            // raw patch capacity: 18 x 18 pixels (x 8 channels) ...
constexpr int RAW_SZ = (RAW_PX + 1) * 8;   // ... + one pixel of zeros (padding outside a box that holds no halo)
constexpr int RAW0 = 2 * BUF;
constexpr int LDS_DWORDS = 2 * BUF + 2 * RAW_SZ;   // 155 968 B
constexpr unsigned OOB = 0x80000000u;

struct Args {
  const float* x;      // input activation [N][H][W][ldi]   (FWD: x;  DGRAD: gy)
  const float* U;      // [16][Cin/8][2][Cout][4]  (xi, chunk, k-quad, cout, 4 input channels); the xi = (a, 3) planes negated
  float* y;            // output [N][H][W][ldo]             (FWD: y;  DGRAD: dx)
  const float* bias;   // FWD: [Cout] or NULL
  const float* ref;    // FWD: addend;  DGRAD: the producer's activation (act');  y's layout;  or NULL
  float slope, gain;
  int N, H, W, Cin, Cout, ldi, ldo;
  int TH, TW;          // tiles per image part in a block (powers of two, TH * TW * NIMG = 64)
  int sh_tw, sh_thw;   // log2(TW), log2(TH * TW)
  int NIMG;            // images per block (small maps: 8x8 -> 4, 4x4 -> 16)
  int PH, PW;          // patches per image
  int NP, NKB;         // patches (image groups x PH x PW), 64-wide cout blocks
  int BH, BW;          // raw box per image part; r_org / c_org = its origin relative to the patch's first output pixel
  int r_org, c_org;    //   (-1: the box holds the halo, pixels outside the image load as zeros;  0: box = image, halo -> zero pixel)
};

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)0x80000000u : 0, 0x00020000);
}
__device__ __forceinline__ float dpp_swap1(float v) {   // value of the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

struct Item { int n_first, ph, pw, kb; };

template <int MODE, int ROLE>   // MODE: MODE_FWD / MODE_DGRAD (epilogue);  ROLE 0: transform waves (xi rows 0, 1), 1: movers (rows 2, 3)
__device__ __forceinline__ void body(const Args& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // (index inside the role's 4 waves)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int NKB = p.NKB;
  const int NCH = p.Cin >> 3;
  const int ppi = p.PH * p.PW;
  // work list of this block: items w = slot, slot + nslots, ... of its XCD's list (item -> kb = w % NKB, patch = (w / NKB) * 8 + xcd:
  // the cout blocks of a patch run at the same time on neighbouring CUs of one XCD and share its x in that L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int L = (p.NP > xcd) ? ((p.NP - xcd + 7) >> 3) * NKB : 0;
  int w_cur = slot;
  if (w_cur >= L) return;

  auto decode = [&](int w) -> Item {
    Item it;
    it.kb = w % NKB;
    const int patch = (w / NKB) * 8 + xcd;
    const int g = patch / ppi, pr = patch - g * ppi;
    it.n_first = g * p.NIMG;
    it.ph = pr / p.PW; it.pw = pr - it.ph * p.PW;
    return it;
  };

  // ---- movers: raw box items (pixel * 2 + kq): tid, tid + 256, tid + 512;  U: (cout, kq, xi group) 8 planes ----
  int rimg[3], rr[3], rc[3], rq[3];
  unsigned vu[8];
  const int npx = p.NIMG * p.BH * p.BW;
  const bool raw3 = tid + 512 < 2 * npx;
  const int ucout = tid & 63, ukq = (tid >> 6) & 1, uxg = tid >> 7;
  const unsigned u_step = (unsigned)(2 * p.Cout * 16);
  const int wrU = V_SZ + (uxg * 8) * PL + ukq * KQS + ucout * 4;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int item = tid + 256 * i, px = item >> 1;
      const int bhw = p.BH * p.BW;
      rq[i] = item & 1;
      rimg[i] = px / bhw;
      const int rem = px - rimg[i] * bhw;
      rr[i] = rem / p.BW; rc[i] = rem - rr[i] * p.BW;
      if (px >= npx) rr[i] = 1 << 20;     // never valid
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) vu[i] = (unsigned)((((uxg * 8 + i) * NCH * 2 + ukq) * p.Cout + ucout) * 16);
  }
  // state of the load streams: raw runs 3 chunks ahead, U 2 chunks ahead of the chunk being multiplied
  unsigned vraw[3];
  const float* xb_raw = nullptr;     // first image of the item the raw stream is in (nullptr: past the end)
  int t_raw = 0, w_raw = 0;          // its chunk / item
  unsigned u_koff = 0; int t_u = 0, w_u = 0; bool u_on = true;
  auto raw_item = [&](int w) {       // per-thread offsets of item w's box
    if (w < L) {
      const Item it = decode(w);
      xb_raw = p.x + (size_t)it.n_first * p.H * p.W * p.ldi;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int hh = it.ph * 2 * p.TH + p.r_org + rr[i], ww = it.pw * 2 * p.TW + p.c_org + rc[i];
        const bool ok = (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W && it.n_first + rimg[i] < p.N;
        vraw[i] = ok ? (unsigned)((((rimg[i] * p.H + hh) * p.W + ww) * p.ldi + rq[i] * 4) * 4) : OOB;
      }
    } else {
      xb_raw = nullptr;
    }
  };
  auto u_item = [&](int w) { u_on = w < L; u_koff = u_on ? (unsigned)((w % NKB) * 64 * 16) : 0u; };

  // ---- transform waves: (half, kq, tile): the lane pair of a tile splits its 4 columns; register A = the column the partner
  // needs (half 0: columns 1 | 0, half 1: columns 2 | 3) ----
  const int half = tid & 1, kq = (tid >> 1) & 1, tile = tid >> 2;
  const float sgn = half ? -1.f : 1.f;
  const int wrV0 = (half ? 3 : 0) * PL + kq * KQS + tile * 4;
  const int wrV1 = (half ? 2 : 1) * PL + kq * KQS + tile * 4;
  int rdRawA[4], rdRawB[4];
  if constexpr (ROLE == 0) {
    const int img = tile >> p.sh_thw, ty = (tile >> p.sh_tw) & (p.TH - 1), tx = tile & (p.TW - 1);
    const int cA = 2 * tx - 1 + (half ? 2 : 1) - p.c_org, cB = 2 * tx - 1 + (half ? 3 : 0) - p.c_org;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 2 * ty - 1 + i - p.r_org;
      const bool rok = (unsigned)r < (unsigned)p.BH;
      rdRawA[i] = (rok && (unsigned)cA < (unsigned)p.BW) ? ((img * p.BH + r) * p.BW + cA) * 8 + kq * 4 : RAW_PX * 8 + kq * 4;
      rdRawB[i] = (rok && (unsigned)cB < (unsigned)p.BW) ? ((img * p.BH + r) * p.BW + cB) * 8 + kq * 4 : RAW_PX * 8 + kq * 4;
    }
  }

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
      rxA[i] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rdRawA[i]);
      rxB[i] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rdRawB[i]);
    }
  };
  auto load_u = [&](int i) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(p.U, u_on);
    ru[i] = bload4(rs, vu[i], (unsigned)t_u * u_step + u_koff);
  };
  auto step_u = [&]() { if (++t_u == NCH) { t_u = 0; w_u += nslots; u_item(w_u); } };
  auto col_stage = [&](int c) {   // channel component c of the lane's two columns: t = B^T d
    const float a0 = ((const float*)&rxA[0])[c], a1 = ((const float*)&rxA[1])[c], a2 = ((const float*)&rxA[2])[c], a3 = ((const float*)&rxA[3])[c];
    const float b0 = ((const float*)&rxB[0])[c], b1 = ((const float*)&rxB[1])[c], b2 = ((const float*)&rxB[2])[c], b3 = ((const float*)&rxB[3])[c];
    tA[0][c] = a0 - a2; tA[1][c] = a1 + a2; tA[2][c] = a2 - a1; tA[3][c] = a1 - a3;
    tB[0][c] = b0 - b2; tB[1][c] = b1 + b2; tB[2][c] = b2 - b1; tB[3][c] = b1 - b3;
  };
  auto row_stage_store = [&](int bufoff, int i) {   // tile row i: (.) B across the lane pair, two planes out
    float4 o0, o1;
    float* q0 = (float*)&o0; float* q1 = (float*)&o1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float r = dpp_swap1(tA[i][c]);
      q0[c] = tB[i][c] - r;                       // half 0: V[i][0];  half 1: -V[i][3]  (U's (., 3) planes are negated)
      q1[c] = __builtin_fmaf(sgn, r, tA[i][c]);   // half 0: V[i][1];  half 1: V[i][2]
    }
    *reinterpret_cast<float4*>(smem + bufoff + wrV0 + i * 4 * PL) = o0;
    *reinterpret_cast<float4*>(smem + bufoff + wrV1 + i * 4 * PL) = o1;
  };
  auto store_u = [&](int bufoff, int i) { *reinterpret_cast<float4*>(smem + bufoff + wrU + i * PL) = ru[i]; };

  // ---- prologue (first item of the block) ----
  if constexpr (ROLE == 1) {
    if (tid < 4) {   // the zero pixel of both raw stages
      *reinterpret_cast<float4*>(smem + RAW0 + (tid >> 1) * RAW_SZ + RAW_PX * 8 + (tid & 1) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
    for (int i = 0; i < 3; ++i) store_raw(1, i);     // (raw stage 1 is read only after the second barrier)
    load_raw3();                                     // raw 2 -- BEFORE U 1, the order the chunk loop keeps them in flight in: the
#pragma unroll                                       // compiler's wait counts at the loop head are the merge of both ways in (with
    for (int i = 0; i < 8; ++i) load_u(i);           // U 1   raw youngest here it drained the queue at the top of every chunk pair)
    step_u();
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    read_raw(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) col_stage(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_stage_store(0, i);
  }
  __syncthreads();

  // iteration g (chunk t of the current item): MFMAs on stage g & 1;  transform waves: raw (g + 1) -> V stage (g+1)&1;
  //   movers: raw (g + 2) registers -> raw stage g & 1, raw (g + 3) into flight, U (g + 1) registers -> U stage (g+1)&1,
  //   U (g + 2) into flight
  // The chunk's barrier sits BEFORE the MFMAs of its last plane, and the first plane's fragments of the NEXT chunk are fetched
  // right behind it (every read of this stage has been issued by then: the last plane's fragments are in registers, and
  // every write of the next stage is done by slot 6) -- so the LDS latency after the barrier hides behind four MFMAs per
  // wave instead of leaving the matrix pipe empty (WINO_EARLY_BARRIER=0: the barrier at the end of the chunk, round-6 A/B).
#ifndef WINO_EARLY_BARRIER
#define WINO_EARLY_BARRIER 1
#endif
  float4 fa[2], fb[2];
  fa[0] = *reinterpret_cast<const float4*>(smem + rdA);
  fb[0] = *reinterpret_cast<const float4*>(smem + rdB);
  auto chunk = [&](auto par) {
    constexpr int P = decltype(par)::value;
    constexpr int cur = P * BUF, nxt = BUF - cur;
#if !WINO_EARLY_BARRIER
    fa[0] = *reinterpret_cast<const float4*>(smem + cur + rdA);
    fb[0] = *reinterpret_cast<const float4*>(smem + cur + rdB);
#endif
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
#if WINO_EARLY_BARRIER
      if (xi == 7) {
        __syncthreads();
        fa[0] = *reinterpret_cast<const float4*>(smem + nxt + rdA);
        fb[0] = *reinterpret_cast<const float4*>(smem + nxt + rdB);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      const float* a = (const float*)&fa[xi & 1];
      const float* b = (const float*)&fb[xi & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if !WINO_EARLY_BARRIER
    __syncthreads();
#endif
  };

  // exchange area (stage 1 is the free one at an item's end): [sub-block][16 rows][lane] float4 = 64 KB
  float4* xch = reinterpret_cast<float4*>(smem + BUF) + (wave * 16) * 64 + lane;
  const float g1 = p.gain, g0 = p.gain * p.slope;

  for (; w_cur < L; w_cur += nslots) {
    for (int t = 0; t < NCH; t += 2) {
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});     // (Cin % 16 == 0)
    }
    // ---- output transform.  Y[i][j] = sum_a AT[i][a] s_j[a],  s_0[a] = M[a][0] + M[a][1] + M[a][2],  s_1[a] = M[a][1] - M[a][2] - M[a][3]
    // (M[a][3] as accumulated: V's and U's (a, 3) planes are both negated).  A wave's part of (y00, y01, y10, y11):
    //   rows a = 0, 1 (ROLE 0): (s_0[0] + s_0[1], s_1[0] + s_1[1], s_0[1], s_1[1]);   a = 2, 3 (ROLE 1): (s_0[2], s_1[2], -s_0[2] - s_0[3], -s_1[2] - s_1[3])
    // The movers hand their part to the transform waves through LDS; those add, apply the epilogue and store.  (The movers
    // carry the next item's in-flight loads -- 44 registers -- across this point: with a share of the stores and of the
    // epilogue's second operand they spilled 65 - 138 registers into per-row scratch round trips, 4 - 7 us per item.)
    // (The last chunk ran on stage 1 and its barrier has passed: stage 1 is free; stage 0 holds the next item's chunk 0.)
    auto part = [&](int r) -> float4 {
      float s0[2], s1[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float m0 = acc[a * 4 + 0][r], m1 = acc[a * 4 + 1][r], m2 = acc[a * 4 + 2][r], m3 = acc[a * 4 + 3][r];
        s0[a] = m0 + m1 + m2;
        s1[a] = m1 - m2 - m3;
      }
      return ROLE == 0 ? make_float4(s0[0] + s0[1], s1[0] + s1[1], s0[1], s1[1])
                       : make_float4(s0[0], s1[0], -s0[0] - s0[1], -s1[0] - s1[1]);
    };
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64] = part(r);
      __syncthreads();
    } else {
      const Item it = decode(w_cur);
      const int cout = it.kb * 64 + wn * 32 + l31;
      float* ybase = p.y + (size_t)it.n_first * p.H * p.W * p.ldo;
      const __amdgpu_buffer_rsrc_t rsY = rsrc(ybase, true);
      const __amdgpu_buffer_rsrc_t rsR = rsrc(p.ref ? p.ref + (size_t)it.n_first * p.H * p.W * p.ldo : ybase, p.ref != nullptr);
      const unsigned dcol = (unsigned)p.ldo * 4u, drow = (unsigned)(p.W * p.ldo) * 4u;
      // Byte offset of accumulator row r's tile (its first output pixel, this lane's cout) in the item's images.  The tile
      // index wm * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2) is a sum of DISJOINT bits and (image, tile row, tile column) are bit
      // fields of it, so the offset is lane part + a wave-uniform part per r: one v_add (+ compare / select on small maps,
      // where a block holds several images and the last block is ragged).  (Written as img / ty / tx per row the compiler
      // hoisted 48 per-lane values out of the item loop into scratch and reloaded them behind vmcnt(0) -- i.e. behind the
      // epilogue loads just issued: 4 - 7 us per item.)
      auto bit_off = [&](int b) -> unsigned {        // offset contribution of tile-index bit b (uniform)
        return b < p.sh_tw ? (unsigned)((2 << b) * p.ldo * 4)
               : b < p.sh_thw ? (unsigned)((2 << (b - p.sh_tw)) * p.W * p.ldo * 4)
                              : (unsigned)((1 << (b - p.sh_thw)) * p.H * p.W * p.ldo * 4);
      };
      const unsigned lane_off = (lhi ? bit_off(2) : 0u) + (wm ? bit_off(5) : 0u) + (unsigned)((wn * 32 + l31) * 4);
      const int lane_img = (wm * 32 + 4 * lhi) >> p.sh_thw;
      const unsigned item_off = (unsigned)(((it.ph * 2 * p.TH * p.W + it.pw * 2 * p.TW) * p.ldo + it.kb * 64) * 4);
      const int img_lim = p.N - it.n_first - lane_img;      // images of this lane's sub-block that exist
      const unsigned base_off = lane_off + item_off;
      auto row_off = [&](int r) -> unsigned {
        const int rbits = (r & 3) + 8 * (r >> 2);
        const unsigned u = ((r & 1) ? bit_off(0) : 0u) + ((r & 2) ? bit_off(1) : 0u) + ((r & 4) ? bit_off(3) : 0u) + ((r & 8) ? bit_off(4) : 0u);
        return ((rbits >> p.sh_thw) < img_lim) ? base_off + u : OOB;
      };
      float rv[16][4];
#ifndef WINO_NO_REF
      if (p.ref) {   // uniform: the second operand of the epilogue goes into flight before the exchange barrier
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned o = row_off(r);
          rv[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)o, 0, 0));
          rv[r][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)(o + dcol), 0, 0));
          rv[r][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)(o + drow), 0, 0));
          rv[r][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)(o + drow + dcol), 0, 0));
        }
      }
#else
      for (int r = 0; r < 16; ++r) for (int e = 0; e < 4; ++e) rv[r][e] = 1.f;
#endif
      const float bj = (MODE == MODE_FWD && p.bias) ? p.bias[cout] : 0.f;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // (rows one by one: hoisted above the barrier, the 16 rows' own parts are 64 more live registers next to the 64 of rv)
        __builtin_amdgcn_sched_barrier(0);
        const float4 mine = part(r), oth = xch[r * 64];
        float v[4] = {mine.x + oth.x, mine.y + oth.y, mine.z + oth.z, mine.w + oth.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (MODE == MODE_DGRAD) {
            if (p.ref) v[e] *= (rv[r][e] > 0.f) ? g1 : g0;
          } else {
            v[e] += bj;
            v[e] *= (v[e] > 0.f) ? g1 : g0;
            if (p.ref) v[e] += rv[r][e];
          }
        }
        const unsigned o = row_off(r);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rsY, (int)o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[1]), rsY, (int)(o + dcol), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[2]), rsY, (int)(o + drow), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[3]), rsY, (int)(o + drow + dcol), 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();     // the exchange area is stage 1: nobody may write the next chunk into it before it has been read
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void wino_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) body<MODE, 0>(p, smem); else body<MODE, 1>(p, smem);
}

// U = G g G^T from the packed weight Wp[(kh * 3 + kw) * C + c][ldw] (cout contiguous), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
//   FWD:   g[k][c] = W[k][c][.][.]                 input channels c (C of them), output channels k:  U[xi][c/8][(c%8)/4][k][c%4]
//   DGRAD: g'[c][k][a][b] = W[k][c][2 - a][2 - b]  input channels k (gy's), output channels c:       U[xi][k/8][(k%8)/4][c][k%4]
// One thread = four input channels x one output channel (one float4 per xi plane; lanes along the output channel: stores
// fully coalesced; FWD loads coalesced along k, DGRAD loads 16-byte pieces of rows 4 * ldw bytes apart).
template <int MODE>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ wp, float* __restrict__ U, int C, int K, int ldw) {
  const int cin = (MODE == MODE_FWD) ? C : K, cout = (MODE == MODE_FWD) ? K : C;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (cin >> 2) * cout) return;
  const int o = idx % cout, q4 = idx / cout;       // output channel, input-channel quad
  float g[9][4];
  if constexpr (MODE == MODE_FWD) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[t][j] = wp[(size_t)(t * C + 4 * q4 + j) * ldw + o];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {   // g'[a][b] = w[2 - a][2 - b]: tap 8 - t
      const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)((8 - t) * C + o) * ldw + 4 * q4);
      g[t][0] = v.x; g[t][1] = v.y; g[t][2] = v.z; g[t][3] = v.w;
    }
  }
  const int nch = cin >> 3;
  float4* Uo = reinterpret_cast<float4*>(U);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float t[3][4];     // row a of G g: [column j of g][channel]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g0 = g[0 * 3 + j][e], g1 = g[1 * 3 + j][e], g2 = g[2 * 3 + j][e];
        t[j][e] = a == 0 ? g0 : a == 1 ? 0.5f * (g0 + g1 + g2) : a == 2 ? 0.5f * (g0 - g1 + g2) : g2;
      }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = b == 0 ? t[0][e] : b == 1 ? 0.5f * (t[0][e] + t[1][e] + t[2][e]) : b == 2 ? 0.5f * (t[0][e] - t[1][e] + t[2][e]) : -t[2][e];
        u[e] = v;     // (b == 3: negated, see body())
      }
      const int xi = a * 4 + b;
      Uo[((size_t)(xi * nch + (q4 >> 1)) * 2 + (q4 & 1)) * cout + o] = make_float4(u[0], u[1], u[2], u[3]);
    }
  }
}


// =====================================================================================================================
// Weight gradient of the same layers: F(3x3, 2x2) -- the 3x3 filter gradient of one 2x2 tile of gy against its 4x4 input
// patch costs 16 multiply-adds per channel pair instead of 36:
//
//   dW = A'^T [ sum_tiles (G'' gy_t G''^T) (.) (B^T d_t B) ] A'      16 GEMMs  S_xi[c][k] = sum_t V_xi[t][c] Gy_xi[t][k]
//
// with the SAME input transform B^T d B as the forward kernel (so V keeps the negated (a, 3) planes: A' below carries the
// sign), G'' = [[1,0],[1,1],[1,-1],[0,1]] and A'^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,-1]] (tools/micro: derived numerically,
// exact).  Reference: the weight gradient of F.conv2d (models/gan/sndcgan.py:91-109, stylegan2/layers.py:115-121).
//
// Block (512 threads, one per CU) = 64 input channels x 64 output channels x all 16 xi x ONE SPLIT of the tile axis; the
// contraction runs over chunks of 8 tiles (2 x 4 tiles of one image; 4x4 maps: 2 images).  Waves as in the forward kernel:
// (32 c x 32 k sub-block, xi half), 128 accumulator registers each; transform waves: raw x box (LDS) -> B^T d B -> V;
// movers: x box and gy global -> registers -> LDS, gy through G'' gy G''^T on the way (its tiles do not overlap), and the
// bias gradient (plane (1, 1) = the sum of the tile's four pixels) in the blocks of the first c-block.  LDS rows are
// CONTRACTION-major here ([xi][tile][64 channels]: the MFMA's k index is the tile), fragments are ds_read_b32 pairs 128
// dwords apart.  The block's result goes through A' (4x4 -> 3x3) into its slab of the workspace in the packed-weight
// layout; wgrad_reduce_kernel (igemm.hip) sums the slabs in fixed order, as for the direct kernels.
constexpr int W_PLANE = 512;                   // dwords per xi plane: 8 tiles x 64 channels
constexpr int W_VSZ = 16 * W_PLANE;            // V, then Gy
constexpr int W_STAGE = 2 * W_VSZ;             // 65 536 B
constexpr int W_RAWPX = 60;                    // raw box capacity: 6 x 10 pixels x 64 channels
constexpr int W_RAWSZ = W_RAWPX * 64;
constexpr int W_RAW0 = 2 * W_STAGE;
constexpr int W_ZERO = W_RAW0 + 2 * W_RAWSZ;   // one pixel of zeros (never written)
constexpr int W_LDS_DWORDS = W_ZERO + 64;      // 162 048 B

struct WArgs {
  const float* x;      // [N][H][W][ldx]
  const float* gy;     // [N][H][W][ldy]
  float* ws;           // [splits][9 * C][K] partial slabs (packed-weight layout, dense)
  float* bias_ws;      // [splits][K] partial bias gradients, or NULL
  int N, H, W, C, K, ldx, ldy;
  int CTH, CTW, sh_ctw, sh_cthw, CNIMG;   // tiles per image part in a chunk, images per chunk (CTH * CTW * CNIMG = 8)
  int QH, QW;          // chunks per image along h / w
  int Q, qps;          // chunks in all, chunks per split
  int BH, BW, r_org, c_org;   // raw box per image part (see Args)
  int CB, KB;          // 64-wide blocks of C and of K
};

template <int ROLE>   // 0: transform waves (xi rows 0, 1), 1: movers (rows 2, 3)
__device__ __forceinline__ void wbody(const WArgs& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;     // sub-block: 32 input channels x 32 output channels
  // block -> (c-block, k-block, split): the blocks of one split are neighbours on one XCD (they read the same x and gy)
  const int per = p.CB * p.KB;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / per, rem = lin - split * per;
  const int cb = rem / p.KB, kb = rem - cb * p.KB;
  const int q_begin = split * p.qps, q_end = min(p.Q, q_begin + p.qps);
  const int T = max(0, q_end - q_begin);
  const int qpi = p.QH * p.QW;

  // ---- chunk streams (movers): raw x runs 3 chunks ahead, gy 2 ahead ----
  struct Stream { int q, ng, cy, cx; };
  auto stream_at = [&](int q) -> Stream {
    Stream s; s.q = q; s.ng = q / qpi;
    const int r = q - s.ng * qpi;
    s.cy = r / p.QW; s.cx = r - s.cy * p.QW;
    return s;
  };
  auto step = [&](Stream& s) { ++s.q; if (++s.cx == p.QW) { s.cx = 0; if (++s.cy == p.QH) { s.cy = 0; ++s.ng; } } };

  // movers, raw x: items (pixel * 16 + channel quad): tid + 256 i
  int xfix[4], xr[4], xc[4], ximg[4], wrRaw[4];
  const int npx = p.CNIMG * p.BH * p.BW;
  // movers, gy: (channel quad, row half of G'' gy G''^T, tile)
  const int gq = tid & 15, hsel = (tid >> 4) & 1, gt = tid >> 5;
  int gfix = 0, gimg = 0;
  const int wrG = W_VSZ + (hsel * 8) * W_PLANE + gt * 64 + gq * 4;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (pieces past the box repeat pieces of its start -- the same load, the same store: no branch around the stores, whose
      // joins made the compiler wait for gy's loads, in flight behind the raw ones, at the top of every chunk)
      const int item = (tid + 256 * i) % (npx * 16), px = item >> 4, cq = item & 15;
      const int bhw = p.BH * p.BW;
      ximg[i] = px / bhw;
      const int r2 = px - ximg[i] * bhw;
      xr[i] = r2 / p.BW + p.r_org; xc[i] = r2 - (r2 / p.BW) * p.BW + p.c_org;      // relative to the chunk's first output pixel
      xfix[i] = ((ximg[i] * p.H + xr[i]) * p.W + xc[i]) * p.ldx * 4 + (cb * 64 + cq * 4) * 4;
      wrRaw[i] = px * 64 + ((cq ^ ((px & 1) * 8)) * 4);
    }
    gimg = gt >> p.sh_cthw;
    const int ty = (gt >> p.sh_ctw) & (p.CTH - 1), tx = gt & (p.CTW - 1);
    gfix = ((gimg * p.H + 2 * ty) * p.W + 2 * tx) * p.ldy * 4 + (kb * 64 + gq * 4) * 4;
  }
  Stream sx = stream_at(q_begin), sg = stream_at(q_begin);
  float4 rraw[4], rg[4];
  float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = p.bias_ws != nullptr && cb == 0;      // uniform per block

  auto load_raw = [&]() {     // chunk sx into flight, then step the stream
    const bool on = sx.q < q_end;
    const float* base = p.x + (size_t)sx.ng * p.CNIMG * p.H * p.W * p.ldx;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, on);
    const int oy = sx.cy * 2 * p.CTH, ox = sx.cx * 2 * p.CTW;
    const int soff = (oy * p.W + ox) * p.ldx * 4;
    const int nleft = p.N - sx.ng * p.CNIMG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (unsigned)(oy + xr[i]) < (unsigned)p.H && (unsigned)(ox + xc[i]) < (unsigned)p.W && ximg[i] < nleft;
      rraw[i] = bload4(rs, ok ? (unsigned)(xfix[i] + soff) : OOB, 0);
    }
    step(sx);
  };
  auto store_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(smem + W_RAW0 + stage * W_RAWSZ + wrRaw[i]) = rraw[i];
  };
  auto load_gy = [&]() {
    const bool on = sg.q < q_end && sg.ng * p.CNIMG + gimg < p.N;
    const float* base = p.gy + (size_t)sg.ng * p.CNIMG * p.H * p.W * p.ldy;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, sg.q < q_end);
    const unsigned v = on ? (unsigned)(gfix + ((sg.cy * 2 * p.CTH * p.W + sg.cx * 2 * p.CTW) * p.ldy) * 4) : OOB;
    const unsigned dc = (unsigned)p.ldy * 4u, dr = (unsigned)(p.W * p.ldy) * 4u;
    rg[0] = bload4(rs, v, 0); rg[1] = bload4(rs, v + dc, 0);
    rg[2] = bload4(rs, v + dr, 0); rg[3] = bload4(rs, v + dr + dc, 0);
    step(sg);
  };
  const float hs = hsel ? 1.f : 0.f;
  auto gy_stage_store = [&](int bufoff) {   // G'' gy G''^T, rows 2 hsel, 2 hsel + 1: 8 planes x float4
    float u0[2][4], u1[2][4];               // [column q][channel]: row a = 2 hsel | 2 hsel + 1 of G'' gy
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g0 = ((const float*)&rg[q])[e], g1 = ((const float*)&rg[2 + q])[e];
        u0[q][e] = __builtin_fmaf(-hs, g1, g0);          // hsel 0: g0        hsel 1: g0 - g1
        u1[q][e] = __builtin_fmaf(1.f - hs, g0, g1);     // hsel 0: g0 + g1   hsel 1: g1
      }
#pragma unroll
    for (int al = 0; al < 2; ++al) {
      const float (*u)[4] = al ? u1 : u0;
      float4 o0, o1, o2, o3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ((float*)&o0)[e] = u[0][e]; ((float*)&o1)[e] = u[0][e] + u[1][e];
        ((float*)&o2)[e] = u[0][e] - u[1][e]; ((float*)&o3)[e] = u[1][e];
      }
      float* dst = smem + bufoff + wrG + al * 4 * W_PLANE;
      *reinterpret_cast<float4*>(dst) = o0;
      *reinterpret_cast<float4*>(dst + W_PLANE) = o1;
      *reinterpret_cast<float4*>(dst + 2 * W_PLANE) = o2;
      *reinterpret_cast<float4*>(dst + 3 * W_PLANE) = o3;
      if (al == 1 && do_bias) {     // plane (1, 1) of the hsel = 0 lanes = g00 + g01 + g10 + g11
        asm volatile("" ::: "memory");
        colacc.x += o1.x; colacc.y += o1.y; colacc.z += o1.z; colacc.w += o1.w;
      }
    }
  };

  // ---- transform waves: (half, channel quad, tile) ----
  const int half = tid & 1, cq = (tid >> 1) & 15, tile = tid >> 5;
  const float sgn = half ? -1.f : 1.f;
  const int wsw = (cq ^ (half ? 4 : 0)) * 4;       // planes written by half 1 (b = 2, 3) keep their rows XOR 16: the lane pair of
  const int wrV0 = (half ? 3 : 0) * W_PLANE + tile * 64 + wsw;     // a tile then stores to different banks (32 of them for stores)
  const int wrV1 = (half ? 2 : 1) * W_PLANE + tile * 64 + wsw;
  int rdRawA[4], rdRawB[4];
  if constexpr (ROLE == 0) {
    const int img = tile >> p.sh_cthw, ty = (tile >> p.sh_ctw) & (p.CTH - 1), tx = tile & (p.CTW - 1);
    const int cA = 2 * tx - 1 + (half ? 2 : 1) - p.c_org, cB = 2 * tx - 1 + (half ? 3 : 0) - p.c_org;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 2 * ty - 1 + i - p.r_org;
      const bool rok = (unsigned)r < (unsigned)p.BH;
      const int pa = (img * p.BH + r) * p.BW + cA, pb = (img * p.BH + r) * p.BW + cB;
      rdRawA[i] = (rok && (unsigned)cA < (unsigned)p.BW) ? pa * 64 + ((cq ^ ((pa & 1) * 8)) * 4) : -1;
      rdRawB[i] = (rok && (unsigned)cB < (unsigned)p.BW) ? pb * 64 + ((cq ^ ((pb & 1) * 8)) * 4) : -1;
    }
  }
  float4 rxA[4], rxB[4];
  float tA[4][4], tB[4][4];
  auto read_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rxA[i] = *reinterpret_cast<const float4*>(smem + (rdRawA[i] >= 0 ? W_RAW0 + stage * W_RAWSZ + rdRawA[i] : W_ZERO + cq * 4));
      rxB[i] = *reinterpret_cast<const float4*>(smem + (rdRawB[i] >= 0 ? W_RAW0 + stage * W_RAWSZ + rdRawB[i] : W_ZERO + cq * 4));
    }
  };
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
    *reinterpret_cast<float4*>(smem + bufoff + wrV0 + i * 4 * W_PLANE) = o0;
    *reinterpret_cast<float4*>(smem + bufoff + wrV1 + i * 4 * W_PLANE) = o1;
  };

  // fragment reads: plane xi = (a, b), a = 2 ROLE + (0 | 1): A = V[t][c row], rows of the b >= 2 planes XOR 16; B = Gy[t][k]
  const int rdA0 = ROLE * 8 * W_PLANE + lhi * 64 + wm * 32 + l31;           // b = 0, 1
  const int rdA1 = ROLE * 8 * W_PLANE + lhi * 64 + wm * 32 + (l31 ^ 16);    // b = 2, 3
  const int rdB = W_VSZ + ROLE * 8 * W_PLANE + lhi * 64 + wn * 32 + l31;

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- prologue ----
  if constexpr (ROLE == 1) {
    if (tid < 16) *reinterpret_cast<float4*>(smem + W_ZERO + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    load_raw();                // raw 0
    load_gy();                 // gy 0
    store_raw(0);
    load_raw();                // raw 1
    gy_stage_store(0);
    store_raw(1);              // (raw stage 1 is read only after the second barrier)
    load_raw();                // raw 2 -- BEFORE gy 1, the order the chunk loop keeps them in flight in: the compiler's wait
    load_gy();                 // gy 1     counts at the loop head are the merge of both ways in (else: vmcnt(0) at every chunk pair)
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    read_raw(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) col_stage(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_stage_store(0, i);
  }
  __syncthreads();

  // iteration g: MFMAs on stage g & 1;  transform waves: raw (g + 1) -> V stage (g+1)&1;  movers: raw (g + 2) registers -> raw
  // stage g & 1, raw (g + 3) into flight, gy (g + 1) registers -> G'' gy G''^T -> Gy stage (g+1)&1, gy (g + 2) into flight
  auto chunk = [&](auto par) {
    constexpr int P = decltype(par)::value;
    constexpr int cur = P * W_STAGE, nxt = W_STAGE - cur;
    float fa[2][4], fb[2][4];
    auto frags = [&](int xi, int s) {
      const int ra = (xi & 2) ? rdA1 : rdA0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fa[s][j] = smem[cur + ra + xi * W_PLANE + j * 128];
        fb[s][j] = smem[cur + rdB + xi * W_PLANE + j * 128];
      }
    };
    frags(0, 0);
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      if (xi + 1 < 8) frags(xi + 1, (xi + 1) & 1);
      if constexpr (ROLE == 0) {
#ifndef WINO_WG_ABL_NO_TR
        if (xi == 0) read_raw(1 - P);
        if (xi == 1) { col_stage(0); col_stage(1); }
        if (xi == 2) { col_stage(2); col_stage(3); }
        if (xi >= 3 && xi < 7) row_stage_store(nxt, xi - 3);
#endif
      } else {
#ifndef WINO_WG_ABL_NO_RAW
        if (xi == 0) store_raw(P);
        if (xi == 1) load_raw();
#endif
#ifndef WINO_WG_ABL_NO_GYT
        if (xi == 3) gy_stage_store(nxt);
#endif
#ifndef WINO_WG_ABL_NO_GY
        if (xi == 5) load_gy();
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[xi & 1][j], fb[xi & 1][j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  for (int t = 0; t < T; t += 2) {
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});      // (an odd T runs one chunk of zeros: the streams are off past q_end)
  }

  // ---- output transform 4x4 -> 3x3 and the slab.  R[a][j] = sum_b S[a][b] ATc[j][b], ATc = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,+1]]
  // (column 3 sign: the negated V planes); dW[i][j] = sum_a ATr[i][a] R[a][j], ATr = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,-1]]:
  //   rows a = 0, 1 (ROLE 0): (R0 + .5 R1, .5 R1, .5 R1);   rows a = 2, 3 (ROLE 1): (.5 R2, -.5 R2, .5 R2 - R3)
  // The movers hand their nine values per element to the transform waves through LDS (free now): [wave][row][tap][lane].
  auto taps9 = [&](int r, float* o) {
    float R[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float s0 = acc[a * 4 + 0][r], s1 = acc[a * 4 + 1][r], s2 = acc[a * 4 + 2][r], s3 = acc[a * 4 + 3][r];
      const float hp = 0.5f * (s1 + s2);
      R[a][0] = s0 + hp; R[a][1] = 0.5f * (s1 - s2); R[a][2] = hp + s3;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (ROLE == 0) { o[0 * 3 + j] = R[0][j] + 0.5f * R[1][j]; o[1 * 3 + j] = 0.5f * R[1][j]; o[2 * 3 + j] = 0.5f * R[1][j]; }
      else { o[0 * 3 + j] = 0.5f * R[0][j]; o[1 * 3 + j] = -0.5f * R[0][j]; o[2 * 3 + j] = 0.5f * R[0][j] - R[1][j]; }
    }
  };
  float* xch = smem + (wave * 16 * 9) * 64 + lane;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float o[9];
      taps9(r, o);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) xch[(r * 9 + tp) * 64] = o[tp];
    }
    __syncthreads();
  } else {
    __syncthreads();
    float* slab = p.ws + (size_t)split * 9 * p.C * p.K;
    const int k = kb * 64 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      __builtin_amdgcn_sched_barrier(0);
      float o[9];
      taps9(r, o);
      const int c = cb * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) slab[((size_t)tp * p.C + c) * p.K + k] = o[tp] + xch[(r * 9 + tp) * 64];
    }
  }
  if (do_bias) {      // uniform: sum the eight tiles' partial column sums (hsel = 0 lanes of the movers)
    __syncthreads();
    float* red = smem;      // [8 tiles][64]
    if (ROLE == 1 && hsel == 0) *reinterpret_cast<float4*>(red + gt * 64 + gq * 4) = colacc;
    __syncthreads();
    if (ROLE == 0 && tid < 64) {
      float sum = 0.f;
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) sum += red[t8 * 64 + tid];
      p.bias_ws[(size_t)split * p.K + kb * 64 + tid] = sum;
    }
  }
}

__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) wbody<0>(p, smem); else wbody<1>(p, smem);
}

}  // namespace wino
