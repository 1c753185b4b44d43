// Winograd F(4x4, 3x3) in fp32 on v_mfma_f32_32x32x2_f32 for the 3x3 stride-1 pad-1 layers with maps >= 8x8 (included by
// igemm.hip after wino.h).
//
// Same layers as wino.h (reference: models/gan/sndcgan.py:91-109, models/gan/stylegan2/layers.py:95-123,
// discriminator.py:60-76; forward and input gradient of F.conv2d), one step further down the multiply-add count:
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      36 independent GEMMs  M_xi[tile][k] = sum_c V_xi[tile][c] U_xi[c][k]
//
// with 6x6 input tiles / 4x4 output tiles: 36 multiply-adds per 16 outputs and channel pair = 2.25 per output instead of
// 4 (F(2x2, 3x3)) or 9 (direct).  Interpolation points (0, +-1, +-2, inf) -- the standard matrices (Lavin & Gray 2016):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 round-off against fp64: rel-L2 2 - 5e-6 at 256 - 512 channels (direct fp32: 3 - 4e-7; the contract is 1e-3).
//
// Block (512 threads, ONE per CU, persistent) = 32 tiles (512 output pixels) x 64 output channels x all 36 xi.  Eight waves,
// two per SIMD: wave = (32-cout half, group of 9 xi) = 9 accumulator tiles = 144 registers.  The 36 planes of U no longer
// fit a double-buffered LDS stage beside V (2 x 110 KB), and no two waves of the block share a U fragment anyway (every
// wave owns its (xi group, cout half)): U is written by the filter kernel in the MFMA B-fragment layout and each wave loads
// its fragments global -> registers directly, one chunk ahead, refilled in place behind the MFMAs that consumed them.
// LDS holds V only (double-buffered, 2 x 36 KB) and the raw input box (2 x 26 KB):
//   * waves 4-7 ("movers"): the raw input box of the item (every pixel once per block and chunk), global -> registers -> LDS;
//   * waves 0-3 ("transform"): one (tile, channel) per thread: 36 raw values -> B^T d B (144 FMA-class operations) -> V.
// At the end of an item the waves exchange accumulators through LDS (both V stages: exactly 4 rows x 8 waves x 9 xi x 64 lanes
// x 4 B) in four passes of four accumulator rows; in pass q wave (half, group g) owns row 4q + g: reads all 36 M values of
// its element, runs A^T M A (100 operations), the epilogue, and stores 16 pixels x its cout.
// Measured (tools/micro/wino44_proto.hip, one MI355X, kernel alone, F(2x2, 3x3) in the step beside it): 1536 x 16^2 x 128 -> 128
// 336 us (492), 1536 x 8^2 x 256 305 (476), 48 x 128^2 x 128 630 (1110), 48 x 64^2 x 256 580 (1017), 48 x 32^2 x 512 540 (957),
// 48 x 256^2 x 64 752 (1297).  Where the time goes at 128 channels (ablations W44_NO_*: timing only, results wrong): MFMAs +
// fragment reads alone 220 us (the matrix pipe full at the clock the chip holds), + epilogue 48, + transform 24, + raw 16, + U 12,
// + 19 of them together.  Counters: matrix pipe busy 0.57 at 2.19 GHz (F(2x2, 3x3): 0.81 at 2.0 - 2.1).
#pragma once

namespace wino44 {

constexpr int NT = 32;                          // tiles per item
constexpr int KQS = NT * 4;                     // dwords per (plane, k-quad): 32 rows x 4 (the two k-quads of a transform thread's
                                                // ds_write_b32 pair share banks: 2-way, free on stores -- 16 dwords of padding: no change)
constexpr int PL = 2 * KQS;                     // per plane
constexpr int V_SZ = 36 * PL;                   // 9 216 dwords
constexpr int RPS = 10;                         // dwords per raw pixel (8 channels + 2): the four tiles of a 32-lane read group are
                                                // 4 pixels = 40 dwords = 8 banks apart
constexpr int RAW_SZ = 800 * RPS;               // 8 000 dwords: boxes of 18 x 34 = 612 (+ padding to 5 x 128 pieces), 2 x 18 x 18 = 648, 8 x 10 x 10 = 800 pixels
constexpr int RAW0 = 2 * V_SZ;
constexpr int LDS_DWORDS = 2 * V_SZ + 2 * RAW_SZ;     // 137 728 B
constexpr unsigned OOB = 0x80000000u;

struct Args {
  const float* x;      // input activation [N][H][W][ldi]   (FWD: x;  DGRAD: gy)
  const float* U;      // [36][Cin/8][2][Cout][4]  (xi, chunk, k-quad, cout, 4 input channels)
  float* y;            // output [N][H][W][ldo]             (FWD: y;  DGRAD: dx)
  const float* bias;   // FWD: [Cout] or NULL
  const float* ref;    // FWD: addend;  DGRAD: the producer's activation (act');  y's layout;  or NULL
  float slope, gain;
  int N, H, W, Cin, Cout, ldi, ldo;
  int TH, TW;          // 4x4 tiles per image part in a block (powers of two, TH * TW * NIMG = 32)
  int sh_tw, sh_thw;   // log2(TW), log2(TH * TW)
  int NIMG;            // images per block (16x16 maps: 2)
  int PH, PW;          // patches per image
  int NP, NKB;         // patches (image groups x PH x PW), 64-wide cout blocks
  int BH, BW;          // raw box per image part = (4 TH + 2) x (4 TW + 2): always with the halo (outside the image: zero fills)
  int n32;             // host side: the launch runs on wino44n_kernel (wino44n.h: 32-wide cout blocks, NKB = Cout / 32)
};

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)0x80000000u : 0, 0x00020000);
}

struct Item { int n_first, ph, pw, kb; };

// 1-D transforms (FMA-class operations: 12 / 10)
__device__ __forceinline__ void bt6(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {     // in place: B^T d
  const float t0 = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
  const float p = __builtin_fmaf(-4.f, d2, d4), q = __builtin_fmaf(-4.f, d1, d3);
  const float r = d4 - d2, s = d3 - d1;
  const float t5 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
  d0 = t0; d1 = p + q; d2 = p - q; d3 = __builtin_fmaf(2.f, s, r); d4 = __builtin_fmaf(-2.f, s, r); d5 = t5;
}
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float* o) {     // o[0..3] = A^T m
  const float p = m1 + m2, q = m1 - m2, u = m3 + m4, v = m3 - m4;
  o[0] = m0 + p + u;
  o[1] = __builtin_fmaf(2.f, v, q);
  o[2] = __builtin_fmaf(4.f, u, p);
  o[3] = __builtin_fmaf(8.f, v, q) + m5;
}

// MODE: MODE_FWD / MODE_DGRAD (epilogue);  ROLE 0: transform waves (0-3), 1: movers (4-7);  BOXW: raw box width, 10 (8x8 maps: 2 x 2
// tiles of eight images per block), 18 (16x16 maps: 4 x 4 tiles of two images) or 34 (wider maps: 4 x 8 tiles of one image) --
// compile-time, so that the transform threads' 36 window offsets are immediates and the movers' piece -> pixel split divides
// by constants
template <int MODE, int ROLE, int BOXW>
__device__ __forceinline__ void body(const Args& p, float* smem) {
  constexpr int TW = (BOXW == 34) ? 8 : (BOXW == 18) ? 4 : 2, TH = (BOXW == 10) ? 2 : 4, NIMG = 32 / (TH * TW), BH = 4 * TH + 2;
  constexpr int SH_TW = (BOXW == 34) ? 3 : (BOXW == 18) ? 2 : 1, SH_THW = SH_TW + ((BOXW == 10) ? 1 : 2);
  // BOXW 18 / 10: the whole image sits in the box (16x16 / 8x8 maps, 2 / 8 images per block): the movers fetch its interior
  // only (512 pixels) and the halo of both raw stages is zeroed once;  BOXW 34: 18 x 34 pixels of a larger image, all fetched
  // (outside the image: hardware zero fills)
  constexpr bool INTERIOR = BOXW != 34;
  constexpr int IW = BOXW - 2, IH = BH - 2;        // interior (= image) size when INTERIOR
  constexpr int NPX = INTERIOR ? NIMG * IH * IW : BH * BOXW;     // fetched pixels: 512 / 612
  constexpr int NRAW = (2 * NPX + 255) / 256;      // raw pieces (pixel, k-quad) per mover thread and chunk: 4 / 5
  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63;
  const int w8 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (wave-uniform: scalar registers)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wn = w8 & 1, grp = ROLE * 2 + ((w8 >> 1) & 1);     // cout half, xi group (planes 9 grp .. 9 grp + 8)
  const int NKB = p.NKB;
  const int NCH = p.Cin >> 3;                      // (>= 4, even)
  const int ppi = p.PH * p.PW;
  // work list: items w = slot, slot + nslots, ... of this XCD's list (item -> kb = w % NKB, patch = (w / NKB) * 8 + xcd)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int L = (p.NP > xcd) ? ((p.NP - xcd + 7) >> 3) * NKB : 0;
  int w_cur = slot;
  if (w_cur >= L) return;

  auto decode = [&](int w) -> Item {
    Item it;
    it.kb = w % NKB;
    const int patch = (w / NKB) * 8 + xcd;
    const int g = patch / ppi, pr = patch - g * ppi;
    it.n_first = g * NIMG;
    it.ph = pr / p.PW; it.pw = pr - it.ph * p.PW;
    return it;
  };

  // ---- movers: raw box pieces (pixel * 2 + k-quad): tid + 256 i.  Two offset sets: the item being multiplied and the next one
  // (the raw stream runs three chunks ahead and crosses into the next item in the last three chunks: a select per load, no
  // branch in the chunk loop -- with control flow there the compiler drains the whole vector-memory queue at every join) ----
  unsigned vraw_cur[NRAW], vraw_nxt[NRAW];
  const float *xb_cur = nullptr, *xb_nxt = nullptr;       // first image of the item (nullptr: no such item)
  auto raw_offsets = [&](int w, unsigned* v, const float*& xb) {
    if (w < L) {
      const Item it = decode(w);
      xb = p.x + (size_t)it.n_first * p.H * p.W * p.ldi;
      const int nleft = p.N - it.n_first;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        const int piece = tid + 256 * i, px = piece >> 1;
        int im, hh, ww;
        if constexpr (INTERIOR) { im = px / (IH * IW); hh = (px / IW) % IH; ww = px % IW; }
        else { im = 0; hh = it.ph * 4 * TH - 1 + px / BOXW; ww = it.pw * 4 * TW - 1 + px % BOXW; }
        const bool ok = px < NPX && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W && im < nleft;
        v[i] = ok ? (unsigned)((((im * p.H + hh) * p.W + ww) * p.ldi + (piece & 1) * 4) * 4) : OOB;
      }
    } else {
      xb = nullptr;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) v[i] = OOB;
    }
  };
  float4 rraw[NRAW];
  auto load_raw = [&](int k) {       // chunk k of the current item (k >= NCH: chunk k - NCH of the next one) into flight
    const bool nx = k >= NCH;
    const float* xb = nx ? xb_nxt : xb_cur;
    const __amdgpu_buffer_rsrc_t rs = rsrc(xb, xb != nullptr);
    const unsigned soff = (unsigned)(nx ? k - NCH : k) * 32u;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) rraw[i] = bload4(rs, nx ? vraw_nxt[i] : vraw_cur[i], soff);
  };
  int wraw[NRAW];                    // LDS dword offset of the thread's pieces inside a raw stage
#pragma unroll
  for (int i = 0; i < NRAW; ++i) {
    const int piece = tid + 256 * i, px = piece >> 1;
    const int bpx = INTERIOR ? ((px / (IH * IW)) * BH + (px / IW) % IH + 1) * BOXW + px % IW + 1 : px;
    wraw[i] = bpx * RPS + (piece & 1) * 4;         // (BOXW 34: pieces past the box land in the stage's padding: no branch)
  }
  auto store_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      float* dst = smem + RAW0 + stage * RAW_SZ + wraw[i];
      *reinterpret_cast<float2*>(dst) = make_float2(rraw[i].x, rraw[i].y);
      *reinterpret_cast<float2*>(dst + 2) = make_float2(rraw[i].z, rraw[i].w);
    }
  };

  // ---- transform waves: one (tile, channel) per thread ----
  const int tch = tid & 7, ttile = tid >> 3;
  int rd0 = 0;
  if constexpr (ROLE == 0) {
    const int img = ttile >> SH_THW, ty = (ttile >> SH_TW) & (TH - 1), tx = ttile & (TW - 1);
    rd0 = RAW0 + ((img * BH + 4 * ty) * BOXW + 4 * tx) * RPS + tch;
  }
  constexpr int rowstep = BOXW * RPS;
  const int wrV = (tch >> 2) * KQS + ttile * 4 + (tch & 3);
  float d[6][6];
  auto tr_read = [&](int rstage) {                    // the thread's 6x6 raw window
    const float* src = smem + rd0 + rstage * RAW_SZ;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) d[i][j] = src[i * rowstep + j * RPS];
  };
  auto tr_cols = [&](int j0) {                        // B^T d, columns j0 .. j0 + 2
#pragma unroll
    for (int j = j0; j < j0 + 3; ++j) bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
  };
  auto tr_row = [&](int vstage, int i) {              // (.) B for row i, six planes out
    bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    float* dst = smem + vstage * V_SZ + wrV + i * 6 * PL;
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j * PL] = d[i][j];
  };

  // ---- every wave: its U fragments, global -> registers (xi plane 9 grp + i; lane: k-quad lhi, cout wn * 32 + l31): a ring of
  // six float4, refilled in place six MFMA slots ahead ----
  const unsigned u_voff = (unsigned)((lhi * p.Cout + wn * 32 + l31) * 16);
  const unsigned u_plane = (unsigned)(NCH * 2 * p.Cout * 16), u_step = (unsigned)(2 * p.Cout * 16);
  const unsigned u_grp = (unsigned)(9 * grp) * u_plane;
  auto u_base = [&](int w) -> unsigned { return u_grp + (unsigned)((w % NKB) * 64 * 16); };      // chunk 0 of item w
  float4 ru[6];
  auto load_u = [&](int ring, unsigned soff, bool on) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(p.U, on);
    ru[ring] = bload4(rs, u_voff, soff);
  };

  // fragment reads of V: plane 9 grp + xi, k-quad lhi, tile l31
  const int rdA = 9 * grp * PL + lhi * KQS + l31 * 4;

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if constexpr (INTERIOR) {          // the halo of both raw stages: zeros, never written again
    for (int i = threadIdx.x; i < 2 * RAW_SZ / 4; i += 512) reinterpret_cast<float4*>(smem + RAW0)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  }
  // ---- prologue (first item): raw 0 / 1 in their stages, raw 2 in flight;  the first six U fragments in flight ----
  if constexpr (ROLE == 1) {
    raw_offsets(w_cur, vraw_cur, xb_cur);
    raw_offsets(w_cur + nslots, vraw_nxt, xb_nxt);
    load_raw(0); store_raw(0);
    load_raw(1); store_raw(1);
    load_raw(2);
  }
  // (the raw loads BEFORE the U fragments, here and after the epilogue, as in the chunk loop: the compiler's wait counts at
  // the loop head are the merge of both ways in, and with the raw pieces youngest on one of them it drains the queue there)
  unsigned u_pair = u_base(w_cur);      // soffset of (plane 9 grp, first chunk of the current pair)
#pragma unroll
  for (int i = 0; i < 6; ++i) load_u(i, u_pair + (unsigned)i * u_plane, true);

  const float g1 = p.gain, g0 = p.gain * p.slope;

  // A pair of chunks (t on V stage 0, t + 1 on stage 1) = 18 slots of four MFMAs, one basic block.  Chunk c: transform waves raw
  // (c + 1) -> V stage (c + 1) & 1 (not in the item's last chunk: both V stages are the exchange area next);  movers: raw (c + 2)
  // registers -> raw stage c & 1, raw (c + 3) into flight (the item's last one: after the epilogue, like the U fragments of the
  // next item -- in flight across the epilogue they cost it 50 - 60 registers).  LAST: the item's last pair (its own copy of the
  // code: no branch inside a pair).
  auto pair = [&](auto last_c, int t) {
    constexpr bool LAST = decltype(last_c)::value;
    float4 fa[2];
    fa[0] = *reinterpret_cast<const float4*>(smem + rdA);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      const int P = s / 9, xi = s - 9 * P;
      if (xi + 1 < 9) fa[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + P * V_SZ + rdA + (xi + 1) * PL);
      if constexpr (ROLE == 1) {
#ifndef W44_NO_RAW
        if (xi == 0) store_raw(P);
        if (xi == 1 && !(P == 1 && LAST)) load_raw(t + P + 3);
#endif
      } else if (!(P == 1 && LAST)) {
#ifndef W44_NO_TRANSFORM
        if (xi == 0) tr_read(1 - P);
        if (xi == 1) tr_cols(0);
        if (xi == 2) tr_cols(3);
        if (xi >= 3) tr_row(1 - P, xi - 3);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* a = (const float*)&fa[s & 1];
      const float* b = (const float*)&ru[s % 6];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the fragment of slot s + 6 into the ring entry just consumed
#ifndef W44_NO_ULOAD
      if (s + 6 < 18) load_u(s % 6, u_pair + (unsigned)((s + 6) / 9) * u_step + (unsigned)((s + 6) % 9) * u_plane, true);
      else if (!LAST) load_u(s % 6, u_pair + 2u * u_step + (unsigned)(s - 12) * u_plane, true);
#endif
      if (xi == 8) {
        __syncthreads();
        if (P == 0) fa[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + V_SZ + rdA);
      }
    }
    u_pair += 2u * u_step;
  };

  for (; w_cur < L; w_cur += nslots) {
    __syncthreads();                 // raw 0 (and 1) of this item are in LDS; the exchange area is free again
    if constexpr (ROLE == 0) {
      tr_read(0); tr_cols(0); tr_cols(3);
#pragma unroll
      for (int i = 0; i < 6; ++i) tr_row(0, i);
    }
    __syncthreads();

    for (int t = 0; t + 2 < NCH; t += 2) pair(std::false_type{}, t);
    pair(std::true_type{}, NCH - 2);

    // ---- output transform through the exchange area: [wave][xi][lane] float4 = accumulator rows 4q .. 4q + 3 ----
    // In pass q this wave owns accumulator row r = 4 q + grp = tile (r & 3) + 8 (r >> 2) + 4 lhi = grp + 8 q + 4 lhi of its cout half.
    // (image, tile row, tile column) are bit fields of the tile index and only bit 2 (lhi) is per lane: byte offset = lane part
    // (one register for the whole kernel) + a wave-uniform part that travels in the buffer instructions' scalar offset.
    // (Written as img / ty / tx arithmetic per pass the compiler hoisted per-lane values out of the item loop into scratch,
    // and every reload waits for the stores in flight.)
    const Item it = decode(w_cur);
    auto bit_off = [&](int bb) -> unsigned {      // offset contribution of tile-index bit bb (uniform)
      return bb < SH_TW ? (unsigned)((4 << bb) * p.ldo * 4)
             : bb < SH_THW ? (unsigned)((4 << (bb - SH_TW)) * p.W * p.ldo * 4)
                           : (unsigned)((1 << (bb - SH_THW)) * p.H * p.W * p.ldo * 4);
    };
    const unsigned lane_off = (lhi ? bit_off(2) : 0u) + (unsigned)((wn * 32 + l31) * 4);
    const unsigned item_off = (unsigned)((((it.ph * TH * 4) * p.W + it.pw * TW * 4) * p.ldo + it.kb * 64) * 4)
                              + ((grp & 1) ? bit_off(0) : 0u) + ((grp & 2) ? bit_off(1) : 0u);
    float* ybase = p.y + (size_t)it.n_first * p.H * p.W * p.ldo;
    const float* rbase = p.ref ? p.ref + (size_t)it.n_first * p.H * p.W * p.ldo : ybase;
    const unsigned dcol = (unsigned)p.ldo * 4u, drow = (unsigned)(p.W * p.ldo) * 4u;
    const float bj = (MODE == MODE_FWD && p.bias) ? p.bias[it.kb * 64 + wn * 32 + l31] : 0.f;
    // (with no second operand the loads below are off and return zeros: FWD adds them; DGRAD's two gains are then both 1)
    const float ga = p.ref ? g1 : 1.f, gb = p.ref ? g0 : 1.f;
    // exchange area: [accumulator row r of the pass (4)][wave (8)][xi (9)][lane (64)] dwords: written as 36 ds_write_b32, read
    // back conflict-free (lanes consecutive).  (As [wave][xi][lane] float4 the readers' dwords were 4 apart: 4-way bank conflicts
    // on every one of the 144 reads of an item, 40 % of the kernel's LDS cycles.)
    float* xw = smem + (w8 * 9) * 64 + lane;
    // reader: the wave of (cout half wn, group g') is w8' = (g' >> 1) * 4 + (g' & 1) * 2 + wn
    const float* xr = smem + (grp * 72 + wn * 9) * 64 + lane;
#ifdef W44_NO_EPILOGUE
    if (p.N < 0)
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (the element's image exists?  per lane only where a half-wave step of four tiles crosses images: 8x8 maps)
      const unsigned voff = (it.n_first + ((grp + 8 * q + 4 * lhi) >> SH_THW) < p.N) ? lane_off : OOB;
      const __amdgpu_buffer_rsrc_t rsY = rsrc(ybase, true);
      const __amdgpu_buffer_rsrc_t rsR = rsrc(rbase, p.ref != nullptr);
      const unsigned s0 = item_off + ((q & 1) ? bit_off(3) : 0u) + ((q & 2) ? bit_off(4) : 0u);
      float rv[4][4];                // the epilogue's second operand goes into flight before the exchange
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rv[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)voff, (int)(s0 + (unsigned)i * drow + (unsigned)j * dcol), 0));
#pragma unroll
      for (int xi = 0; xi < 9; ++xi)
#pragma unroll
        for (int r = 0; r < 4; ++r) xw[(r * 72 + xi) * 64] = acc[xi][4 * q + r];
      __syncthreads();
      float S[6][4];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        if ((a & 1) == 0) __builtin_amdgcn_sched_barrier(0);      // (two rows of M in flight at a time; 3 or 6 rows: the same time)
        float m[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          const int xi36 = a * 6 + b, gq = xi36 / 9, xq = xi36 - gq * 9;
          m[b] = xr[(((gq >> 1) * 4 + (gq & 1) * 2) * 9 + xq) * 64];
        }
        at6(m[0], m[1], m[2], m[3], m[4], m[5], S[a]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
        at6(S[0][j], S[1][j], S[2][j], S[3][j], S[4][j], S[5][j], v);      // column j of the tile: pixels (0..3, j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (MODE == MODE_DGRAD) {
            v[i] *= (rv[j][i] > 0.f) ? ga : gb;
          } else {
            v[i] += bj;
            v[i] = __builtin_fmaf(v[i], (v[i] > 0.f) ? g1 : g0, rv[j][i]);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rsY, (int)voff, (int)(s0 + (unsigned)i * drow + (unsigned)j * dcol), 0);
        }
      }
      if (q < 3) __syncthreads();    // (after the last pass: the barrier at the top of the item loop)
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // the prefetches the last chunk skipped, and the movers' offsets one item further on
    const int w_next = w_cur + nslots;
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int i = 0; i < NRAW; ++i) vraw_cur[i] = vraw_nxt[i];
      xb_cur = xb_nxt;
      load_raw(2);
    }
    u_pair = u_base(w_next);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_u(i, u_pair + (unsigned)i * u_plane, w_next < L);
    if constexpr (ROLE == 1) raw_offsets(w_next + nslots, vraw_nxt, xb_nxt);
  }
}

template <int MODE, int BOXW>
__global__ __launch_bounds__(512, 2) void wino44_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) body<MODE, 0, BOXW>(p, smem); else body<MODE, 1, BOXW>(p, smem);
}

// U = G g G^T (6x6) from the packed weight Wp[(kh * 3 + kw) * C + c][ldw] (cout contiguous).
//   FWD:   g[k][c] = W[k][c][.][.]                 input channels c, output channels k:  U[xi][c/8][(c%8)/4][k][c%4]
//   DGRAD: g'[c][k][a][b] = W[k][c][2 - a][2 - b]  input channels k (gy's), output channels c:  U[xi][k/8][(k%8)/4][c][k%4]
// One thread = four input channels x one output channel.
template <int MODE>
__global__ __launch_bounds__(256) void wino44_filter_kernel(const float* __restrict__ wp, float* __restrict__ U, int C, int K, int ldw) {
  const int cin = (MODE == MODE_FWD) ? C : K, cout = (MODE == MODE_FWD) ? K : C;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (cin >> 2) * cout) return;
  const int o = idx % cout, q4 = idx / cout;
  float g[9][4];
  if constexpr (MODE == MODE_FWD) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[t][j] = wp[(size_t)(t * C + 4 * q4 + j) * ldw + o];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)((8 - t) * C + o) * ldw + 4 * q4);
      g[t][0] = v.x; g[t][1] = v.y; g[t][2] = v.z; g[t][3] = v.w;
    }
  }
  const int nch = cin >> 3;
  float4* Uo = reinterpret_cast<float4*>(U);
  auto grow = [](int a, float g0, float g1, float g2) -> float {      // row a of G applied to (g0, g1, g2)
    return a == 0 ? 0.25f * g0
         : a == 1 ? (-1.f / 6.f) * (g0 + g1 + g2)
         : a == 2 ? (-1.f / 6.f) * (g0 - g1 + g2)
         : a == 3 ? (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2
         : a == 4 ? (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2
                  : g2;
  };
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float t[3][4];     // row a of G g: [column j of g][channel]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[j][e] = grow(a, g[0 * 3 + j][e], g[1 * 3 + j][e], g[2 * 3 + j][e]);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = grow(b, t[0][e], t[1][e], t[2][e]);
      const int xi = a * 6 + b;
      Uo[((size_t)(xi * nch + (q4 >> 1)) * 2 + (q4 & 1)) * cout + o] = make_float4(u[0], u[1], u[2], u[3]);
    }
  }
}

}  // namespace wino44
