// Winograd F(4x4, 3x3) for layers with 32 (or an odd multiple of 32) output channels -- StyleGAN2_512's 32 -> 32 channel layers
// at 512^2 (reference: models/gan/stylegan2/discriminator.py:60-76, layers.py:95-123), the largest single layer of its step
// (included by igemm.hip after wino44.h, whose transforms, stages, movers and item walk this is).
//
// wino44_kernel's block is 32 tiles x 64 couts with wave = (cout half, group of 9 planes); with one 32-wide cout block the
// second half has nothing to multiply.  Here an item is 32 tiles x 32 couts and the 36 planes are dealt over all eight waves:
// waves 0-3 (the transform waves) five each, waves 4-7 (the movers) four each -- waves s and s + 4 share SIMD s, nine planes
// per SIMD.  80 / 64 accumulator registers; the U ring holds a whole chunk pair.  Exchange area [row (8)][plane (36)][lane (64)]
// = both V stages, two passes of eight accumulator rows; in pass q wave w owns row 8 q + w.
// Where the time goes (ablations W44N_NO_*: timing only, results wrong; kernel alone, ms): 48 x 512^2 x 32 -> 32 forward 1.34: without the
// epilogue 0.96, without the input transform 1.07, without the raw stream 1.10, without all three 0.55 (the layer moves 3.2 GB: 0.6 ms
// at the rate the FIR kernels reach); 1536 x 4^2 x 512 -> 512 forward 0.45: 0.43 / 0.28 / 0.41 / 0.28 -- there the transform waves' five
// slots per chunk are the limiter.  Issuing every load of the epilogue and of the next item's start before the first store (the
// vector-memory counter retires in order) changed nothing: the passes do not wait for their stores.
#pragma once

namespace wino44n {

using wino44::Args; using wino44::Item; using wino44::bload4; using wino44::rsrc; using wino44::bt6; using wino44::at6;
using wino44::NT; using wino44::KQS; using wino44::PL; using wino44::V_SZ; using wino44::RPS; using wino44::RAW_SZ;
using wino44::RAW0; using wino44::LDS_DWORDS; using wino44::OOB;

// MODE: MODE_FWD / MODE_DGRAD (epilogue);  ROLE 0: transform waves (0-3), 1: movers (4-7);  BOXW: raw box width, 10 (8x8 maps: 2 x 2
// tiles of eight images per block), 18 (16x16 maps: 4 x 4 tiles of two images) or 34 (wider maps: 4 x 8 tiles of one image) --
// compile-time, so that the transform threads' 36 window offsets are immediates and the movers' piece -> pixel split divides
// by constants
template <int MODE, int ROLE, int BOXW>
__device__ __forceinline__ void body_n32(const Args& p, float* smem) {
  constexpr int TW = (BOXW == 34) ? 8 : (BOXW == 18) ? 4 : (BOXW == 10) ? 2 : 1, TH = (BOXW == 6) ? 1 : (BOXW == 10) ? 2 : 4;
  constexpr int NIMG = 32 / (TH * TW), BH = 4 * TH + 2;
  constexpr int SH_TW = (BOXW == 34) ? 3 : (BOXW == 18) ? 2 : (BOXW == 10) ? 1 : 0, SH_THW = SH_TW + ((BOXW == 6) ? 0 : (BOXW == 10) ? 1 : 2);
  // BOXW 6 (4x4 maps, this variant only: a tile is an image, 32 images per item): 32 boxes of 6 x 6 do not fit a raw stage, and all
  // they add to the 4 x 4 interiors is zeros -- the boxes share their halos: rows of 5 pixels (a row's right halo is the next
  // row's left one), images of 5 rows (an image's bottom halo row is the next image's top one), 8 dwords per pixel: image
  // pitch 200 dwords = 8 banks, the four tiles of a 32-lane read group stay conflict-free
  constexpr int RPSB = (BOXW == 6) ? 8 : RPS, ROWP = (BOXW == 6) ? 5 : BOXW, IMGP = (BOXW == 6) ? 25 : BH * BOXW;
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
#ifndef W44N_SWAP
#define W44N_SWAP 0
#endif
  constexpr int NP0 = W44N_SWAP ? 4 : 5;           // planes of a transform wave (dev knob W44N_SWAP: the movers take five -- same times on both layer kinds)
  constexpr int NPW = ROLE == 0 ? NP0 : 9 - NP0;   // planes of this wave: waves 0-3 five each (0 .. 19), waves 4-7 four each (20 .. 35)
  const int plane0 = ROLE == 0 ? NP0 * (w8 & 3) : 4 * NP0 + (9 - NP0) * (w8 & 3);      // (waves s and s + 4 share SIMD s: nine planes per SIMD)
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
    const int bpx = INTERIOR ? (px / (IH * IW)) * IMGP + ((px / IW) % IH + 1) * ROWP + px % IW + 1 : px;
    wraw[i] = bpx * RPSB + (piece & 1) * 4;         // (BOXW 34: pieces past the box land in the stage's padding: no branch)
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
    rd0 = RAW0 + (img * IMGP + 4 * ty * ROWP + 4 * tx) * RPSB + tch;
  }
  constexpr int rowstep = ROWP * RPSB;
  const int wrV = (tch >> 2) * KQS + ttile * 4 + (tch & 3);
  float d[6][6];
  // BOXW 6: the window's border ring is the zero halo of a 4x4 image -- known at compile time: 16 reads instead of 36, and B^T d
  // with d0 = d5 = 0 on four columns instead of six (the transform waves' share of a chunk is what limits the 4x4-map layers)
  constexpr bool ZB = BOXW == 6;
  auto bt6z = [](float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {      // bt6 with d0 = d5 = 0 on entry
    const float t0 = __builtin_fmaf(-5.f, d2, d4);
    const float p_ = __builtin_fmaf(-4.f, d2, d4), q_ = __builtin_fmaf(-4.f, d1, d3);
    const float r_ = d4 - d2, s_ = d3 - d1;
    const float t5 = __builtin_fmaf(-5.f, d3, 4.f * d1);
    d0 = t0; d1 = p_ + q_; d2 = p_ - q_; d3 = __builtin_fmaf(2.f, s_, r_); d4 = __builtin_fmaf(-2.f, s_, r_); d5 = t5;
  };
  auto tr_read = [&](int rstage) {                    // the thread's 6x6 raw window
    const float* src = smem + rd0 + rstage * RAW_SZ;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        d[i][j] = (ZB && (i == 0 || i == 5 || j == 0 || j == 5)) ? 0.f : src[i * rowstep + j * RPSB];
  };
  auto tr_cols = [&](int j0) {                        // B^T d, columns j0 .. j0 + 2
#pragma unroll
    for (int j = j0; j < j0 + 3; ++j) {
      if constexpr (ZB) { if (j != 0 && j != 5) bt6z(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]); }      // (columns 0 and 5 stay zero)
      else bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
    }
  };
  auto tr_row = [&](int vstage, int i) {              // (.) B for row i, six planes out
    if constexpr (ZB) bt6z(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    else bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    float* dst = smem + vstage * V_SZ + wrV + i * 6 * PL;
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j * PL] = d[i][j];
  };

  // ---- every wave: its U fragments, global -> registers (plane plane0 + i; lane: k-quad lhi, cout l31): a ring of
  // six float4, refilled in place six MFMA slots ahead ----
  const unsigned u_voff = (unsigned)((lhi * p.Cout + l31) * 16);
  const unsigned u_plane = (unsigned)(NCH * 2 * p.Cout * 16), u_step = (unsigned)(2 * p.Cout * 16);
  const unsigned u_grp = (unsigned)plane0 * u_plane;
  auto u_base = [&](int w) -> unsigned { return u_grp + (unsigned)((w % NKB) * 32 * 16); };      // chunk 0 of item w
  float4 ru[2 * NPW];      // the fragments of a whole chunk pair; entry s is refilled with slot s of the NEXT pair behind its MFMAs
  auto load_u = [&](int ring, unsigned soff, bool on) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(p.U, on);
    ru[ring] = bload4(rs, u_voff, soff);
  };

  // fragment reads of V: plane plane0 + xi, k-quad lhi, tile l31
  const int rdA = plane0 * PL + lhi * KQS + l31 * 4;

  f32x16 acc[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
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
  unsigned u_pair = u_base(w_cur);      // soffset of (plane plane0, first chunk of the current pair)
#pragma unroll
  for (int i = 0; i < 2 * NPW; ++i) load_u(i, u_pair + (unsigned)(i / NPW) * u_step + (unsigned)(i % NPW) * u_plane, true);

  const float g1 = p.gain, g0 = p.gain * p.slope;

  // A pair of chunks (t on V stage 0, t + 1 on stage 1) = 2 NPW slots of four MFMAs, one basic block.  Chunk c: transform waves raw
  // (c + 1) -> V stage (c + 1) & 1 (not in the item's last chunk: both V stages are the exchange area next) -- nine actions in
  // five slots;  movers: raw (c + 2) registers -> raw stage c & 1, raw (c + 3) into flight (the item's last one: after the
  // epilogue, like the U fragments of the next item).  LAST: the item's last pair (its own copy of the code).
  auto pair = [&](auto last_c, int t) {
    constexpr bool LAST = decltype(last_c)::value;
    float4 fa[2];
    fa[0] = *reinterpret_cast<const float4*>(smem + rdA);
#pragma unroll
    for (int s = 0; s < 2 * NPW; ++s) {
      const int P = s / NPW, xi = s - NPW * P;
      if (xi + 1 < NPW) fa[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + P * V_SZ + rdA + (xi + 1) * PL);
      if constexpr (ROLE == 1) {
#ifndef W44N_NO_RAW
        if (xi == 0) store_raw(P);
        if (xi == 1 && !(P == 1 && LAST)) load_raw(t + P + 3);
#endif
      } else if (!(P == 1 && LAST)) {
#ifndef W44N_NO_TRANSFORM
        if (xi == 0) tr_read(1 - P);
        if (xi == 1) { tr_cols(0); tr_cols(3); }
        if constexpr (NPW == 5) {
          if (xi == 2) { tr_row(1 - P, 0); tr_row(1 - P, 1); }
          if (xi == 3) { tr_row(1 - P, 2); tr_row(1 - P, 3); }
          if (xi == 4) { tr_row(1 - P, 4); tr_row(1 - P, 5); }
        } else {
          if (xi == 2) { tr_row(1 - P, 0); tr_row(1 - P, 1); tr_row(1 - P, 2); }
          if (xi == 3) { tr_row(1 - P, 3); tr_row(1 - P, 4); tr_row(1 - P, 5); }
        }
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* a = (const float*)&fa[s & 1];
      const float* b = (const float*)&ru[s];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // slot s of the next pair into the entry just consumed
      if (!LAST) load_u(s, u_pair + (unsigned)(2 + P) * u_step + (unsigned)xi * u_plane, true);
      if (xi == NPW - 1) {
        __syncthreads();
        if (P == 0) fa[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + V_SZ + rdA);
      }
    }
    u_pair += 2u * u_step;
  };

  for (; w_cur < L; w_cur += nslots) {
    __syncthreads();                 // raw 0 (and 1) of this item are in LDS; the exchange area is free again
#ifndef W44N_NO_TRANSFORM
    if constexpr (ROLE == 0) {
      tr_read(0); tr_cols(0); tr_cols(3);
#pragma unroll
      for (int i = 0; i < 6; ++i) tr_row(0, i);
    }
#endif
    __syncthreads();

    for (int t = 0; t + 2 < NCH; t += 2) pair(std::false_type{}, t);
    pair(std::true_type{}, NCH - 2);

    // ---- output transform through the exchange area (both V stages): [accumulator row of the pass (8)][plane (36)][lane (64)]
    // dwords.  Two passes of eight accumulator rows; in pass q wave w8 owns row r = 8 q + w8 = tile (r & 3) + 8 (r >> 2) + 4 lhi
    // = (w8 & 3) + 4 lhi + 8 (w8 >> 2) + 16 q of the item's 32 couts: reads all 36 M values of its element, A^T M A, epilogue,
    // 16 pixels x its cout.  (image, tile row, tile column) are bit fields of the tile index; only bit 2 (lhi) is per lane.
    const Item it = decode(w_cur);
    auto bit_off = [&](int bb) -> unsigned {      // offset contribution of tile-index bit bb (uniform)
      return bb < SH_TW ? (unsigned)((4 << bb) * p.ldo * 4)
             : bb < SH_THW ? (unsigned)((4 << (bb - SH_TW)) * p.W * p.ldo * 4)
                           : (unsigned)((1 << (bb - SH_THW)) * p.H * p.W * p.ldo * 4);
    };
    const unsigned lane_off = (lhi ? bit_off(2) : 0u) + (unsigned)(l31 * 4);
    const unsigned item_off = (unsigned)((((it.ph * TH * 4) * p.W + it.pw * TW * 4) * p.ldo + it.kb * 32) * 4)
                              + ((w8 & 1) ? bit_off(0) : 0u) + ((w8 & 2) ? bit_off(1) : 0u) + ((w8 & 4) ? bit_off(3) : 0u);
    float* ybase = p.y + (size_t)it.n_first * p.H * p.W * p.ldo;
    const float* rbase = p.ref ? p.ref + (size_t)it.n_first * p.H * p.W * p.ldo : ybase;
    const unsigned dcol = (unsigned)p.ldo * 4u, drow = (unsigned)(p.W * p.ldo) * 4u;
    const float bj = (MODE == MODE_FWD && p.bias) ? p.bias[it.kb * 32 + l31] : 0.f;
    // (with no second operand the loads below are off and return zeros: FWD adds them; DGRAD's two gains are then both 1)
    const float ga = p.ref ? g1 : 1.f, gb = p.ref ? g0 : 1.f;
    float* xw = smem + plane0 * 64 + lane;
    const float* xr = smem + (w8 * 36) * 64 + lane;
#ifdef W44N_NO_EPILOGUE
    if (p.N < 0)
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // (the element's image exists?  per lane only where a half-wave step of four tiles crosses images: 8x8 maps)
      const int tile_lo = (w8 & 3) + 8 * (w8 >> 2) + 16 * q;
      const unsigned voff = (it.n_first + ((tile_lo + 4 * lhi) >> SH_THW) < p.N) ? lane_off : OOB;
      const __amdgpu_buffer_rsrc_t rsY = rsrc(ybase, true);
      const __amdgpu_buffer_rsrc_t rsR = rsrc(rbase, p.ref != nullptr);
      const unsigned s0 = item_off + (q ? bit_off(4) : 0u);
      float rv[4][4];                // the epilogue's second operand goes into flight before the exchange
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rv[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)voff, (int)(s0 + (unsigned)i * drow + (unsigned)j * dcol), 0));
#pragma unroll
      for (int xi = 0; xi < NPW; ++xi)
#pragma unroll
        for (int r = 0; r < 8; ++r) xw[(r * 36 + xi) * 64] = acc[xi][8 * q + r];
      __syncthreads();
      float S[6][4];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        if ((a & 1) == 0) __builtin_amdgcn_sched_barrier(0);      // (two rows of M in flight at a time)
        float m[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) m[b] = xr[(a * 6 + b) * 64];
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
      if (q < 1) __syncthreads();    // (after the last pass: the barrier at the top of the item loop)
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // the prefetches the last pair skipped, and the movers' offsets one item further on
    const int w_next = w_cur + nslots;
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int i = 0; i < NRAW; ++i) vraw_cur[i] = vraw_nxt[i];
      xb_cur = xb_nxt;
      load_raw(2);
    }
    u_pair = u_base(w_next);
#pragma unroll
    for (int i = 0; i < 2 * NPW; ++i) load_u(i, u_pair + (unsigned)(i / NPW) * u_step + (unsigned)(i % NPW) * u_plane, w_next < L);
    if constexpr (ROLE == 1) raw_offsets(w_next + nslots, vraw_nxt, xb_nxt);
  }
}

template <int MODE, int BOXW>
__global__ __launch_bounds__(512, 2) void wino44n_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) body_n32<MODE, 0, BOXW>(p, smem); else body_n32<MODE, 1, BOXW>(p, smem);
}

}  // namespace wino44n
