// Forward of the 3x3 stride-2 pad-0 convolutions on (2G + 1) x (2G + 1) maps -- StyleGAN2's blurred down-sampling conv2 of
// every ResBlock (reference: models/gan/stylegan2/layers.py:174-198 = Blur -> EqualConv2d stride 2, discriminator.py:60-76) --
// as Winograd F(2x2, 2x2) over the four input phases with the structurally zero planes skipped (included by igemm.hip
// after wino22.h, whose machinery this is).
//
// y[ho][wo] = sum_{kh,kw < 3} b[2 ho + kh][2 wo + kw] w[kh][kw] is the 4x4 stride-2 pad-1 convolution of wino22.h with the
// filter w4 = [0 | w] (a zero first row and column) on an input with one more row and column: with the phases
// X_pq[i][j] = b[2i + p][2j + q] it is the sum of four stride-1 convolutions with 2x2 (p = q = 0), 1x2, 2x1 and 1x1 taps.
// Each runs as F(2x2, 2x2); for an odd phase the filter's first tap is zero, so row (column) 0 of U = G g G^T vanishes: the
// phases need 9, 6, 6 and 4 of the 9 planes -- 25 multiply-adds per 2x2 output tile and channel pair instead of the dense
// layer's 36.  The zero planes are skipped at compile time: an item's chunks run phase by phase, every phase its own copy
// of the chunk code (planes multiplied, planes the transform threads produce for the NEXT chunk, window rows / columns read).
//
// Block, waves, LDS stages, streams and the per-lane output transform are wino22_kernel<MODE_FWD>'s (128 tiles x 64 couts x 9
// xi, eight 32 x 32 sub-blocks with all nine accumulator tiles, transform waves 0-3 / movers 4-7).  New: PATCHES -- the
// 32^2 ... 256^2 output grids of StyleGAN2_512 do not fit a block's raw box, so an item is a patch of 8 x 16 tiles (16 x 32
// output pixels, box 17 x 33 phase pixels) of one image; grids of 16 / 8 / 4 keep whole images (2 / 8 / 32 per item).
#pragma once

namespace wino23 {

constexpr int TB = 128;                         // tiles per block
constexpr int VKQ = TB * 4;                     // dwords per (plane, k-quad) of V: 128 rows x 4; the kq = 1 half XOR-swizzled (rows ^ 4)
constexpr int VPL = 2 * VKQ;
constexpr int V_SZ = 9 * VPL;                   // 9 216 dwords
constexpr int UKQ = 64 * 4, UPL = 2 * UKQ;
constexpr int U_SZ = 9 * UPL;                   // 4 608 dwords
constexpr int BUF = V_SZ + U_SZ;                // one stage: 55 296 B
constexpr int RAW_PX = 800;                     // raw box capacity: 17 x 33 (patches), 2 x 17 x 17, 8 x 9 x 9, 32 x 5 x 5
constexpr int RAW_SZ = RAW_PX * 8;
constexpr int RAW0 = 2 * BUF;
constexpr int LDS_DWORDS = 2 * BUF + 2 * RAW_SZ;     // 161 792 B
constexpr unsigned OOB = 0x80000000u;
constexpr int NU = 5;                           // float4 pieces of U per thread and chunk: 1152 / 256 (all nine planes travel: the
                                                // filter kernel writes the structural zeros, only the MFMAs skip them)

struct Args {
  const float* x;      // input [N][Hi][Wi][ldi], Hi = 2 GH + 1, Wi = 2 GW + 1
  const float* U;      // [4 phases][9][Cin/8][2][Cout][4]
  float* y;            // output [N][GH][GW][ldo]
  const float* bias;   // [Cout] or NULL
  const float* ref;    // addend (y's layout) or NULL
  float slope, gain;
  int N, Hi, Wi, GH, GW, Cin, Cout, ldi, ldo;
  int TH, TW;          // 2x2-pixel tiles per image part of an item (powers of two, TH * TW * NIMG = 128)
  int sh_tw, sh_thw;   // log2(TW), log2(TH * TW)
  int NIMG;            // images per item (patches: 1)
  int PH, PW;          // patches per image
  int NP, NKB;         // patches in all (image groups x PH x PW), 64-wide cout blocks
};

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)0x80000000u : 0, 0x00020000);
}

// planes of phase ph = (p << 1 | q) that are not structurally zero: rows i >= p, columns j >= q of the 3 x 3 plane grid
__host__ __device__ constexpr bool active(int ph, int xi) { return (xi / 3) >= (ph >> 1) && (xi % 3) >= (ph & 1); }
__host__ __device__ constexpr int nactive(int ph) { return (3 - (ph >> 1)) * (3 - (ph & 1)); }

template <int ROLE, int NRAW>   // ROLE 0: transform waves (0-3), 1: movers (4-7)
__device__ __forceinline__ void body(const Args& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63, w8 = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = w8 & 3, wn = w8 >> 2;           // sub-block: tiles wm * 32 .., couts wn * 32 ..
  const int NCHc = p.Cin >> 3;                   // chunks per phase (even: Cin % 16 == 0)
  const int NCH = 4 * NCHc;                      // chunks per item
  const int NKB = p.NKB;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int L = (p.NP > xcd) ? ((p.NP - xcd + 7) >> 3) * NKB : 0;
  int w_cur = slot;
  if (w_cur >= L) return;
  const int BH = 2 * p.TH + 1, BW = 2 * p.TW + 1, BHW = BH * BW;      // raw box per image part (phase pixels)
  const int ppi = p.PH * p.PW;

  struct Item { int n_first, py, px, kb; };
  auto decode = [&](int w) -> Item {
    Item it;
    it.kb = w % NKB;
    const int patch = (w / NKB) * 8 + xcd;
    const int g = patch / ppi, pr = patch - g * ppi;
    it.n_first = g * p.NIMG;
    it.py = pr / p.PW; it.px = pr - it.py * p.PW;
    return it;
  };

  // ---- movers: raw box pieces (pixel * 2 + k-quad) ----
  int rfix[NRAW], rpk[NRAW];      // rpk = image << 16 | box row << 8 | box column (image 32767: no such pixel)
  unsigned vu[NU];
  const int npx = p.NIMG * BHW;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int item = tid + 256 * i, px = item >> 1, q = item & 1;
      const int im = px / BHW, rem = px - im * BHW, rr = rem / BW, rc = rem - rr * BW;
      rpk[i] = (px < npx) ? (im << 16 | rr << 8 | rc) : 0x7FFF0000;
      rfix[i] = ((im * p.Hi + 2 * rr) * p.Wi + 2 * rc) * p.ldi * 4 + q * 16;
    }
  }
  if constexpr (ROLE == 0) {      // (the transform waves carry the U stream: the movers' registers are full of raw pieces)
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int item = tid + 256 * i, co = item & 63, q = (item >> 6) & 1, pl = item >> 7;
      vu[i] = (item < 9 * 128) ? (unsigned)((((pl * NCHc) * 2 + q) * p.Cout + co) * 16) : OOB;
    }
  }
  // streams: raw 3 chunks ahead, U 2 chunks ahead of the chunk being multiplied.  Chunk t of an item -> (phase t / NCHc,
  // channels (t % NCHc) * 8)
  unsigned vraw[NRAW];
  const float* xb_raw = nullptr;
  int t_raw = 0, w_raw = 0, ph_raw = 0, cc_raw = 0, nleft_raw = 0, r0_raw = 0, c0_raw = 0;
  auto raw_phase = [&](int ph) {      // per-thread offsets of input phase (p, q) = (ph >> 1, ph & 1) of the stream's item
    // box row rr <-> phase row r0 + rr + s_p (s_1 = -1, s_0 = 0) <-> input row 2 (r0 + rr + s_p) + p = 2 (r0 + rr) - p
    const int dh = -(ph >> 1), dw = -(ph & 1);
    const int soff = ((2 * r0_raw + dh) * p.Wi + 2 * c0_raw + dw) * p.ldi * 4;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int rr = (rpk[i] >> 8) & 255, rc = rpk[i] & 255;
      const bool ok = (unsigned)(2 * (r0_raw + rr) + dh) < (unsigned)p.Hi && (unsigned)(2 * (c0_raw + rc) + dw) < (unsigned)p.Wi &&
                      (rpk[i] >> 16) < nleft_raw;
      vraw[i] = ok ? (unsigned)(rfix[i] + soff) : OOB;
    }
  };
  auto raw_item = [&](int w) {
    if (w < L) {
      const Item it = decode(w);
      xb_raw = p.x + (size_t)it.n_first * p.Hi * p.Wi * p.ldi;
      ph_raw = 0; cc_raw = 0;
      nleft_raw = p.N - it.n_first;
      r0_raw = it.py * 2 * p.TH; c0_raw = it.px * 2 * p.TW;
      raw_phase(0);
    } else {
      xb_raw = nullptr;
    }
  };
  unsigned u_off = 0, u_cur = 0; int t_u = 0, cc_u = 0, w_u = 0; bool u_on = true;
  auto u_item = [&](int w) {
    u_on = w < L;
    if (u_on) u_off = (unsigned)(((w % NKB) * 64) * 16);
    u_cur = u_off; cc_u = 0;
  };
  const unsigned u_step = (unsigned)(2 * p.Cout * 16), u_phase = (unsigned)(9 * NCHc * 2 * p.Cout * 16);

  float4 rraw[NRAW], ru[NU];
  auto load_raw = [&]() {
    const __amdgpu_buffer_rsrc_t rs = rsrc(xb_raw, xb_raw != nullptr);
#pragma unroll
    for (int i = 0; i < NRAW; ++i) rraw[i] = bload4(rs, vraw[i], (unsigned)cc_raw * 32u);
    ++t_raw;
    if (++cc_raw == NCHc) {
      cc_raw = 0;
      if (t_raw == NCH) { t_raw = 0; w_raw += nslots; raw_item(w_raw); }
      else { ++ph_raw; raw_phase(ph_raw); }      // next input phase of the same item
    }
  };
  auto store_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NRAW; ++i)
      if (tid + 256 * i < 2 * npx) *reinterpret_cast<float4*>(smem + RAW0 + stage * RAW_SZ + (tid + 256 * i) * 4) = rraw[i];
  };
  auto load_u = [&]() {
    const __amdgpu_buffer_rsrc_t rs = rsrc(p.U, u_on);
#pragma unroll
    for (int i = 0; i < NU; ++i) ru[i] = bload4(rs, vu[i], u_cur);
    u_cur += u_step;
    if (++cc_u == NCHc) { cc_u = 0; u_cur += u_phase - (unsigned)NCHc * u_step; }      // on to the next input phase's slice
    if (++t_u == NCH) { t_u = 0; w_u += nslots; u_item(w_u); }
  };
  auto store_u = [&](int bufoff) {
#pragma unroll
    for (int i = 0; i < NU; ++i)
      if (tid + 256 * i < 9 * 128) *reinterpret_cast<float4*>(smem + bufoff + V_SZ + (tid + 256 * i) * 4) = ru[i];
  };

  // ---- transform waves: (k-quad, tile): a whole 3 x 3 window per thread ----
  const int kq = tid & 1, tile = tid >> 1;
  int rd0 = 0;
  if constexpr (ROLE == 0) {
    const int img = tile >> p.sh_thw, ty = (tile >> p.sh_tw) & (p.TH - 1), tx = tile & (p.TW - 1);
    rd0 = ((img * BH + 2 * ty) * BW + 2 * tx) * 8 + kq * 4;
  }
  const int rowstep = BW * 8;
  const int wrV = kq * VKQ + ((tile ^ (kq * 4)) * 4);
  float4 d[3][3];
  auto sub4 = [](const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  // the window / transform / planes of a chunk of phase PH: rows r >= p, columns c >= q only
  auto read_raw = [&](auto phc, int stage) {
    constexpr int PH = decltype(phc)::value;
#pragma unroll
    for (int r = PH >> 1; r < 3; ++r)
#pragma unroll
      for (int c = PH & 1; c < 3; ++c) d[r][c] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rd0 + r * rowstep + c * 8);
  };
  auto row_ops = [&](auto phc) {      // B^T d: rows 0, 2 minus row 1
    constexpr int PH = decltype(phc)::value;
#pragma unroll
    for (int c = PH & 1; c < 3; ++c) {
      if constexpr ((PH >> 1) == 0) d[0][c] = sub4(d[0][c], d[1][c]);
      d[2][c] = sub4(d[2][c], d[1][c]);
    }
  };
  auto col_ops_store = [&](auto phc, int bufoff, int r) {      // (.) B for window row r: columns 0, 2 minus column 1
    constexpr int PH = decltype(phc)::value;
    if constexpr ((PH & 1) == 0) *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 0) * VPL + wrV) = sub4(d[r][0], d[r][1]);
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 1) * VPL + wrV) = d[r][1];
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 2) * VPL + wrV) = sub4(d[r][2], d[r][1]);
  };

  // fragment reads (quad layout: lane half = k-quad; the kq = 1 plane's rows are XOR 4)
  const int rdA = lhi * VKQ + (((wm * 32 + l31) ^ (lhi * 4)) * 4);
  const int rdB = V_SZ + lhi * UKQ + (wn * 32 + l31) * 4;

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- prologue ----
  if constexpr (ROLE == 1) {
    w_raw = w_cur; raw_item(w_raw);
    load_raw();            // raw 0
    store_raw(0);
    load_raw();            // raw 1
  } else {
    w_u = w_cur; u_item(w_u);
    load_u();              // U 0
    store_u(0);
    load_u();              // U 1
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    read_raw(std::integral_constant<int, 0>{}, 0);
    row_ops(std::integral_constant<int, 0>{});
#pragma unroll
    for (int r = 0; r < 3; ++r) col_ops_store(std::integral_constant<int, 0>{}, 0, r);
  } else {
    store_raw(1);
    load_raw();            // raw 2
  }
  __syncthreads();

  // One chunk of phase PH on stage P: its active planes' MFMAs; the transform waves turn the NEXT chunk's raw box into V --
  // phase NPH = PH, or the next phase after the phase's last chunk (phase 0 of the next item after the item's last: LAST, whose
  // prefetch loads are issued after the epilogue -- their registers are what its second operand needs).
  auto chunk = [&](auto par, auto phc, auto nphc, auto last_c) {
    constexpr int P = decltype(par)::value, PH = decltype(phc)::value;
    constexpr bool last = decltype(last_c)::value;
    constexpr int NPH = decltype(nphc)::value;
    constexpr int cur = P * BUF, nxt = BUF - cur;
    constexpr int S = nactive(PH);                // slots of this chunk: 9 / 6 / 6 / 4
    constexpr int R0 = NPH >> 1;                  // first window row the next chunk's transform produces
    // active planes, in order
    constexpr int first = (PH >> 1) * 3 + (PH & 1);
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(smem + cur + rdA + first * VPL);
    fb[0] = *reinterpret_cast<const float4*>(smem + cur + rdB + first * UPL);
    int s = 0;      // (compile-time after unrolling)
#pragma unroll
    for (int xi = 0; xi < 9; ++xi) {
      if (!active(PH, xi)) continue;
      // the next active plane
      int nx = xi + 1;
      while (nx < 9 && !active(PH, nx)) ++nx;
      if (nx < 9) {
        fa[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdA + nx * VPL);
        fb[(s + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdB + nx * UPL);
      }
      if constexpr (ROLE == 0) {
        if (s == 0) read_raw(std::integral_constant<int, NPH>{}, 1 - P);
        if (s == 1) row_ops(std::integral_constant<int, NPH>{});
        if (s == 1 && S == 4) col_ops_store(std::integral_constant<int, NPH>{}, nxt, R0);          // (4 slots: two actions per slot)
        if (s == 2) { if (S == 4) { for (int r = R0 + 1; r < 3; ++r) col_ops_store(std::integral_constant<int, NPH>{}, nxt, r); }
                      else col_ops_store(std::integral_constant<int, NPH>{}, nxt, R0); }
        if (S > 4 && s == 3) col_ops_store(std::integral_constant<int, NPH>{}, nxt, R0 + 1);
        if (S > 4 && s == 4 && R0 == 0) col_ops_store(std::integral_constant<int, NPH>{}, nxt, 2);
        if (s == S - 2) store_u(nxt);
        if (s == S - 1 && !last) load_u();
      } else {
        if (s == 0) store_raw(P);
        if (s == 1 && !last) load_raw();
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* a = (const float*)&fa[s & 1];
      const float* b = (const float*)&fb[s & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ++s;
    }
    __syncthreads();
  };
  // all chunks of phase PH of an item
  auto phase = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    constexpr int NX = (PH + 1) & 3;
    for (int t = 0; t + 2 < NCHc; t += 2) {
      chunk(std::integral_constant<int, 0>{}, phc, phc, std::false_type{});
      chunk(std::integral_constant<int, 1>{}, phc, phc, std::false_type{});
    }
    chunk(std::integral_constant<int, 0>{}, phc, phc, std::false_type{});
    chunk(std::integral_constant<int, 1>{}, phc, std::integral_constant<int, NX>{}, std::integral_constant<bool, PH == 3>{});
  };

  const float g1 = p.gain, g0 = p.gain * p.slope;
  // offset contribution of tile-index bit b (tiles: [image][ty][tx], each tile 2 x 2 output pixels)
  auto bit_off = [&](int b) -> unsigned {
    return b < p.sh_tw ? (unsigned)((2 << b) * p.ldo * 4)
           : b < p.sh_thw ? (unsigned)((2 << (b - p.sh_tw)) * p.GW * p.ldo * 4)
                          : (unsigned)((1 << (b - p.sh_thw)) * p.GH * p.GW * p.ldo * 4);
  };

  for (; w_cur < L; w_cur += nslots) {
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    // ---- output transform (per lane): s_a = m_a0 + m_a1, s'_a = m_a1 + m_a2;  Y00 = s_0 + s_1, Y10 = s_1 + s_2, Y01 = s'_0 + s'_1, Y11 = s'_1 + s'_2
    const Item it = decode(w_cur);
    const int cout = it.kb * 64 + wn * 32 + l31;
    float* ybase = p.y + (size_t)it.n_first * p.GH * p.GW * p.ldo;
    const __amdgpu_buffer_rsrc_t rsY = rsrc(ybase, true);
    // (no addend: its loads are off and return zeros: no branch)
    const __amdgpu_buffer_rsrc_t rsR = rsrc(p.ref ? p.ref + (size_t)it.n_first * p.GH * p.GW * p.ldo : ybase, p.ref != nullptr);
    const unsigned dcol = (unsigned)p.ldo * 4u, drow = (unsigned)(p.GW * p.ldo) * 4u;
    // byte offset of accumulator row r = lane part (one register) + a wave-uniform part in the instructions' scalar offset
    const unsigned lane_off = (lhi ? bit_off(2) : 0u) + (unsigned)((wn * 32 + l31) * 4);
    const unsigned wave_off = ((wm & 1) ? bit_off(5) : 0u) + ((wm & 2) ? bit_off(6) : 0u) + (unsigned)(it.kb * 64 * 4) +
                              (unsigned)(((it.py * 2 * p.TH) * p.GW + it.px * 2 * p.TW) * p.ldo * 4);
    const int lane_img = (wm * 32 + 4 * lhi) >> p.sh_thw;
    const int img_lim = p.N - it.n_first - lane_img;
    const float bj = p.bias ? p.bias[cout] : 0.f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {       // four quarters of 4 accumulator rows: the addend of 4 rows in flight at a time
      float rv[4][4];
      unsigned vo[4], so[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = h * 4 + q, rbits = (r & 3) + 8 * (r >> 2);
        vo[q] = ((rbits >> p.sh_thw) < img_lim) ? lane_off : OOB;
        so[q] = wave_off + ((r & 1) ? bit_off(0) : 0u) + ((r & 2) ? bit_off(1) : 0u) + ((r & 4) ? bit_off(3) : 0u) + ((r & 8) ? bit_off(4) : 0u);
        rv[q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)vo[q], (int)so[q], 0));
        rv[q][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)vo[q], (int)(so[q] + dcol), 0));
        rv[q][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)vo[q], (int)(so[q] + drow), 0));
        rv[q][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)vo[q], (int)(so[q] + drow + dcol), 0));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        const int r = h * 4 + q;
        float s[3], s2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float m0 = acc[a * 3 + 0][r], m1 = acc[a * 3 + 1][r], m2 = acc[a * 3 + 2][r];
          s[a] = m0 + m1; s2[a] = m1 + m2;
        }
        float v[4] = {s[0] + s[1], s2[0] + s2[1], s[1] + s[2], s2[1] + s2[2]};      // (y00, y01, y10, y11)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += bj;
          v[e] = __builtin_fmaf(v[e], (v[e] > 0.f) ? g1 : g0, rv[q][e]);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rsY, (int)vo[q], (int)so[q], 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[1]), rsY, (int)vo[q], (int)(so[q] + dcol), 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[2]), rsY, (int)vo[q], (int)(so[q] + drow), 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[3]), rsY, (int)vo[q], (int)(so[q] + drow + dcol), 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ROLE == 1) load_raw(); else load_u();      // the prefetch the last chunk skipped
  }
}

template <int NRAW>
__global__ __launch_bounds__(512, 2) void wino23_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) body<0, NRAW>(p, smem); else body<1, NRAW>(p, smem);
}

// U_ph = G g_ph G^T (G = [[1,0],[1,1],[0,1]]) per input phase (p, q) from the packed 3x3 weight Wp[(kh * 3 + kw) * C + c][ldw]:
// g[a][b] = w4[kh(p,a)][kw(q,b)], kh(1,a) = 2a, kh(0,a) = 1 + 2a (wino22.h), w4[k][l] = w[k - 1][l - 1] and 0 for k = 0 or l = 0.
// U[phase][xi][cin / 8][(cin % 8) / 4][cout][cin % 4].  One thread = one phase x four input channels x one output channel.
__global__ __launch_bounds__(256) void wino23_filter_kernel(const float* __restrict__ wp, float* __restrict__ U, int C, int K, int ldw) {
  const int per = (C >> 2) * K;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * per) return;
  const int ph = idx / per, rem = idx - ph * per;
  const int o = rem % K, q4 = rem / K;
  const int phh = ph >> 1, phw = ph & 1;
  float g[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kh = (phh ? 2 * a : 1 + 2 * a) - 1, kw = (phw ? 2 * b : 1 + 2 * b) - 1;      // tap of the 3x3 filter (-1: the zero row / column)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[a][b][j] = (kh < 0 || kw < 0) ? 0.f : wp[(size_t)((kh * 3 + kw) * C + 4 * q4 + j) * ldw + o];
    }
  const int nch = C >> 3;
  float4* Uo = reinterpret_cast<float4*>(U) + (size_t)ph * 9 * nch * 2 * K;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r0 = j == 0 ? g[0][0][e] : j == 1 ? g[0][0][e] + g[0][1][e] : g[0][1][e];      // (g G^T) rows 0, 1
        const float r1 = j == 0 ? g[1][0][e] : j == 1 ? g[1][0][e] + g[1][1][e] : g[1][1][e];
        u[e] = i == 0 ? r0 : i == 1 ? r0 + r1 : r1;
      }
      Uo[((size_t)((i * 3 + j) * nch + (q4 >> 1)) * 2 + (q4 & 1)) * K + o] = make_float4(u[0], u[1], u[2], u[3]);
    }
}

}  // namespace wino23
