// Winograd F(2x2, 2x2) in exact fp32 on v_mfma_f32_32x32x2_f32 for the 4x4 stride-2 pad-1 layers (included by igemm.hip).
//
// Replaces F.conv2d forward / input gradient of SNDCGAN's strided layers and the generator's ConvTranspose2d
// (reference: models/gan/sndcgan.py:26-38,91-109).  Round 6, after the 3x3 layers (wino.h): these three layers were 48 % of
// the headline step.  A 4x4 stride-2 pad-1 convolution is the SUM of four 2x2 stride-1 convolutions, one per phase
// (row parity p, column parity q) of the input, X_pq[i][j] = x[2i + p][2j + q]:
//
//   y[ho][wo] = sum_pq sum_{a,b in {0,1}} X_pq[ho + a + s_p][wo + b + s_q] w[kh(p,a)][kw(q,b)]     s_1 = -1, s_0 = 0
//                                                                                                  kh(1,a) = 2a, kh(0,a) = 1 + 2a
//
// and each of them runs as F(2x2, 2x2): 9 multiply-adds per 2x2 output tile and channel pair instead of 16 (1.78x fewer),
// accumulated over the phases in the transform domain:  Y = A^T [ sum_pq sum_c (G g_pq G^T) (.) (B^T d_pq B) ] A  with
// B^T = [[1,-1,0],[0,1,0],[0,-1,1]], G = [[1,0],[1,1],[0,1]], A^T = [[1,1,0],[0,1,1]] (no constants other than +-1).
// The data gradient is the same machinery the other way round: each PHASE of dx is a 2x2 stride-1 correlation of gy,
// dx[2i + ph][2j + pw] = sum_{al,be} gy[i + al + s'][j + be + s'] w[kh'(ph,al)][kw'(pw,be)] (s' = -1 for phase 0, 0 for phase 1;
// kh'(0,al) = 3 - 2 al, kh'(1,al) = 2 - 2 al): one item per output phase, input unstrided, output strided.
//
// Block (512 threads, one per CU, persistent) = 128 tiles x 64 output channels x all 9 xi: eight waves, each a 32 tile x
// 32 cout sub-block with all nine accumulator tiles (144 registers): the output transform is per-lane, nothing is
// exchanged.  Contraction in 8-channel chunks of one phase through a double-buffered LDS stage, as in wino.h; waves 0-3
// transform (raw box -> B^T d B -> V, a whole 3x3 tile per thread), waves 4-7 move (phase image and U, global ->
// registers -> LDS).  The raw box of a chunk is the phase image of the block's images with ONE padding row / column on the
// side the phase's window hangs over ((G+1) x (G+1) pixels per image, the padding loaded as hardware zero fills), so the
// transform threads read fixed offsets.  Maps: output (FWD) / gy (DGRAD) grids of 4, 8 or 16 -- SNDCGAN's.
#pragma once

namespace wino22 {

constexpr int TB = 128;                         // tiles per block
constexpr int VKQ = TB * 4;                     // dwords per (plane, k-quad) of V: 128 rows x 4; the kq = 1 half XOR-swizzled (rows ^ 4)
constexpr int VPL = 2 * VKQ;                    // per plane
constexpr int V_SZ = 9 * VPL;                   // 9 216 dwords
constexpr int UKQ = 64 * 4, UPL = 2 * UKQ;
constexpr int U_SZ = 9 * UPL;                   // 4 608 dwords
constexpr int BUF = V_SZ + U_SZ;                // one stage: 55 296 B
constexpr int RAW_PX = 800;                     // raw box capacity: 32 images x 5 x 5 (4x4 grids), 8 x 9 x 9, 2 x 17 x 17
constexpr int RAW_SZ = RAW_PX * 8;
constexpr int RAW0 = 2 * BUF;
constexpr int LDS_DWORDS = 2 * BUF + 2 * RAW_SZ;     // 161 792 B
constexpr unsigned OOB = 0x80000000u;
constexpr int NU = 5;                           // float4 pieces of U per thread and chunk: 1152 / 256
// (raw box pieces per mover thread and chunk = template parameter NRAW: 5 / 6 / 7 for grids of 16 / 8 / 4 -- 1156 / 1296 / 1600 items)

struct Args {
  const float* x;      // input [N][Hi][Wi][ldi]   (FWD: x, Hi = 2 G;  DGRAD: gy, Hi = G)
  const float* U;      // [4 phases][9][Cin/8][2][Cout][4]
  float* y;            // output [N][Hout][Wout][ldo]   (FWD: G x G;  DGRAD: dx, 2G x 2G)
  const float* bias;   // FWD or NULL
  const float* ref;    // FWD: addend;  DGRAD: act_ref;  output layout;  or NULL
  float slope, gain;
  int N, Hi, Wi, Hout, Wout, Cin, Cout, ldi, ldo;
  int GH, GW;          // the tile grid's image: FWD the output map, DGRAD the gy map (4, 8 or 16)
  int sh_tw, sh_thw;   // log2 tiles per row / per image (GW/2, GH/2 * GW/2)
  int NIMG;            // images per block = 128 / tiles per image
  int NTB, NKB;        // tile blocks (image groups), 64-wide cout blocks
  int dgrad;           // 0: FWD (four input phases contracted, dense output);  1: DGRAD (one output phase per item)
};

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)0x80000000u : 0, 0x00020000);
}

template <int MODE, int ROLE, int NRAW>   // ROLE 0: transform waves (0-3), 1: movers (4-7)
__device__ __forceinline__ void body(const Args& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63, w8 = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = w8 & 3, wn = w8 >> 2;           // sub-block: tiles wm * 32 .., couts wn * 32 ..
  constexpr bool DG = (MODE == MODE_DGRAD);
  const int NCHc = p.Cin >> 3;                   // chunks per phase
  const int NCH = DG ? NCHc : 4 * NCHc;          // chunks per item (even: Cin % 16 == 0 for DGRAD)
  const int NSUB = DG ? 4 * p.NKB : p.NKB;       // items per tile block: (output phase,) cout block
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int L = (p.NTB > xcd) ? ((p.NTB - xcd + 7) >> 3) * NSUB : 0;
  int w_cur = slot;
  if (w_cur >= L) return;
  const int BW = p.GW + 1, BHW = (p.GH + 1) * BW;      // raw box per image
  const int RS = DG ? 1 : 2;                           // input rows per grid row

  struct Item { int n_first, kb, ph; };               // ph: DGRAD output phase (ph_h * 2 + ph_w)
  auto decode = [&](int w) -> Item {
    Item it;
    const int sub = w % NSUB;
    it.kb = DG ? (sub >> 2) : sub;
    it.ph = DG ? (sub & 3) : 0;
    it.n_first = ((w / NSUB) * 8 + xcd) * p.NIMG;
    return it;
  };

  // ---- movers ----
  int rfix[NRAW], rpk[NRAW];      // rpk = image << 16 | box row << 8 | box column (0xFFFFFF..: no such pixel)
  unsigned vu[NU];
  const int npx = p.NIMG * BHW;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int item = tid + 256 * i, px = item >> 1, q = item & 1;
      const int im = px / BHW, rem = px - im * BHW, rr = rem / BW, rc = rem - rr * BW;
      rpk[i] = (px < npx) ? (im << 16 | rr << 8 | rc) : 0x7FFF0000;      // (image 32767: never present)
      rfix[i] = ((im * p.Hi + RS * rr) * p.Wi + RS * rc) * p.ldi * 4 + q * 16;
    }
  }
  if constexpr (ROLE == 0) {      // (the transform waves carry the U stream: the movers' registers are full of raw pieces)
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int item = tid + 256 * i, co = item & 63, q = (item >> 6) & 1, pl = item >> 7;
      vu[i] = (item < 9 * 128) ? (unsigned)((((pl * NCHc) * 2 + q) * p.Cout + co) * 16) : OOB;
    }
  }
  // streams: raw 3 chunks ahead, U 2 chunks ahead of the chunk being multiplied.  A chunk = (item, t): FWD t -> (phase t / NCHc,
  // channels (t % NCHc) * 8);  DGRAD: the item's phase, channels t * 8
  unsigned vraw[NRAW];
  const float* xb_raw = nullptr;
  int t_raw = 0, w_raw = 0, ph_raw = 0, cc_raw = 0, nleft_raw = 0;      // (nleft: images of the stream's item that exist)
  auto raw_phase = [&](int ph) {      // per-thread offsets for input phase / window shift ph (FWD: p = ph >> 1, q = ph & 1)
    // FWD: rows 2 r + dh, dh = -1 (p = 1) | 0 (p = 0);  DGRAD (output phase): rows r + dh, dh = -1 (phase 0) | 0 (phase 1)
    const int phh = ph >> 1, phw = ph & 1;
    const int dh = DG ? (phh ? 0 : -1) : (phh ? -1 : 0), dw = DG ? (phw ? 0 : -1) : (phw ? -1 : 0);
    const int soff = (dh * p.Wi + dw) * p.ldi * 4;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int rr = (rpk[i] >> 8) & 255, rc = rpk[i] & 255;
      const bool ok = (unsigned)(RS * rr + dh) < (unsigned)p.Hi && (unsigned)(RS * rc + dw) < (unsigned)p.Wi && (rpk[i] >> 16) < nleft_raw;
      vraw[i] = ok ? (unsigned)(rfix[i] + soff) : OOB;
    }
  };
  auto raw_item = [&](int w) {
    if (w < L) {
      const Item it = decode(w);
      xb_raw = p.x + (size_t)it.n_first * p.Hi * p.Wi * p.ldi;
      ph_raw = DG ? it.ph : 0; cc_raw = 0;
      nleft_raw = p.N - it.n_first;
      raw_phase(ph_raw);
    } else {
      xb_raw = nullptr;
    }
  };
  unsigned u_off = 0, u_cur = 0; int t_u = 0, cc_u = 0, w_u = 0; bool u_on = true;
  auto u_item = [&](int w) {
    u_on = w < L;
    if (u_on) {
      const Item it = decode(w);
      u_off = (unsigned)((it.ph * 9 * NCHc * 2 * p.Cout + it.kb * 64) * 16);
    }
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
      else { ++ph_raw; raw_phase(ph_raw); }      // FWD: next input phase of the same item
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
    if (++cc_u == NCHc) { cc_u = 0; u_cur += u_phase - (unsigned)NCHc * u_step; }      // FWD: on to the next input phase's slice
    if (++t_u == NCH) { t_u = 0; w_u += nslots; u_item(w_u); }
  };
  auto store_u = [&](int bufoff) {
#pragma unroll
    for (int i = 0; i < NU; ++i)
      if (tid + 256 * i < 9 * 128) *reinterpret_cast<float4*>(smem + bufoff + V_SZ + (tid + 256 * i) * 4) = ru[i];
  };

  // ---- transform waves: (k-quad, tile): a whole 3 x 3 tile per thread ----
  const int kq = tid & 1, tile = tid >> 1;
  int rd0 = 0;
  if constexpr (ROLE == 0) {
    const int img = tile >> p.sh_thw, ty = (tile >> p.sh_tw) & ((p.GH >> 1) - 1), tx = tile & ((p.GW >> 1) - 1);
    rd0 = ((img * (p.GH + 1) + 2 * ty) * BW + 2 * tx) * 8 + kq * 4;
  }
  const int rowstep = BW * 8;
  const int wrV = kq * VKQ + ((tile ^ (kq * 4)) * 4);
  float4 d[3][3];
  auto read_raw = [&](int stage) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) d[r][c] = *reinterpret_cast<const float4*>(smem + RAW0 + stage * RAW_SZ + rd0 + r * rowstep + c * 8);
  };
  auto sub4 = [](const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  auto row_ops = [&]() {      // B^T d: rows 0, 2 minus row 1
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[0][c] = sub4(d[0][c], d[1][c]); d[2][c] = sub4(d[2][c], d[1][c]); }
  };
  auto col_ops_store = [&](int bufoff, int r) {      // (.) B for tile row r: columns 0, 2 minus column 1; three planes out
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 0) * VPL + wrV) = sub4(d[r][0], d[r][1]);
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
    read_raw(0);
    row_ops();
#pragma unroll
    for (int r = 0; r < 3; ++r) col_ops_store(0, r);
  } else {
    store_raw(1);
    load_raw();            // raw 2
  }
  __syncthreads();

  // (last: the item's last chunk.  Its prefetch loads are issued AFTER the epilogue instead -- their 28 / 20 registers are what
  // the epilogue's second operand needs; with them in flight across the epilogue the kernel spilled 70 - 100 registers, part
  // of them inside these loops)
  auto chunk = [&](auto par, auto last_c) {
    constexpr int P = decltype(par)::value;
    constexpr bool last = decltype(last_c)::value;
    constexpr int cur = P * BUF, nxt = BUF - cur;
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(smem + cur + rdA);
    fb[0] = *reinterpret_cast<const float4*>(smem + cur + rdB);
#pragma unroll
    for (int xi = 0; xi < 9; ++xi) {
      if (xi + 1 < 9) {
        fa[(xi + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdA + (xi + 1) * VPL);
        fb[(xi + 1) & 1] = *reinterpret_cast<const float4*>(smem + cur + rdB + (xi + 1) * UPL);
      }
      if constexpr (ROLE == 0) {      // (W22_ABL_*: ablation switches of tools/dev/ab_variants.sh -- timing only, results wrong)
#ifndef W22_ABL_NO_TR
        if (xi == 0) read_raw(1 - P);
        if (xi == 2) row_ops();
        if (xi >= 3 && xi < 6) col_ops_store(nxt, xi - 3);
#endif
#ifndef W22_ABL_NO_U
        if (xi == 6) store_u(nxt);
        if (xi == 7 && !last) load_u();
#endif
      } else {
#ifndef W22_ABL_NO_RAW
        if (xi == 0) store_raw(P);
        if (xi == 1 && !last) load_raw();
#endif
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

  const float g1 = p.gain, g0 = p.gain * p.slope;
  const int OS = DG ? 2 : 1;                         // output pixels per grid step
  // offset contribution of tile-index bit b (tiles: [image][ty][tx], each tile 2 x 2 grid points = 2 OS x 2 OS output pixels)
  auto bit_off = [&](int b) -> unsigned {
    return b < p.sh_tw ? (unsigned)((2 << b) * OS * p.ldo * 4)
           : b < p.sh_thw ? (unsigned)((2 << (b - p.sh_tw)) * OS * p.Wout * p.ldo * 4)
                          : (unsigned)((1 << (b - p.sh_thw)) * p.Hout * p.Wout * p.ldo * 4);
  };

  for (; w_cur < L; w_cur += nslots) {
    for (int t = 0; t + 2 < NCH; t += 2) {
      chunk(std::integral_constant<int, 0>{}, std::false_type{});
      chunk(std::integral_constant<int, 1>{}, std::false_type{});
    }
    chunk(std::integral_constant<int, 0>{}, std::false_type{});      // (the item's last pair: its own copy, no branch inside a chunk)
    chunk(std::integral_constant<int, 1>{}, std::true_type{});
    // ---- output transform (per lane): s_a = m_a0 + m_a1, s'_a = m_a1 + m_a2;  Y00 = s_0 + s_1, Y10 = s_1 + s_2, Y01 = s'_0 + s'_1, Y11 = s'_1 + s'_2
    const Item it = decode(w_cur);
    const int cout = it.kb * 64 + wn * 32 + l31;
    float* ybase = p.y + (size_t)it.n_first * p.Hout * p.Wout * p.ldo;
    const __amdgpu_buffer_rsrc_t rsY = rsrc(ybase, true);
    // (no second operand: its loads are off and return zeros -- FWD adds them, DGRAD's two gains are then both 1: no branch)
    const __amdgpu_buffer_rsrc_t rsR = rsrc(p.ref ? p.ref + (size_t)it.n_first * p.Hout * p.Wout * p.ldo : ybase, p.ref != nullptr);
    const float ga = p.ref ? g1 : 1.f, gb = p.ref ? g0 : 1.f;
    const unsigned dcol = (unsigned)(OS * p.ldo) * 4u, drow = (unsigned)(OS * p.Wout * p.ldo) * 4u;
    const unsigned ph_off = DG ? (unsigned)((((it.ph >> 1) * p.Wout) + (it.ph & 1)) * p.ldo * 4) : 0u;
    // byte offset of accumulator row r = lane part (one register) + a wave-uniform part in the instructions' scalar offset
    const unsigned lane_off = (lhi ? bit_off(2) : 0u) + (unsigned)((wn * 32 + l31) * 4);
    const unsigned wave_off = ((wm & 1) ? bit_off(5) : 0u) + ((wm & 2) ? bit_off(6) : 0u) + (unsigned)(it.kb * 64 * 4) + ph_off;
    const int lane_img = (wm * 32 + 4 * lhi) >> p.sh_thw;
    const int img_lim = p.N - it.n_first - lane_img;
    const float bj = (MODE == MODE_FWD && p.bias) ? p.bias[cout] : 0.f;
#ifdef W22_ABL_NO_EPI
    if (p.N < 0)
#endif
#pragma unroll
    for (int h = 0; h < 4; ++h) {       // four quarters of 4 accumulator rows: the second operand of 4 rows in flight at a time
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
          if constexpr (DG) {
            v[e] *= (rv[q][e] > 0.f) ? ga : gb;
          } else {
            v[e] += bj;
            v[e] = __builtin_fmaf(v[e], (v[e] > 0.f) ? g1 : g0, rv[q][e]);
          }
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

template <int MODE, int NRAW>
__global__ __launch_bounds__(512, 2) void wino22_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) body<MODE, 0, NRAW>(p, smem); else body<MODE, 1, NRAW>(p, smem);
}

// U_ph = G g_ph G^T (G = [[1,0],[1,1],[0,1]]) per phase from the packed 4x4 weight Wp[(kh * 4 + kw) * C + c][ldw]:
//   FWD:   input phase (p, q):  g[a][b] = w[kh(p,a)][kw(q,b)],  kh(1,a) = 2a, kh(0,a) = 1 + 2a;   input channels c, output channels k
//   DGRAD: output phase (ph, pw): g[al][be] = w[kh'(ph,al)][kw'(pw,be)], kh'(0,al) = 3 - 2 al, kh'(1,al) = 2 - 2 al;   input channels k, output c
// U[phase][xi][cin / 8][(cin % 8) / 4][cout][cin % 4].  One thread = one phase x four input channels x one output channel.
template <int MODE>
__global__ __launch_bounds__(256) void wino22_filter_kernel(const float* __restrict__ wp, float* __restrict__ U, int C, int K, int ldw) {
  const int cin = (MODE == MODE_FWD) ? C : K, cout = (MODE == MODE_FWD) ? K : C;
  const int per = (cin >> 2) * cout;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * per) return;
  const int ph = idx / per, rem = idx - ph * per;
  const int o = rem % cout, q4 = rem / cout;
  const int phh = ph >> 1, phw = ph & 1;
  float g[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kh = (MODE == MODE_FWD) ? (phh ? 2 * a : 1 + 2 * a) : (phh ? 2 - 2 * a : 3 - 2 * a);
      const int kw = (MODE == MODE_FWD) ? (phw ? 2 * b : 1 + 2 * b) : (phw ? 2 - 2 * b : 3 - 2 * b);
      if constexpr (MODE == MODE_FWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[a][b][j] = wp[(size_t)((kh * 4 + kw) * C + 4 * q4 + j) * ldw + o];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)((kh * 4 + kw) * C + o) * ldw + 4 * q4);
        g[a][b][0] = v.x; g[a][b][1] = v.y; g[a][b][2] = v.z; g[a][b][3] = v.w;
      }
    }
  const int nch = cin >> 3;
  float4* Uo = reinterpret_cast<float4*>(U) + (size_t)ph * 9 * nch * 2 * cout;
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
      Uo[((size_t)((i * 3 + j) * nch + (q4 >> 1)) * 2 + (q4 & 1)) * cout + o] = make_float4(u[0], u[1], u[2], u[3]);
    }
}


// =====================================================================================================================
// Weight gradient of the 4x4 stride-2 layers.  Per input phase (p, q) the 2x2 block of taps {kh(p,a), kw(q,b)} is the filter
// gradient of a 2x2 stride-1 correlation, i.e. again F(2x2, 2x2) with gy's 2x2 tile in the filter's place:
//
//   dW_pq = A^T [ sum_tiles (G gy_t G^T) (.) (B^T d_pq,t B) ] A      9 GEMMs  S_xi[(pq, c)][k] = sum_t V_xi[t][(pq, c)] Gy_xi[t][k]
//
// Rows of the GEMM = (phase, input channel), 4 C of them; a block = 128 rows (C >= 128: one phase x 128 channels; C = 64: two
// phases x 64 channels) x 64 output channels x all 9 xi x one split of the tile axis, chunks of 8 tiles; eight waves, each a
// 32 x 32 sub-block with nine accumulator tiles.  LDS rows are contraction-major ([xi][tile][row], fragments ds_read_b32
// pairs 2 tiles apart) as in wino_wgrad_kernel; waves 0-3 transform x (raw boxes -> B^T d B, a 3x3 tile x 4 channels per
// thread), waves 4-7 move x and gy (gy through G gy G^T on the way; the (1,1) plane is the tile's pixel sum = the bias
// gradient, summed in the blocks of the first row block).  The block's 2x2 taps per phase go into its slab of the workspace
// in the packed-weight layout [(kh * 4 + kw) * C + c][K]; wgrad_reduce_kernel sums the slabs.
constexpr int WV_PL = 8 * 128, WG_PL = 8 * 64;        // dwords per plane: V 8 tiles x 128 rows, Gy 8 tiles x 64 cols
constexpr int WV_SZ = 9 * WV_PL, WG_SZ = 9 * WG_PL;
constexpr int W_STAGE = WV_SZ + WG_SZ;                // 13 824 dwords
constexpr int W_RAWPX = 50;                           // raw box pixels per chunk (5 x 9, or 2 images x 5 x 5) ...
constexpr int W_RAWSZ = W_RAWPX * 128;                // ... x 128 rows' channels
constexpr int W_RAW0 = 2 * W_STAGE;
constexpr int W_LDS_DWORDS = 2 * W_STAGE + 2 * W_RAWSZ;   // 161 792 B
constexpr int W_NRAW = 7;                             // 50 px x 32 quads / 256

struct WArgs {
  const float* x;      // [N][H][W][ldx]     H = 2 GH
  const float* gy;     // [N][GH][GW][ldy]
  float* ws;           // [splits][16 * C][K]
  float* bias_ws;      // [splits][K] or NULL
  int N, H, W, C, K, ldx, ldy, GH, GW;
  int CTH, CTW, sh_ctw, sh_cthw, CNIMG;   // tiles per image part in a chunk, images per chunk (CTH * CTW * CNIMG = 8)
  int QH, QW, Q, qps;  // chunks per image along h / w, in all, per split
  int RBN, KB;         // 128-row blocks of the 4 C phase-channel rows, 64-wide blocks of K
  int CPB, sh_cpb;     // channels per box (min(C, 128)) and log2;  boxes per row block = 128 / CPB (1 or 2 phases)
};

template <int ROLE>
__device__ __forceinline__ void wbody(const WArgs& p, float* smem) {
  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63, w8 = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wr = w8 & 3, wc = w8 >> 2;            // sub-block: rows wr * 32 .., cols wc * 32 ..
  const int per = p.RBN * p.KB;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / per, rem = lin - split * per;
  const int rb = rem / p.KB, kb = rem - rb * p.KB;
  const int q_begin = split * p.qps, q_end = min(p.Q, q_begin + p.qps);
  const int T = max(0, q_end - q_begin);
  const int qpi = p.QH * p.QW;
  const int BH = 2 * p.CTH + 1, BW = 2 * p.CTW + 1, BHW = BH * BW;
  const int NBOX = 128 >> p.sh_cpb;                // phases per row block
  const int row0 = rb * 128;                       // first phase-channel row of the block; phase = row / C, channel = row % C

  struct Stream { int q, ng, cy, cx; };
  auto stream_at = [&](int q) -> Stream {
    Stream s; s.q = q; s.ng = q / qpi;
    const int r = q - s.ng * qpi;
    s.cy = r / p.QW; s.cx = r - s.cy * p.QW;
    return s;
  };
  auto step = [&](Stream& s) { ++s.q; if (++s.cx == p.QW) { s.cx = 0; if (++s.cy == p.QH) { s.cy = 0; ++s.ng; } } };

  // ---- movers, raw x: items (box, image, pixel, channel quad) ----
  int xfix[W_NRAW], xpk[W_NRAW];       // xpk: image << 24 | x row offset + 64 << 12 | x col offset + 64 (relative to the chunk's first x pixel)
  const int qpb = p.CPB >> 2;          // channel quads per box
  const int nitems = NBOX * p.CNIMG * BHW * qpb;
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int i = 0; i < W_NRAW; ++i) {
      const int item = tid + 256 * i;
      const int cq = item % qpb, t1 = item / qpb;
      const int px = t1 % BHW, t2 = t1 / BHW;
      const int im = t2 % p.CNIMG, box = t2 / p.CNIMG;
      const int r = px / BW, c = px - r * BW;
      const int row = row0 + box * p.CPB;
      const int ph = row / p.C, c0 = row - ph * p.C;
      const int pp = ph >> 1, pq = ph & 1;
      // box row r of phase p <-> x row 2 (row0_grid + r + s_p) + p, s_1 = -1, s_0 = 0: relative to the chunk's first grid row: 2 r - 1 | 2 r
      const int dr = 2 * r - pp, dc = 2 * c - pq;
      xpk[i] = (item < nitems) ? (im << 24 | (dr + 64) << 12 | (dc + 64)) : (127 << 24);
      xfix[i] = ((im * p.H + dr) * p.W + dc) * p.ldx * 4 + (c0 + cq * 4) * 4;
    }
  }
  // gy: (channel quad, tile) on the first 128 threads of the TRANSFORM waves (the movers' registers hold the x pieces in flight:
  // with gy's four on top they spilled inside the loop)
  const int gq = tid & 15, gt = (tid >> 4) & 7;
  const bool g_on = tid < 128;
  int gfix = 0, gimg = 0;
  if constexpr (ROLE == 0) {
    gimg = gt >> p.sh_cthw;
    const int ty = (gt >> p.sh_ctw) & (p.CTH - 1), tx = gt & (p.CTW - 1);
    gfix = ((gimg * p.GH + 2 * ty) * p.GW + 2 * tx) * p.ldy * 4 + (kb * 64 + gq * 4) * 4;
  }
  Stream sx = stream_at(q_begin), sg = stream_at(q_begin);
  float4 rraw[W_NRAW], rg[4];
  float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_bias = p.bias_ws != nullptr && rb == 0;

  auto load_raw = [&]() {
    const bool on = sx.q < q_end;
    const float* base = p.x + (size_t)sx.ng * p.CNIMG * p.H * p.W * p.ldx;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, on);
    const int oy = sx.cy * 4 * p.CTH, ox = sx.cx * 4 * p.CTW;          // the chunk's first x pixel (2 x 2 grid points per tile, 2 x pixels each)
    const int soff = (oy * p.W + ox) * p.ldx * 4;
    const int nleft = p.N - sx.ng * p.CNIMG;
#pragma unroll
    for (int i = 0; i < W_NRAW; ++i) {
      const int dr = ((xpk[i] >> 12) & 4095) - 64, dc = (xpk[i] & 4095) - 64;
      const bool ok = (unsigned)(oy + dr) < (unsigned)p.H && (unsigned)(ox + dc) < (unsigned)p.W && (xpk[i] >> 24) < nleft;
      rraw[i] = bload4(rs, ok ? (unsigned)(xfix[i] + soff) : OOB, 0);
    }
    step(sx);
  };
  auto store_raw = [&](int stage) {
#pragma unroll
    for (int i = 0; i < W_NRAW; ++i)
      if (tid + 256 * i < nitems) *reinterpret_cast<float4*>(smem + W_RAW0 + stage * W_RAWSZ + (tid + 256 * i) * 4) = rraw[i];
  };
  auto load_gy = [&]() {
    const bool on = g_on && sg.q < q_end && sg.ng * p.CNIMG + gimg < p.N;
    const float* base = p.gy + (size_t)sg.ng * p.CNIMG * p.GH * p.GW * p.ldy;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, sg.q < q_end);
    const unsigned v = on ? (unsigned)(gfix + ((sg.cy * 2 * p.CTH * p.GW + sg.cx * 2 * p.CTW) * p.ldy) * 4) : OOB;
    const unsigned dc = (unsigned)p.ldy * 4u, dr = (unsigned)(p.GW * p.ldy) * 4u;
    rg[0] = bload4(rs, v, 0); rg[1] = bload4(rs, v + dc, 0);
    rg[2] = bload4(rs, v + dr, 0); rg[3] = bload4(rs, v + dr + dc, 0);
    step(sg);
  };
  auto add4 = [](const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  const int wrG = WV_SZ + gt * 64 + gq * 4;
  auto gy_stage_store = [&](int bufoff) {      // G gy G^T, G = [[1,0],[1,1],[0,1]]: 9 planes x float4
    if (!g_on) return;
    const float4 m0 = add4(rg[0], rg[2]), m1 = add4(rg[1], rg[3]);      // middle row: g0. + g1.
    float* dst = smem + bufoff + wrG;
    *reinterpret_cast<float4*>(dst + 0 * WG_PL) = rg[0];
    *reinterpret_cast<float4*>(dst + 1 * WG_PL) = add4(rg[0], rg[1]);
    *reinterpret_cast<float4*>(dst + 2 * WG_PL) = rg[1];
    *reinterpret_cast<float4*>(dst + 3 * WG_PL) = m0;
    const float4 mm = add4(m0, m1);
    *reinterpret_cast<float4*>(dst + 4 * WG_PL) = mm;
    *reinterpret_cast<float4*>(dst + 5 * WG_PL) = m1;
    *reinterpret_cast<float4*>(dst + 6 * WG_PL) = rg[2];
    *reinterpret_cast<float4*>(dst + 7 * WG_PL) = add4(rg[2], rg[3]);
    *reinterpret_cast<float4*>(dst + 8 * WG_PL) = rg[3];
    if (do_bias) {
      asm volatile("" ::: "memory");
      colacc = add4(colacc, mm);
    }
  };

  // ---- transform waves: (row quad, tile) ----
  const int cq = tid & 31, tile = tid >> 5;
  int rd0 = 0;
  if constexpr (ROLE == 0) {
    const int img = tile >> p.sh_cthw, ty = (tile >> p.sh_ctw) & (p.CTH - 1), tx = tile & (p.CTW - 1);
    const int box = (cq * 4) >> p.sh_cpb, cql = cq - box * qpb;
    rd0 = (((box * p.CNIMG + img) * BHW + 2 * ty * BW + 2 * tx) * qpb + cql) * 4;
  }
  const int pxstep = qpb * 4, rowstep = BW * qpb * 4;
  const int wrV = tile * 128 + cq * 4;
  float4 d[3][3];
  auto read_raw = [&](int stage) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) d[r][c] = *reinterpret_cast<const float4*>(smem + W_RAW0 + stage * W_RAWSZ + rd0 + r * rowstep + c * pxstep);
  };
  auto sub4 = [](const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  auto row_ops = [&]() {
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[0][c] = sub4(d[0][c], d[1][c]); d[2][c] = sub4(d[2][c], d[1][c]); }
  };
  auto col_ops_store = [&](int bufoff, int r) {
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 0) * WV_PL + wrV) = sub4(d[r][0], d[r][1]);
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 1) * WV_PL + wrV) = d[r][1];
    *reinterpret_cast<float4*>(smem + bufoff + (r * 3 + 2) * WV_PL + wrV) = sub4(d[r][2], d[r][1]);
  };

  const int rdA = lhi * 128 + wr * 32 + l31;
  const int rdB = WV_SZ + lhi * 64 + wc * 32 + l31;

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- prologue ----
  if constexpr (ROLE == 1) {
    load_raw();            // raw 0
    store_raw(0);
    load_raw();            // raw 1
  } else {
    load_gy();             // gy 0
    gy_stage_store(0);
    load_gy();             // gy 1
  }
  __syncthreads();
  if constexpr (ROLE == 0) {
    read_raw(0);
    row_ops();
#pragma unroll
    for (int r = 0; r < 3; ++r) col_ops_store(0, r);
  } else {
    store_raw(1);
    load_raw();            // raw 2
  }
  __syncthreads();

  auto chunk = [&](auto par) {
    constexpr int P = decltype(par)::value;
    constexpr int cur = P * W_STAGE, nxt = W_STAGE - cur;
    float fa[2][4], fb[2][4];
    auto frags = [&](int xi, int s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fa[s][j] = smem[cur + rdA + xi * WV_PL + j * 256];
        fb[s][j] = smem[cur + rdB + xi * WG_PL + j * 128];
      }
    };
    frags(0, 0);
#pragma unroll
    for (int xi = 0; xi < 9; ++xi) {
      if (xi + 1 < 9) frags(xi + 1, (xi + 1) & 1);
      if constexpr (ROLE == 0) {
#ifndef W22_ABL_NO_TR
        if (xi == 0) read_raw(1 - P);
        if (xi == 2) row_ops();
        if (xi >= 3 && xi < 6) col_ops_store(nxt, xi - 3);
#endif
#ifndef W22_ABL_NO_U
        if (xi == 6) gy_stage_store(nxt);
        if (xi == 7) load_gy();
#endif
      } else {
#ifndef W22_ABL_NO_RAW
        if (xi == 0) store_raw(P);
        if (xi == 1) load_raw();
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

  // ---- taps: T[a][b] = (A^T S A)[a][b], A^T = [[1,1,0],[0,1,1]]; tap (kh(p,a), kw(q,b)), kh(1,a) = 2a, kh(0,a) = 1 + 2a ----
  {
    float* slab = p.ws + (size_t)split * 16 * p.C * p.K;
    const int k = kb * 64 + wc * 32 + l31;
    const int rowg = row0 + wr * 32;              // this wave's 32 rows lie in one phase (C % 32 == 0)
    const int ph = rowg / p.C, cbase = rowg - ph * p.C;
    const int pp = ph >> 1, pq = ph & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s[3], s2[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float m0 = acc[a * 3 + 0][r], m1 = acc[a * 3 + 1][r], m2 = acc[a * 3 + 2][r];
        s[a] = m0 + m1; s2[a] = m1 + m2;
      }
      const float t00 = s[0] + s[1], t01 = s2[0] + s2[1], t10 = s[1] + s[2], t11 = s2[1] + s2[2];
      const int c = cbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int kh0 = pp ? 0 : 1, kw0 = pq ? 0 : 1;        // a = 0 / b = 0; a = 1: + 2
      slab[((size_t)((kh0 * 4 + kw0) * p.C + c)) * p.K + k] = t00;
      slab[((size_t)((kh0 * 4 + kw0 + 2) * p.C + c)) * p.K + k] = t01;
      slab[((size_t)(((kh0 + 2) * 4 + kw0) * p.C + c)) * p.K + k] = t10;
      slab[((size_t)(((kh0 + 2) * 4 + kw0 + 2) * p.C + c)) * p.K + k] = t11;
    }
  }
  if (do_bias) {      // uniform: sum the eight tiles' partial column sums
    __syncthreads();
    float* red = smem;
    if (ROLE == 0 && g_on) *reinterpret_cast<float4*>(red + gt * 64 + gq * 4) = colacc;
    __syncthreads();
    if (ROLE == 1 && tid < 64) {
      float sum = 0.f;
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) sum += red[t8 * 64 + tid];
      p.bias_ws[(size_t)split * p.K + kb * 64 + tid] = sum;
    }
  }
}

__global__ __launch_bounds__(512, 2) void wino22_wgrad_kernel(const WArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 256) wbody<0>(p, smem); else wbody<1>(p, smem);
}

}  // namespace wino22
