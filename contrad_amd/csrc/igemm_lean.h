// Lean main loop of the implicit-GEMM engine (included by igemm.hip; shares IgemmArgs / the mode enum).
//
// Why a second kernel: on gfx950 every VALU / SALU instruction a wave issues between its MFMAs takes matrix-pipe
// time away from the SIMD (tools/micro/mfma_valu.hip: +16 plain VALU per 4-MFMA k-step = 155 -> 135 TF/s, 64-bit adds
// and v_cndmask cost double, SALU about half) -- the fp32 MFMA ceiling is reached only by a loop that is almost
// nothing but MFMA + memory instructions.  This kernel therefore moves ALL index arithmetic out of the loop:
//
//   * operands are fetched with raw BUFFER loads: voffset = a per-thread byte offset fixed in the prologue,
//     soffset = a wave-uniform (SGPR) per-tile offset, and "this element is padding / past the edge" is expressed
//     as voffset = 0x80000000, which the hardware range check turns into a zero fill -- no predicate masks,
//     no 64-bit address arithmetic, no selects on the data;
//   * shapes are restricted so that one K-tile (16 deep) never straddles a filter tap (FWD: Cin % 16 == 0,
//     DGRAD: Cout % 16 == 0) or, for WGRAD, is a fixed 16-position patch of the output grid: the tap / patch walk
//     is scalar code, the per-thread part collapses to "base | sign-extended invalid bit";
//   * K-contiguous operands (im2col rows of x, rows of gy, the packed weight read along cout) are stored in LDS as
//     [k/4][row][4] "quads": one ds_write_b128 per global float4 (no transpose) and one ds_read_b128 per lane per
//     four k-steps.  The MFMA consumes k in the permuted order k = 8h + 4*(lane>>5) + j (h = half of the tile,
//     j = k-step in the half) -- a contraction is order-free as long as A and B agree;
//   * row-contiguous operands (packed weight rows, WGRAD's position-major x / gy) are stored [k][cols] unpadded with
//     the column XOR-swizzled by bit 2 of k (conflict-free ds_read_b32 for both lane halves, leading dimension a
//     multiple of 64 dwords so the compiler can pair reads as ds_read2st64_b32 with immediate offsets);
//   * the LDS double buffer is selected by adding a scalar to five per-thread base registers per tile.
//
// Anything that does not fit (odd channel counts, non-power-of-two WGRAD grids, > 32 taps) runs on the general
// kernel in igemm.hip.  The two kernels produce bit-identical sums only for equal k order, which they do not share:
// parity tests compare each against the fp32 oracle with the stated tolerance.
#pragma once

#ifndef LEAN_DEEP
#define LEAN_DEEP 0      // dev knob: 1 = the narrow WGRAD instances (<= 128 x 64) keep TWO K-tiles of global loads in flight
#endif                   //           3 = also the narrow FWD / DGRAD instances (see lean_tile, "DEEP")
#ifndef LEAN_LD_NT
#define LEAN_LD_NT 0     // dev knob: the epilogue's second operand (act_ref / addend, read once) loaded non-temporally too
#endif
#ifndef LEAN_ST_NT
#define LEAN_ST_NT -1    // dev knob: 0 / 1 forces plain / non-temporal activation stores in every launch (-1: by output size)
#endif
#ifndef LEAN_QPAD
#define LEAN_QPAD 8      // dwords between the four k-quad planes of a quad-layout LDS tile beyond 4 * rows (see lean_tile)
#endif

constexpr unsigned LEAN_OOB = 0x80000000u;       // voffset of an element that must read as zero: out of range whether or
                                                 // not the hardware adds soffset (< 2^31) before the range check
constexpr unsigned LEAN_RANGE = 0x80000000u;     // num_records of every descriptor (offsets are block-relative)
constexpr unsigned WB_TOP = 2u, WB_BOT = 4u, WB_LEFT = 8u, WB_RIGHT = 16u;   // WGRAD border flags (inv / edge / wskip)

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  // (the builtin returns a GCC vector_size(16) type; bit_cast it, an implicit conversion to an ext_vector splats .x)
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lean_rsrc(const float* base, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, on ? (int)LEAN_RANGE : 0, 0x00020000);
}

// compile-time value of the K loop's buffer parity (std::integral_constant when the loop is unrolled by two, else 0)
template <class P> struct lean_par_value { static constexpr int v = 0; };
template <int V> struct lean_par_value<std::integral_constant<int, V>> { static constexpr int v = V; };

// x = q * d + r with a wave-uniform divisor: power-of-two extents (every level of the image pyramids here except the
// blur-padded 2^k + 1 ones) take a shift and a mask instead of the ~35-instruction runtime division -- the prologue is
// a quarter of all instructions a block of a shallow-K layer (K = 9 * 32) executes, and each of them costs matrix-pipe time
__device__ __forceinline__ int pow2_shift(int d) { return (d & (d - 1)) == 0 ? __builtin_ctz(d) : -1; }
__device__ __forceinline__ void divmod_u(int x, int d, int sh, int& q, int& r) {
  if (sh >= 0) { q = x >> sh; r = x & (d - 1); }
  else { q = x / d; r = x - q * d; }
}

// One tile of the launch: everything after the block -> (tile_m, tile_n, class / split) decode.  Its early returns are
// block-uniform and sit before the first barrier.  (A separate function so that a block can walk SEVERAL tiles: the strided
// data gradient hands its light parity classes out in runs of 2 / 4 M-tiles per block, see the decode in the kernel.)
template <int MODE, int BM, int BN, bool BAL>
__device__ __forceinline__ void lean_tile(const IgemmArgs& p, float* smem, const int tile_m, const int tile_n, const int by,
                                          const int wc_h0, const int wc_w0, const int wc_hc, const int wc_wc) {
  static_assert(BK == 16, "the lean loop is written for a 16-deep K-tile");
  static_assert(!BAL || MODE == MODE_DGRAD, "balanced order: strided data gradient only");
  // BAL (balanced strided DGRAD): image-major tiles of the parity classes only -- no pixel-major tiles, border classes or
  // split-K; folding the three plan fields to constants keeps that instance's scalar registers for its tile loop
  const int pl_pixmajor = BAL ? 0 : p.pixmajor, pl_nwin = BAL ? 0 : p.nwin, pl_dsplits = BAL ? 0 : p.dsplits;
  constexpr bool A_Q = (MODE != MODE_WGRAD);   // A K-contiguous in memory -> quad layout
  constexpr bool B_Q = (MODE == MODE_DGRAD);
  // quad stride = 4 * rows + LEAN_QPAD dwords.  ds_write_b128 is serviced in 8-lane groups and banks at (dword address)
  // mod 32 (MI355X_MICROARCH.md, LDS): the 8 lanes of a group are the 4 k-quads of two consecutive rows, so the four
  // planes must start 8 banks apart -- pad = 8 (mod 32).  (Rounds 1 - 4 used + 16, chosen for 64 banks: planes 0 / 2 and
  // 1 / 3 then collide, every quad store took two passes -- the "LDS bank-conflict fraction 0.333" of every DGRAD
  // instance and 0.11 - 0.17 of FWD in profiles/r0[2-4]_*_pmc.json.  Round 5, `tools/pmc_lds.sh base qpad8`, 3x3 256 -> 256
  // at 1 536 images: SQ_LDS_BANK_CONFLICT 24.0 M -> 0 (DGRAD), 12.0 M -> 0 (FWD), SQ_LDS_IDX_ACTIVE 72 M -> 48 M / 60 M;
  // the kernels' durations and every step time are unchanged -- the LDS array was never the limiter)
  // Reads stay conflict-free for any 16-byte-aligned stride: a ds_read_b128 lane group holds 16 consecutive rows of ONE plane.
  constexpr int QSA = BM * 4 + LEAN_QPAD, QSB = BN * 4 + LEAN_QPAD;
  constexpr int LDB = BN < 64 ? 64 : BN;                // row layout: >= 64 columns so the XOR-32 swizzle stays inside a row
  constexpr int A_SZ = A_Q ? 4 * QSA : 16 * BM;
  constexpr int B_SZ = B_Q ? 4 * QSB : 16 * LDB;
  constexpr int BUF = A_SZ + B_SZ;
  constexpr int PA = BM / 64, PB = BN < 64 ? 1 : BN / 64;   // float4 prefetch registers per operand
  constexpr int A_RPP = 1024 / BM, B_RPP = 1024 / BN;   // k-rows per pass of the row-layout loader
  // BN = 32 (StyleGAN2's 32-channel layers at 512x512): the B tile is 128 float4, so only threads with b_r < 16
  // (row layout) / qrow < 32 (quad layout) own a piece of it; the 4 waves stack 4 x 1 over the rows.
  constexpr bool B_HALF = BN < 64;
  constexpr int WAVES_N = BN < 64 ? 1 : 2, WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave arrangement");

  const contrad_conv_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // loader coordinates: quad loader = (row qrow + 64 i, k-quad kq); row loader = (k-row r + RPP i, column quad c4)
  const int kq = tid & 3, qrow = tid >> 2;
  const int a_c4 = tid % (BM / 4), a_r = tid / (BM / 4);
  const int b_c4 = tid % (BN / 4), b_r = tid / (BN / 4);

  int M = p.M, Ncol = p.Ncol;
  int T = 0;                       // K-tiles this block contracts over
  // pixel-major tiles (pl_pixmajor, small maps): the rows of an M-tile are BM IMAGES at ONE output pixel, so "this tap is
  // padding" is the same for every row and the K walk simply skips such taps (a 3x3 pad-1 layer on a 4x4 map multiplies
  // zeros in 31 % of its tap-positions, a 4x4 stride-2 layer onto 4x4 in 23 %)
  unsigned tapmask = 0xFFFFFFFFu;  // taps the walk visits (wave-uniform)
  unsigned wskip = 0;              // WGRAD pixel-major: border bits at which this tile's K-tiles are skipped
  long long px_off = 0;            // pixel-major: element offset of row 0 of the tile in the output, its row pitch, rows
  int px_pitch = 0, px_rows = 0;
  const float* baseA = p.A;        // descriptor bases (block-relative, so byte offsets stay far below 2^31)
  const float* baseB = p.B;
  unsigned va[PA], vb[PB];         // per-thread byte offsets (fixed)
  unsigned inv[PA];                // FWD / DGRAD: bit t set = tap t of this row is padding
                                   // WGRAD: bit 0/1/2/3 set = padding when the patch is in the top / bottom row,
                                   //        left / right column of the output grid (interior patches never pad)

  // wave-uniform walk state (of the NEXT tile to be loaded)
  int u_tap = 0, u_a = 0, u_b = 0, u_c0 = 0;   // FWD: tap, kh, kw, c0   DGRAD: ti, th, tw, co0
  int u_n = 0, u_h = 0, u_w = 0;               // WGRAD: patch origin (image, ho, wo)
  int ph = 0, pw = 0, Hc = 0, Wc = 0, kh0 = 0, kw0 = 0, nth = 0, ntw = 1, bh = 0, bw = 0;   // DGRAD class
  int gw = 1, gh = 1, gn = 1, n_begin = 0;                                                    // WGRAD patch

  if constexpr (MODE == MODE_FWD) {
    // split-K (p.ny > 1, small-M GEMMs such as the merged head layer): this block contracts K-tiles
    // [t_begin, t_begin + T) and writes a raw partial slab; fwd_reduce_kernel sums the slabs and applies the epilogue
    const int t_total = p.Kg / BK, t_begin = by * p.ptiles_per_split;
    T = min(p.ptiles_per_split, t_total - t_begin);
    if (T < 0) T = 0;
    const int ntaps = d.KH * d.KW;
    u_c0 = (t_begin / ntaps) * BK;          // taps-inner order: tile q = (chunk q / ntaps, tap q % ntaps)
    u_tap = t_begin % ntaps;
    u_a = u_tap / d.KW;
    u_b = u_tap - u_a * d.KW;
    const int HoWo = d.Ho * d.Wo;
    if (pl_pixmajor) {
      // tile_m = image block * Ho*Wo + pixel: neighbouring blocks read the same BM images
      int ib, pix;
      if (p.px_full) {   // the whole launch's tile order from the host (slot-balanced per XCD, igemm.hip pixel_order_full())
        const int e = p.px_order[tile_m];
        ib = e / HoWo; pix = e - ib * HoWo;
      } else {
        ib = tile_m / HoWo; pix = p.px_order[tile_m - ib * HoWo];
      }
      const int ho = pix / d.Wo, wo = pix - ho * d.Wo;
      const int n_first = ib * BM;
      unsigned wmask = 0, valid = 0;
      for (int kw = 0; kw < d.KW; ++kw) wmask |= ((unsigned)(wo * d.stride - d.pad + kw) < (unsigned)d.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < d.KH; ++kh)
        if ((unsigned)(ho * d.stride - d.pad + kh) < (unsigned)d.H) valid |= wmask << (kh * d.KW);
      tapmask = valid ? valid : 0xFFFFFFFFu;   // (no tap at all: T = 0, the walk must still terminate)
      T = __builtin_popcount(valid) * (d.C / BK);
      u_tap = valid ? __builtin_ctz(valid) : 0;
      u_a = u_tap / d.KW;
      u_b = u_tap - u_a * d.KW;
      u_c0 = 0;
      baseA = p.A + ((long long)n_first * d.H * d.W + (long long)(ho * d.stride - d.pad) * d.W + (wo * d.stride - d.pad)) * d.ldx;
      const int img = d.H * d.W * d.ldx;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = qrow + 64 * i;
        va[i] = (unsigned)((r * img + kq * 4) * 4);
        inv[i] = (n_first + r < d.N) ? 0u : 0xFFFFFFFFu;
      }
      px_off = ((long long)n_first * HoWo + pix) * d.ldy;
      px_pitch = HoWo * d.ldy;
      px_rows = min(BM, d.N - n_first);
    }
    // rows of the tile = (image, pixel) over the whole output map, or over this tile's border class (window)
    const bool win = pl_nwin > 0;
    const int gHo = win ? wc_hc : d.Ho, gWo = win ? wc_wc : d.Wo;
    if (win) {
      M = d.N * gHo * gWo;
      unsigned wmask = 0, valid = 0;      // the class's taps: those of its first pixel
      for (int kw = 0; kw < d.KW; ++kw) wmask |= ((unsigned)(wc_w0 * d.stride - d.pad + kw) < (unsigned)d.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < d.KH; ++kh)
        if ((unsigned)(wc_h0 * d.stride - d.pad + kh) < (unsigned)d.H) valid |= wmask << (kh * d.KW);
      tapmask = valid ? valid : 0xFFFFFFFFu;
      T = __builtin_popcount(valid) * (d.C / BK);
      u_tap = valid ? __builtin_ctz(valid) : 0;
      u_a = u_tap / d.KW;
      u_b = u_tap - u_a * d.KW;
      u_c0 = 0;
      if (m0 >= M) return;   // (uniform per block, before any barrier)
    }
    const int sh_w = pow2_shift(gWo), sh_h = pow2_shift(gHo);
    const int n_first = (sh_w >= 0 && sh_h >= 0) ? (m0 >> (sh_w + sh_h)) : m0 / (gHo * gWo);
    if (!pl_pixmajor) baseA = p.A + ((long long)n_first * d.H * d.W - (d.pad * d.W + d.pad)) * d.ldx;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (pl_pixmajor) break;
      const int m = m0 + qrow + 64 * i;
      const bool ok = m < M;
      const int mm = ok ? m : m0;
      int wo, t, ho, n;
      divmod_u(mm, gWo, sh_w, t, wo);
      divmod_u(t, gHo, sh_h, n, ho);
      ho += wc_h0; wo += wc_w0;
      va[i] = (unsigned)(((((n - n_first) * d.H + ho * d.stride) * d.W + wo * d.stride) * d.ldx + kq * 4) * 4);
      // valid taps form a rectangle: a KW-bit column mask replicated into the valid kernel rows (KH + KW steps, not KH*KW)
      unsigned wmask = 0, valid = 0;
      for (int kw = 0; kw < d.KW; ++kw) wmask |= ((unsigned)(wo * d.stride - d.pad + kw) < (unsigned)d.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < d.KH; ++kh)
        if ((unsigned)(ho * d.stride - d.pad + kh) < (unsigned)d.H) valid |= wmask << (kh * d.KW);
      inv[i] = ok ? ~valid : 0xFFFFFFFFu;   // (bits past the last tap are never looked at)
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int col = n0 + b_c4 * 4;
      vb[i] = (col < Ncol && b_r + B_RPP * i < BK) ? (unsigned)((((b_r + B_RPP * i) * d.ldw) + col) * 4) : LEAN_OOB;
    }
  } else if constexpr (MODE == MODE_DGRAD) {
    const int s = d.stride;
    // stride 1 with split-K (pl_dsplits > 1): grid.y counts K-splits, there is a single parity class
    const int cls = (pl_dsplits > 1) ? 0 : by;
    ph = cls / s;
    pw = cls % s;
    Hc = (d.H - ph + s - 1) / s;
    Wc = (d.W - pw + s - 1) / s;
    kh0 = (ph + d.pad) % s;
    kw0 = (pw + d.pad) % s;
    nth = (kh0 < d.KH) ? (d.KH - kh0 + s - 1) / s : 0;
    ntw = (kw0 < d.KW) ? (d.KW - kw0 + s - 1) / s : 0;
    bh = (ph + d.pad - kh0) / s;
    bw = (pw + d.pad - kw0) / s;
    M = d.N * Hc * Wc;
    Ncol = d.C;
    T = nth * ntw * d.K / BK;
    if (pl_dsplits > 1) {   // this block contracts K-tiles [t_begin, t_begin + T) into its slab (taps-inner order)
      const int t_begin = by * p.ptiles_per_split, ntaps = nth * ntw;
      T = min(p.ptiles_per_split, T - t_begin);
      if (T < 0) T = 0;
      u_c0 = (t_begin / ntaps) * BK;
      u_tap = t_begin % ntaps;
      u_a = u_tap / ntw;
      u_b = u_tap - u_a * ntw;
    }
    if (ntw == 0) ntw = 1;
    if (pl_nwin > 0) {   // stride 1: this tile's border class of dx pixels (a window of the map), its taps are those of its first pixel
      Hc = wc_hc; Wc = wc_wc;
      M = d.N * Hc * Wc;
      const int ah0 = wc_h0 + bh, aw0 = wc_w0 + bw;
      unsigned wmask = 0, valid = 0;
      for (int tw = 0; tw < ntw; ++tw) wmask |= ((unsigned)(aw0 - tw) < (unsigned)d.Wo ? 1u : 0u) << tw;
      for (int th = 0; th < nth; ++th)
        if ((unsigned)(ah0 - th) < (unsigned)d.Ho) valid |= wmask << (th * ntw);
      tapmask = valid ? valid : 0xFFFFFFFFu;
      T = __builtin_popcount(valid) * (d.K / BK);
      u_tap = valid ? __builtin_ctz(valid) : 0;
      u_a = u_tap / ntw;
      u_b = u_tap - u_a * ntw;
      u_c0 = 0;
    }
    const int HcWc = Hc * Wc;
    if (pl_pixmajor) {
      // tile_m = image block * (pixels of the largest class) + pixel of this class: BM images at ONE dx pixel
      int ib, cpix;
      if (p.px_full) {
        const int e = p.px_order[tile_m];
        ib = e / p.px_pixels; cpix = e - ib * p.px_pixels;
      } else {
        ib = tile_m / p.px_pixels; cpix = p.px_order[tile_m - ib * p.px_pixels];
      }
      if (cpix >= HcWc) return;   // uniform per block, before any barrier
      int hq = cpix / Wc, wq = cpix - hq * Wc;
      if (p.px_full == 2) {       // strided layer, one table for all four parity classes: the classes are mirror images of
        if (ph) hq = Hc - 1 - hq; // class (0,0) (checked on the host), so the table's pixel is mirrored along the odd axes
        if (pw) wq = Wc - 1 - wq;
      }
      const int ah = hq + bh, aw = wq + bw;
      const int n_first = ib * BM;
      unsigned wmask = 0, valid = 0;
      for (int tw = 0; tw < ntw; ++tw) wmask |= ((unsigned)(aw - tw) < (unsigned)d.Wo ? 1u : 0u) << tw;
      for (int th = 0; th < nth; ++th)
        if ((unsigned)(ah - th) < (unsigned)d.Ho) valid |= wmask << (th * ntw);
      tapmask = valid ? valid : 0xFFFFFFFFu;
      T = __builtin_popcount(valid) * (d.K / BK);
      u_tap = valid ? __builtin_ctz(valid) : 0;
      u_a = u_tap / ntw;
      u_b = u_tap - u_a * ntw;
      u_c0 = 0;
      M = d.N;
      baseA = p.A + ((long long)n_first * d.Ho * d.Wo + (long long)(ah - (nth - 1)) * d.Wo + (aw - (ntw - 1))) * d.ldy;
      const int img = d.Ho * d.Wo * d.ldy;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int r = qrow + 64 * i;
        va[i] = (unsigned)((r * img + kq * 4) * 4);
        inv[i] = (n_first + r < d.N) ? 0u : 0xFFFFFFFFu;
      }
      px_off = ((long long)n_first * d.H * d.W + (long long)(hq * s + ph) * d.W + (wq * s + pw)) * d.ldx;
      px_pitch = d.H * d.W * d.ldx;
      px_rows = min(BM, d.N - n_first);
    }
    if (!pl_pixmajor && m0 >= M) return;  // uniform per block, before any barrier
    const int sh_w = pow2_shift(Wc), sh_h = pow2_shift(Hc);
    const int n_first = (sh_w >= 0 && sh_h >= 0) ? (m0 >> (sh_w + sh_h)) : m0 / HcWc;
    if (!pl_pixmajor) baseA = p.A + ((long long)n_first * d.Ho * d.Wo - ((nth - 1) * d.Wo + (ntw - 1))) * d.ldy;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (pl_pixmajor) break;
      const int m = m0 + qrow + 64 * i;
      const bool ok = m < M;
      const int mm = ok ? m : m0;
      int wq, t, n, hq;
      divmod_u(mm, Wc, sh_w, t, wq);
      divmod_u(t, Hc, sh_h, n, hq);
      hq += wc_h0; wq += wc_w0;               // (border class: position inside the window -> position in the map)
      const int ah = hq + bh, aw = wq + bw;   // ho = ah - th, wo = aw - tw
      va[i] = (unsigned)(((((n - n_first) * d.Ho + ah) * d.Wo + aw) * d.ldy + kq * 4) * 4);
      unsigned wmask = 0, valid = 0;
      for (int tw = 0; tw < ntw; ++tw) wmask |= ((unsigned)(aw - tw) < (unsigned)d.Wo ? 1u : 0u) << tw;
      for (int th = 0; th < nth; ++th)
        if ((unsigned)(ah - th) < (unsigned)d.Ho) valid |= wmask << (th * ntw);
      inv[i] = ok ? ~valid : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int c = n0 + qrow + 64 * i;
      vb[i] = (c < Ncol && qrow + 64 * i < BN) ? (unsigned)((c * d.ldw + kq * 4) * 4) : LEAN_OOB;
    }
  } else {
    const int tiles_total = p.P / BK;
    const int t_begin = by * p.ptiles_per_split;
    T = min(p.ptiles_per_split, tiles_total - t_begin);
    if (T < 0) T = 0;
    gw = min(d.Wo, 16);
    gh = min(d.Ho, 16 / gw);
    if (pl_pixmajor) { gw = 1; gh = 1; }   // pixel-major positions: a K-tile is 16 IMAGES at one output pixel
    gn = 16 / (gw * gh);
    const int tpr = d.Wo / gw, tpi = tpr * (d.Ho / gh);   // patches per output row / per image
    u_w = (t_begin % tpr) * gw;
    u_h = ((t_begin / tpr) % (d.Ho / gh)) * gh;
    u_n = (t_begin / tpi) * gn;
    n_begin = u_n;
    baseA = p.A + ((long long)n_begin * d.H * d.W - (d.pad * d.W + d.pad)) * d.ldx;
    baseB = pl_pixmajor ? p.B + (long long)n_begin * d.Ho * d.Wo * d.ldy : p.B + (long long)t_begin * BK * d.ldy;
    if (pl_pixmajor) {
      // Every row of this tile belongs to ONE filter tap (the plan guarantees C % BM == 0), so "the tap reads padding at
      // this pixel" holds for the whole K-tile: such K-tiles are not visited at all.  wskip = the borders (bits as in
      // inv / edge) at which this tile's tap is padding.  The blocks that also sum the bias gradient visit everything.
      const int tap0 = m0 / d.C;
      const int kh = tap0 / d.KW, kw = tap0 - kh * d.KW;
      const int chh = kh - d.pad, cww = kw - d.pad;
      const int hb = (d.Ho - 1) * d.stride, wb = (d.Wo - 1) * d.stride;
      wskip = ((unsigned)chh < (unsigned)d.H ? 0u : WB_TOP) | ((unsigned)(hb + chh) < (unsigned)d.H ? 0u : WB_BOT) |
              ((unsigned)cww < (unsigned)d.W ? 0u : WB_LEFT) | ((unsigned)(wb + cww) < (unsigned)d.W ? 0u : WB_RIGHT);
      if (p.bias_ws != nullptr && tile_m == 0) wskip = 0;
      int hh = u_h, ww = u_w, cnt = 0, first = -1;
      for (int q = 0; q < T; ++q) {            // scalar: <= a few hundred patches per split
        const unsigned e = (hh == 0 ? WB_TOP : 0u) | (hh == d.Ho - 1 ? WB_BOT : 0u) | (ww == 0 ? WB_LEFT : 0u) | (ww == d.Wo - 1 ? WB_RIGHT : 0u);
        if (!(wskip & e)) { ++cnt; if (first < 0) first = q; }
        if (++ww == d.Wo) { ww = 0; if (++hh == d.Ho) hh = 0; }
      }
      T = cnt;
      if (cnt == 0) wskip = 0;                 // (nothing to visit: the walk must still terminate)
      for (int q = 0; q < first; ++q) {        // on to the first K-tile this block visits
        u_w += 1;
        if (u_w == d.Wo) { u_w = 0; u_h += 1; if (u_h == d.Ho) { u_h = 0; u_n += gn; } }
      }
    }
    const int ic = m0 + a_c4 * 4;
    const bool colok = ic < p.Kg;
    const int icc = colok ? ic : 0;
    const int tap = icc / d.C, c = icc - tap * d.C;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int r = a_r + A_RPP * i;
      const int dw = r % gw, dh = (r / gw) % gh, dn = r / (gw * gh);
      const int chh = dh * d.stride - d.pad + kh, cww = dw * d.stride - d.pad + kw;   // input coords at patch origin 0
      const int hb = (d.Ho - gh) * d.stride, wb = (d.Wo - gw) * d.stride;
      inv[i] = ((unsigned)chh < (unsigned)d.H ? 0u : WB_TOP) | ((unsigned)(hb + chh) < (unsigned)d.H ? 0u : WB_BOT) |
               ((unsigned)cww < (unsigned)d.W ? 0u : WB_LEFT) | ((unsigned)(wb + cww) < (unsigned)d.W ? 0u : WB_RIGHT);
      va[i] = colok ? (unsigned)((((dn * d.H + dh * d.stride + kh) * d.W + dw * d.stride + kw) * d.ldx + c) * 4)
                    : LEAN_OOB;     // a column past Kg never validates
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int col = n0 + b_c4 * 4;
      const int rowpitch = pl_pixmajor ? d.Ho * d.Wo * d.ldy : d.ldy;      // pixel-major: the 16 rows are 16 images
      vb[i] = (col < Ncol && b_r + B_RPP * i < BK) ? (unsigned)((((b_r + B_RPP * i) * rowpitch) + col) * 4) : LEAN_OOB;
    }
  }

  // ---------------- per-tile (wave-uniform) state and loaders ----------------
  // DEEP: two register sets -- the loads of tile t + 2 are issued while tile t is multiplied and land in LDS during tile
  // t + 1.  One K-tile of a narrow instance is 1 - 2 us of matrix-pipe time for the waves of a SIMD together, the order of a
  // load that misses the L2 (SQ_WAIT_ANY 0.37 of the wave cycles of the 128 x 64 WGRAD instance vs 0.10 of 128 x 128).
  constexpr bool DEEP = ((LEAN_DEEP & 1) && MODE == MODE_WGRAD && BM * BN <= 128 * 64) ||
                        ((LEAN_DEEP & 2) && MODE != MODE_WGRAD && BM * BN <= 128 * 64);
  constexpr int NSET = DEEP ? 2 : 1;
  float4 ra[NSET][PA], rb[NSET][PB];
  unsigned soffA = 0, soffB = 0;
  unsigned edge = 0;   // WGRAD: which borders of the output grid the next patch touches (bits as in inv)
  // ... and the same word for the walk's current position.  The four flags are 2 / 4 / 8 / 16, not 1 / 2 / 4 / 8: the
  // compiler turns "cond ? 1 : 0" of a scalar compare into a VECTOR select (v_cndmask + v_or per K-tile in every wave),
  // any other constant stays an s_cselect_b32
  auto wgrad_pos_edge = [&]() -> unsigned {
    return (u_h == 0 ? WB_TOP : 0u) | (u_h == d.Ho - gh ? WB_BOT : 0u) | (u_w == 0 ? WB_LEFT : 0u) | (u_w == d.Wo - gw ? WB_RIGHT : 0u);
  };
  unsigned pos_edge = (MODE == MODE_WGRAD) ? wgrad_pos_edge() : 0u;
  // WGRAD: byte offsets of the walk position in x and gy, stepped with the walk (additions; as closed forms they were four
  // s_mul and a pixel-major branch per K-tile): one patch to the right / down a patch row (from past the last column) /
  // on to the next image group (from past the last row); gy positions are consecutive except across image groups of the
  // pixel-major walk (one pixel of 16 images -> the next pixel; after the last pixel 15 images further)
  // FWD / DGRAD: the same for the (tap, 16-channel chunk) walk -- one tap to the right / on to the next tap row (from past
  // the last tap of a row) / on to the next chunk (from past the last tap row)
  unsigned pos_offA = 0, pos_offB = 0, wk_aw = 0, wk_ah = 0, wk_an = 0, wk_b = 0, wk_bh = 0, wk_bn = 0;
  (void)wk_bh;
  if constexpr (MODE == MODE_FWD) {
    pos_offA = (unsigned)(((u_a * d.W + u_b) * d.ldx + u_c0) * 4);
    pos_offB = (unsigned)((u_tap * d.C + u_c0) * d.ldw * 4);
    wk_aw = (unsigned)(d.ldx * 4);
    wk_ah = (unsigned)((d.W - d.KW) * d.ldx * 4);
    wk_an = (unsigned)((BK - d.KH * d.W * d.ldx) * 4);
    wk_b = (unsigned)(d.C * d.ldw * 4);
    wk_bn = (unsigned)((BK - d.KH * d.KW * d.C) * d.ldw * 4);
  } else if constexpr (MODE == MODE_DGRAD) {
    pos_offA = (unsigned)((((nth - 1 - u_a) * d.Wo + (ntw - 1 - u_b)) * d.ldy + u_c0) * 4);
    pos_offB = (unsigned)((((kh0 + d.stride * u_a) * d.KW + (kw0 + d.stride * u_b)) * d.C * d.ldw + u_c0) * 4);
    wk_aw = (unsigned)(-d.ldy * 4);
    wk_ah = (unsigned)((ntw - d.Wo) * d.ldy * 4);
    wk_an = (unsigned)((nth * d.Wo * d.ldy + BK) * 4);
    wk_b = (unsigned)(d.stride * d.C * d.ldw * 4);
    wk_bh = (unsigned)(d.stride * (d.KW - ntw) * d.C * d.ldw * 4);
    wk_bn = (unsigned)((BK - d.stride * nth * d.KW * d.C * d.ldw) * 4);
  }
  if constexpr (MODE == MODE_WGRAD) {
    pos_offA = (unsigned)(((((u_n - n_begin) * d.H + u_h * d.stride) * d.W + u_w * d.stride) * d.ldx) * 4);
    pos_offB = pl_pixmajor ? (unsigned)((((u_n - n_begin) * d.Ho + u_h) * d.Wo + u_w) * d.ldy * 4) : 0u;
    wk_aw = (unsigned)(gw * d.stride * d.ldx * 4);
    wk_ah = (unsigned)((gh * d.stride * d.W - d.Wo * d.stride) * d.ldx * 4);
    wk_an = (unsigned)((gn * d.H - d.Ho * d.stride) * d.W * d.ldx * 4);
    wk_b = (unsigned)((pl_pixmajor ? 1 : BK) * d.ldy * 4);
    wk_bn = pl_pixmajor ? (unsigned)((gn - 1) * d.Ho * d.Wo * d.ldy * 4) : 0u;
  }
  int t_next = 0;
  __amdgpu_buffer_rsrc_t rsA = lean_rsrc(baseA, false), rsB = lean_rsrc(baseB, false);

  // fix the scalar offsets / descriptors of tile t_next, then step the walk
  auto begin_tile = [&]() {
    const bool on = t_next < T;          // past the last tile: descriptors with 0 records -> every load is a zero fill
    rsA = lean_rsrc(baseA, on);
    rsB = lean_rsrc(baseB, on);
    soffA = pos_offA;   // (offsets -- and WGRAD's border word -- of the walk position: kept up to date by the walk, end_tile)
    soffB = pos_offB;
    if constexpr (MODE == MODE_WGRAD) edge = pos_edge;
  };
  auto end_tile = [&]() {   // advance the walk to the tile after t_next
    ++t_next;
    if constexpr (MODE == MODE_FWD) {
      // contraction order: all taps of one 16-channel chunk, then the next chunk.  The KH*KW taps of a chunk re-read the
      // same few input pixels (64 B each) in consecutive tiles -> L1/L2 hits; tap-major order swept the block's whole
      // input window (all channels) once per tap, 128 blocks per XCD x ~100 KB did not fit the 4 MB L2, and the
      // kernel fetched 5x its algorithmic bytes from HBM.
      do {   // (pixel-major tiles: on to the next tap that is not padding; otherwise every tap is visited)
        ++u_tap; ++u_b; pos_offA += wk_aw; pos_offB += wk_b;
        if (u_b == d.KW) {
          u_b = 0; ++u_a; pos_offA += wk_ah;
          if (u_a == d.KH) { u_a = 0; u_tap = 0; pos_offA += wk_an; pos_offB += wk_bn; }
        }
      } while (!((tapmask >> u_tap) & 1u));
    } else if constexpr (MODE == MODE_DGRAD) {
      do {
        ++u_tap; ++u_b; pos_offA += wk_aw; pos_offB += wk_b;
        if (u_b == ntw) {
          u_b = 0; ++u_a; pos_offA += wk_ah; pos_offB += wk_bh;
          if (u_a == nth) { u_a = 0; u_tap = 0; pos_offA += wk_an; pos_offB += wk_bn; }
        }
      } while (!((tapmask >> u_tap) & 1u));
    } else {
      do {   // (pixel-major: on to the next K-tile whose pixel is not padding for this tile's tap)
        u_w += gw; pos_offA += wk_aw; pos_offB += wk_b;
        if (u_w == d.Wo) {
          u_w = 0; u_h += gh; pos_offA += wk_ah;
          if (u_h == d.Ho) { u_h = 0; u_n += gn; pos_offA += wk_an; pos_offB += wk_bn; }
        }
        pos_edge = wgrad_pos_edge();
      } while (wskip & pos_edge);
    }
  };
  auto load_a_piece = [&](int set, int i) {
    unsigned v;
    if constexpr (MODE == MODE_WGRAD) {
      v = (inv[i] & edge) ? LEAN_OOB : va[i];
    } else {
      v = va[i] | ((inv[i] >> u_tap) << 31);
    }
    ra[set][i] = bload4(rsA, v, soffA);
  };
  auto load_b_piece = [&](int set, int i) { rb[set][i] = bload4(rsB, vb[i], soffB); };

  // ---------------- LDS addressing (dword indices relative to the current buffer) ----------------
  int wrA, wrB, rdA[TM], rdB[TN];
  if constexpr (A_Q) {
    wrA = kq * QSA + qrow * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) rdA[i] = lhi * QSA + (wm * WM + i * 32 + l31) * 4;
  } else {
    wrA = a_r * BM + ((a_c4 * 4) ^ (((a_r >> 2) & 1) * 32));
#pragma unroll
    for (int i = 0; i < TM; ++i) rdA[i] = 4 * lhi * BM + ((wm * WM + i * 32 + l31) ^ (32 * lhi));
  }
  if constexpr (B_Q) {
    wrB = A_SZ + kq * QSB + qrow * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) rdB[j] = A_SZ + lhi * QSB + (wn * WN + j * 32 + l31) * 4;
  } else {
    wrB = A_SZ + b_r * LDB + ((b_c4 * 4) ^ (((b_r >> 2) & 1) * 32));
#pragma unroll
    for (int j = 0; j < TN; ++j) rdB[j] = A_SZ + 4 * lhi * LDB + ((wn * WN + j * 32 + l31) ^ (32 * lhi));
  }

  // WGRAD bias gradient: B IS gy, so the first M-tile's blocks also accumulate its column sums
  float4 colacc = zero4();
  const bool do_bias = (MODE == MODE_WGRAD) && p.bias_ws != nullptr && tile_m == 0;   // uniform per block

  auto store_a_piece = [&](int bufoff, int set, int i) {
    float* dst = smem + bufoff + wrA + (A_Q ? i * 256 : i * A_RPP * BM);
    *reinterpret_cast<float4*>(dst) = ra[set][i];
  };
  const bool b_owner = !B_HALF || (B_Q ? qrow < BN : b_r < BK);   // does this thread own a piece of the B tile?
  auto store_b_piece = [&](int bufoff, int set, int i) {
    float* dst = smem + bufoff + wrB + (B_Q ? i * 256 : i * B_RPP * LDB);
    if (b_owner) *reinterpret_cast<float4*>(dst) = rb[set][i];
    if constexpr (MODE == MODE_WGRAD) {
      if (do_bias) {   // the empty asm keeps this a real uniform branch: if-converted it costs 8 VALU + 4 selects per
        asm volatile("" ::: "memory");   // tile in EVERY block, and only 1 block in tiles_m needs it
        colacc.x += rb[set][i].x; colacc.y += rb[set][i].y; colacc.z += rb[set][i].z; colacc.w += rb[set][i].w;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (T > 0) {
    begin_tile();
#pragma unroll
    for (int i = 0; i < PA; ++i) load_a_piece(0, i);
#pragma unroll
    for (int i = 0; i < PB; ++i) load_b_piece(0, i);
    end_tile();
    if constexpr (DEEP) {      // tile 1 goes into flight before tile 0 is waited for (set 1: stored during K-tile 0)
      begin_tile();
#pragma unroll
      for (int i = 0; i < PA; ++i) load_a_piece(1, i);
#pragma unroll
      for (int i = 0; i < PB; ++i) load_b_piece(1, i);
      end_tile();
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) store_a_piece(0, 0, i);
#pragma unroll
    for (int i = 0; i < PB; ++i) store_b_piece(0, 0, i);
  }
  __syncthreads();

  // does this wave hold any row of the matrix at all (wave-uniform)?  Only the narrow WGRAD instances test it: the
  // 128 x 128 tiles sit at the 128-VGPR limit of 4 waves per SIMD and extra control flow around the accumulators made
  // the register allocator spill (per-32-row-group tests: 500 ... 2200 spilled values, -Rpass-analysis=kernel-resource-usage)
  constexpr bool RAGGED_SKIP = (MODE == MODE_WGRAD) && BN <= 64;
  const bool wave_on = !RAGGED_SKIP || __builtin_amdgcn_readfirstlane((m0 + wm * WM < M) ? 1 : 0) != 0;

  // Main loop: 8 k-steps of TM x TN MFMAs per tile; the next tile's global loads ride between the first k-steps,
  // its LDS stores (other buffer) between the last ones; fragments of the whole tile are fetched at the top.
  constexpr int KS = BK / 2;
#ifndef IGEMM_ST_SHIFT
#define IGEMM_ST_SHIFT 1   // k-steps of MFMA left after the last LDS store of a tile (dev A/B knob: 0 -> 1 = +0.5 %)
#endif
  constexpr int A_LD0 = 0, B_LD0 = PA, A_ST0 = KS - PA - PB - IGEMM_ST_SHIFT, B_ST0 = KS - PB - IGEMM_ST_SHIFT;
  static_assert(A_ST0 >= A_LD0 + PA - 1 && B_ST0 >= B_LD0 + PB - 1 && A_ST0 >= 0,
                "a piece must be stored after its own load was issued");
  // One K-tile out of LDS buffer PAR (compile-time): the loop below is unrolled by two so that every LDS address of the
  // loop is "per-thread base register + immediate" -- selecting the buffer at run time cost 3 SALU + 6 v_add per K-tile in
  // every wave (round 4: each VALU instruction between MFMAs takes matrix-pipe time, §3).
  // (The 128 x 128 FWD / DGRAD instances sit at the 128-VGPR limit and spill 25 ... 215 registers in every unrolled shape
  // tried -- break in the middle, or an even tile count with a zero-filled last tile: they keep the run-time buffer select.)
  constexpr bool UNROLL2 = !(BM * BN > 128 * 64 && MODE != MODE_WGRAD);
  static_assert(!DEEP || UNROLL2, "the two-deep prefetch needs the buffer parity at compile time");
  auto k_tile = [&](auto par, auto with_mfma, const int t) {
    const int cur = (int)par * BUF, nxt = BUF - cur;   // (par: std::integral_constant when unrolled -> folds to immediates)
    (void)t;
    float fa[2][TM][4], fb[2][TN][4];
#if defined(LEAN_ABLATE_FRAGS)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[h][i][j] = (float)(t + i + j);
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[h][i][j] = (float)(t - i - j);
      }
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_Q) {
          const float4 q = *reinterpret_cast<const float4*>(smem + cur + rdA[i] + 2 * h * QSA);
          fa[h][i][0] = q.x; fa[h][i][1] = q.y; fa[h][i][2] = q.z; fa[h][i][3] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[h][i][j] = smem[cur + rdA[i] + (8 * h + j) * BM];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if constexpr (B_Q) {
          const float4 q = *reinterpret_cast<const float4*>(smem + cur + rdB[i] + 2 * h * QSB);
          fb[h][i][0] = q.x; fb[h][i][1] = q.y; fb[h][i][2] = q.z; fb[h][i][3] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[h][i][j] = smem[cur + rdB[i] + (8 * h + j) * LDB];
        }
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int h = ks >> 2, j = ks & 3;
#if !defined(LEAN_ABLATE_LOADS)     // (ablation builds give WRONG results: timing only, tools/build_variant.sh)
      // (DEEP: K-tile t loads tile t + 2 into set t & 1 -- its previous content, tile t, is in LDS -- and stores set
      // (t + 1) & 1; the K loop is unrolled by two for every DEEP instance, so `par` = t & 1 at compile time)
      constexpr int LSET = DEEP ? lean_par_value<decltype(par)>::v : 0, SSET = DEEP ? 1 - lean_par_value<decltype(par)>::v : 0;
      if (ks == A_LD0) begin_tile();
      if (ks >= A_LD0 && ks < A_LD0 + PA) load_a_piece(LSET, ks - A_LD0);
      if (ks >= B_LD0 && ks < B_LD0 + PB) load_b_piece(LSET, ks - B_LD0);
      if (ks == B_LD0 + PB - 1) end_tile();
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ragged last M-tile (WGRAD with Kg = 288 = 2.25 x 128: the 32-channel 3x3 layers): a wave whose rows all lie
      // past M holds zero fills only -- its MFMAs are skipped, the matrix pipe goes to the other blocks' waves on this SIMD
      if constexpr (decltype(with_mfma)::value) {   // (a wave of rows past M: selected once per block, below)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i][j], fb[h][jj][j], acc[i][jj], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#if !defined(LEAN_ABLATE_STORES)
      if (ks >= A_ST0 && ks < A_ST0 + PA) store_a_piece(nxt, SSET, ks - A_ST0);
      if (ks >= B_ST0 && ks < B_ST0 + PB) store_b_piece(nxt, SSET, ks - B_ST0);
#endif
    }
#if !defined(LEAN_ABLATE_BARRIER)
    __syncthreads();
#endif
  };
  auto k_loop = [&](auto with_mfma) {
    if constexpr (UNROLL2) {
      for (int t = 0; t < T; t += 2) {
        k_tile(std::integral_constant<int, 0>{}, with_mfma, t);
        if (t + 1 >= T) break;
        k_tile(std::integral_constant<int, 1>{}, with_mfma, t + 1);
      }
    } else {
      for (int t = 0; t < T; ++t) k_tile(t & 1, with_mfma, t);
    }
  };
  // RAGGED_SKIP: a wave whose rows all lie past M runs a copy of the loop without MFMAs (one branch per block; testing
  // wave_on at every k-step was 8 branches per K-tile in every wave of the narrow WGRAD instances)
  if constexpr (RAGGED_SKIP) {
    if (wave_on) k_loop(std::true_type{}); else k_loop(std::false_type{});
  } else {
    k_loop(std::true_type{});
  }

  // ---------------- epilogue ----------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // Stores go through a block-relative buffer descriptor too: voffset = (per-lane column part) + (row part), rows past
  // M and columns past Ncol fall outside num_records and are dropped by the hardware -- one v_add per store instead of
  // a 64-bit address, two compares and a branch (the 64 stores of a wave were ~2000 instructions; blocks of the
  // shallow-K StyleGAN2 layers spent more time here than in their 18 K-tiles).
  // activation stores of outputs no cache will hold until the consumer runs: non-temporal (aux = 2), igemm.hip st_nt_for()
  // (The 128 x 128 FWD / DGRAD instances -- the headline's kernels, at the 128-VGPR limit -- do not carry the second form:
  // with it they spill 52 - 56 bytes; the layers they serve in StyleGAN2_512 have outputs of 0.4 GB, the smallest that
  // would qualify.)
  constexpr bool NT_CAP = (MODE != MODE_WGRAD) && !(BM * BN > 128 * 64);
  const bool st_nt = NT_CAP && ((LEAN_ST_NT >= 0) ? (LEAN_ST_NT != 0) : (p.st_nt != 0));
  constexpr unsigned COL_OOB = 0x40000000u;    // > every valid block-relative offset, and 2 * COL_OOB does not wrap
  unsigned colpart[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = n0 + wn * WN + j * 32 + l31;
    colpart[j] = (c < Ncol) ? (unsigned)c * 4u : COL_OOB;
  }
  // epilogue rows whose second operand is in flight together (see below); the balanced 128 x 128 instance has room for 4
  constexpr int EPI_G = (BAL && BM * BN > 128 * 64) ? 4 : 8;
  const bool rowtab = (MODE == MODE_DGRAD) || (MODE == MODE_FWD && pl_nwin > 0);   // rows scattered over the output map
  if (rowtab) {
    // row -> byte offset of the output pixel (dx for DGRAD, y for a FWD border class) relative to the tile's first image
    // (or out of range), staged in LDS
    unsigned* rowoff = reinterpret_cast<unsigned*>(smem);
    const int rH = (MODE == MODE_DGRAD) ? Hc : wc_hc, rW = (MODE == MODE_DGRAD) ? Wc : wc_wc;   // extent the rows enumerate
    const int oH = (MODE == MODE_DGRAD) ? d.H : d.Ho, oW = (MODE == MODE_DGRAD) ? d.W : d.Wo;   // the output map
    const int old_ = (MODE == MODE_DGRAD) ? d.ldx : d.ldy;
    const int sh_w = pow2_shift(rW), sh_h = pow2_shift(rH);
    const int n_first = (sh_w >= 0 && sh_h >= 0) ? (m0 >> (sh_w + sh_h)) : m0 / (rH * rW);
    if (tid < BM) {
      const int m = m0 + tid;
      unsigned off = 2u * COL_OOB;
      if (MODE == MODE_DGRAD && pl_pixmajor) {
        if (tid < px_rows) off = (unsigned)(tid * px_pitch) * 4u;
      } else if (m < M) {
        int wq, t2, hq, n;
        divmod_u(m, rW, sh_w, t2, wq);
        divmod_u(t2, rH, sh_h, n, hq);
        hq += wc_h0; wq += wc_w0;
        if constexpr (MODE == MODE_DGRAD)
          off = (unsigned)((((n - n_first) * oH + hq * d.stride + ph) * oW + (wq * d.stride + pw)) * old_) * 4u;
        else
          off = (unsigned)((((n - n_first) * oH + hq) * oW + wq) * old_) * 4u;
      }
      rowoff[tid] = off;
    }
    __syncthreads();
    const size_t img0 = (MODE == MODE_DGRAD && pl_pixmajor)
                            ? (size_t)px_off
                            : (size_t)n_first * oH * oW * old_ +
                                  ((MODE == MODE_DGRAD && pl_dsplits > 1) ? (size_t)by * (size_t)p.slab_elems : 0);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + img0, 0, (int)COL_OOB, 0x00020000);
    // second operand of the epilogue, in the output's own layout: DGRAD the producer's activation (act'), FWD the addend
    const float* ref = (MODE == MODE_DGRAD) ? p.act_ref : p.addend;
    const __amdgpu_buffer_rsrc_t rsR =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ref ? ref + img0 : p.C + img0), 0,
                                          ref ? (int)COL_OOB : 0, 0x00020000);
    const float g1 = p.gain, g0 = p.gain * p.slope;
    float bj[TN];
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      const int c = n0 + wn * WN + jj * 32 + l31;
      bj[jj] = (MODE == MODE_FWD && p.bias && c < Ncol) ? p.bias[c] : 0.f;
    }
    // The second operand (act_ref / addend) is fetched for EPI_G rows at a time -- EPI_G x TN loads in flight -- BEFORE that
    // group's first store.  (Round 4: written as "load, use, store" per element the compiler kept that order -- it cannot
    // move a buffer load across a buffer store -- and waited vmcnt(0) 64 times per thread, i.e. paid the full memory latency
    // of a load AND of the preceding store per element: a block's epilogue lasted as long as ~30 K-tiles.  A whole 32-row
    // group in flight spills the 128 x 128 instances: 8 rows keep them inside 128 VGPRs.)
    // (The store loop exists twice, plain and non-temporal -- the cache-policy operand of a buffer store is an immediate
    // -- behind ONE launch-uniform branch: written as a test per store the compiler kept 243 branches in the epilogue.)
    auto store_rows = [&](auto nt_c) {
    constexpr int ST_AUX = decltype(nt_c)::value ? 2 : 0;
    constexpr int LD_AUX = (LEAN_LD_NT && decltype(nt_c)::value) ? 2 : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += EPI_G) {
        unsigned offs[EPI_G];
#pragma unroll
        for (int q = 0; q < EPI_G; ++q) {
          const int r = r0 + q;
          offs[q] = rowoff[wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
        }
        float rv[EPI_G][TN];
        if (ref) {   // uniform
#pragma unroll
          for (int q = 0; q < EPI_G; ++q)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
              rv[q][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)(offs[q] + colpart[jj]), 0, LD_AUX));
        }
#pragma unroll
        for (int q = 0; q < EPI_G; ++q) {
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) {
            const unsigned vo = offs[q] + colpart[jj];
            float v = acc[i][jj][r0 + q];
            if constexpr (MODE == MODE_DGRAD) {
              if (ref) v *= (rv[q][jj] > 0.f) ? g1 : g0;
            } else {
              v += bj[jj];
              v = (v > 0.f) ? v : v * p.slope;
              v *= p.gain;
              if (ref) v += rv[q][jj];
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, (int)vo, 0, ST_AUX);
          }
        }
      }
    };
    // (split-K slabs of the stride-1 data gradient stay plain, like the forward's: the reduce reads them back at once)
    if (st_nt && !(MODE == MODE_DGRAD && pl_dsplits > 1)) store_rows(std::true_type{}); else store_rows(std::false_type{});
  } else {
    const bool slab = (MODE == MODE_WGRAD) || p.ny > 1;              // split-K partial slab [split][M][Ncol]
    const bool pxm = (MODE == MODE_FWD) && pl_pixmajor;               // rows = images at one pixel: pitch = one image of y
    const int pitch = pxm ? px_pitch : (slab ? Ncol : d.ldy);
    const size_t out_off = pxm ? (size_t)px_off : (slab ? (size_t)by * M * Ncol : (size_t)0) + (size_t)m0 * pitch;
    float* outp = p.C + out_off;
    const int rows_here = pxm ? px_rows : min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(outp, 0, rows_here * pitch * 4, 0x00020000);
    const int wave_row = __builtin_amdgcn_readfirstlane(wm * WM);
    const float g1 = p.gain, g0 = p.gain * p.slope;
    // FWD residual merge: y = act(conv + bias) + addend, addend in y's own layout (same descriptor geometry)
    const bool has_add = (MODE == MODE_FWD) && !slab && p.addend != nullptr;                 // uniform
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_add ? p.addend + out_off : outp), 0, has_add ? rows_here * pitch * 4 : 0, 0x00020000);
    unsigned lanepart[TN];
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      lanepart[j] = colpart[j] + (unsigned)(4 * lhi * pitch) * 4u;
      const int c = n0 + wn * WN + j * 32 + l31;
      bj[j] = (MODE == MODE_FWD && !slab && p.bias && c < Ncol) ? p.bias[c] : 0.f;
    }
    if constexpr (MODE == MODE_FWD) {
      // (residual addend fetched EPI_F rows at a time before that group's stores: see the row-table branch above; the
      // 128 x 128 instance has registers for 4 rows)
      constexpr int EPI_F = (BM * BN > 128 * 64) ? 4 : EPI_G;
      auto store_fwd = [&](auto nt_c) {
      constexpr int ST_AUX = decltype(nt_c)::value ? 2 : 0;
      constexpr int LD_AUX = (LEAN_LD_NT && decltype(nt_c)::value) ? 2 : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EPI_F) {
          unsigned rowpart[EPI_F];
#pragma unroll
          for (int q = 0; q < EPI_F; ++q) {
            const int r = r0 + q;
            rowpart[q] = (unsigned)((wave_row + i * 32 + (r & 3) + 8 * (r >> 2)) * pitch) * 4u;   // wave-uniform
          }
          float rv[EPI_F][TN];
          if (has_add) {   // uniform
#pragma unroll
            for (int q = 0; q < EPI_F; ++q)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                rv[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (int)(lanepart[j] + rowpart[q]), 0, LD_AUX));
          }
#pragma unroll
          for (int q = 0; q < EPI_F; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              float v = acc[i][j][r0 + q];
              if (!slab) {
                v += bj[j];
                v *= (v > 0.f) ? g1 : g0;
                if (has_add) v += rv[q][j];
              }
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, (int)(lanepart[j] + rowpart[q]), 0, ST_AUX);
            }
        }
      };
      if (st_nt && !slab) store_fwd(std::true_type{}); else store_fwd(std::false_type{});      // (split-K slabs stay plain)
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned rowpart = (unsigned)((wave_row + i * 32 + (r & 3) + 8 * (r >> 2)) * pitch) * 4u;   // wave-uniform
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float v = acc[i][j][r];   // (a named copy: __builtin_bit_cast applied to the vector element directly read element 0 for every r -- hipcc 7.2)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, (int)(lanepart[j] + rowpart), 0, 0);
          }
        }
    }
    if constexpr (MODE == MODE_WGRAD) {
      if (do_bias) {
        float* red = smem;   // [B_RPP][BN]; the main-loop buffers are free after the last barrier
        // the store pieces of the last iteration added the zero fill of the tile past the end: harmless
        // (threads that own no piece of a BN = 32 tile loaded zero fills only: their colacc is 0)
        *reinterpret_cast<float4*>(red + b_r * BN + b_c4 * 4) = colacc;
        __syncthreads();
        if (tid < BN) {
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < B_RPP; ++r) sum += red[r * BN + tid];
          const int c = n0 + tid;
          if (c < Ncol) p.bias_ws[(size_t)by * Ncol + c] = sum;
        }
      }
    }
  }
}

template <int MODE, int BM, int BN, bool BAL = false>
__global__ __launch_bounds__(NTHREADS, IGEMM_MIN_WAVES) void igemm_lean_kernel(const IgemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // 1-D grid, remapped so that each XCD (own L2) owns a contiguous run of ids.  Decode order = who shares operands:
  //   FWD    tile_n fastest                      (the N-tiles of one M-tile read the same im2col rows)
  //   DGRAD  tile_n, then parity class, tile_m   (the s*s classes of one M-tile read the same gy pixels); stride 2:
  //          groups of M-tiles, class-major inside a group (equal work on neighbouring block ids)
  //   WGRAD  all (tile_m, tile_n) of one split   (every tile of a split reads the same positions of x and gy)
  // Before this the 9..36 tiles of a WGRAD split sat on 8 different XCDs and HBM traffic was 6.5x the algorithmic bytes.
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  int by, tile_n, tile_m;
  int nrep = 1;                                     // M-tiles this block walks (strided DGRAD, balanced order: 1 / 2 / 4)
  int wc_h0 = 0, wc_w0 = 0, wc_hc = 0, wc_wc = 0;   // border class of this tile: window origin and size (p.nwin > 0)
  if constexpr (MODE == MODE_WGRAD) {
    const int tiles = p.tiles_m * p.tiles_n;
    by = lin / tiles;
    const int b = lin - by * tiles;
    tile_n = b % p.tiles_n; tile_m = b / p.tiles_n;
  } else if constexpr (BAL) {
    // strided DGRAD, equal work per block (igemm.hip, dgrad_balance()): the parity classes of a 3x3 stride-2 layer contract
    // over 4 / 2 / 2 / 1 taps, so a block of class c walks cb_reps[c] = 1 / 2 / 2 / 4 consecutive M-tiles of its class -- every
    // block multiplies the same number of K-tiles.  Order: groups of cbal M-tile indices, class-major inside a group (the
    // gy rows the classes of a group share are still in the L2 when the next class reads them).
    const int per_group = p.cb_start[4] * p.tiles_n;
    const int g = lin / per_group, r = lin - g * per_group;
    const int blk = r / p.tiles_n;
    tile_n = r - blk * p.tiles_n;
    by = (blk >= p.cb_start[1]) + (blk >= p.cb_start[2]) + (blk >= p.cb_start[3]);
    nrep = p.cb_reps[by];
    tile_m = g * p.cbal + (blk - p.cb_start[by]) * nrep;
    if (tile_m >= p.tiles_m) return;
  } else if (MODE == MODE_DGRAD && p.cgroup > 0) {
    // strided DGRAD: groups of cgroup M-tiles, class-major inside a group (igemm.hip, contrad_conv2d_dgrad)
    const int per_group = p.cgroup * p.ny * p.tiles_n;
    const int g = lin / per_group, r = lin - g * per_group;
    by = r / (p.cgroup * p.tiles_n);
    const int q = r - by * (p.cgroup * p.tiles_n);
    tile_m = g * p.cgroup + q / p.tiles_n;
    tile_n = q % p.tiles_n;
    if (tile_m >= p.tiles_m) return;
  } else if (MODE != MODE_WGRAD && p.nwin > 0) {
    // border classes (igemm.hip, border_classes()): the output map (FWD) / dx map (stride-1 DGRAD) is cut into the <= 16
    // rectangles of pixels that share their set of non-padding taps; every class has its own run of M-tiles over
    // (image, pixel of the rectangle), heaviest class first
    // The classes carry unequal work per tile (9 / 6 / 4 taps), and the dispatcher places block b on XCD b % 8: every XCD
    // takes an eighth of EVERY class (contiguous in the class, so neighbours still share images in the L2), heaviest
    // class first.  (With the engine's usual contiguous-run-per-XCD order two XCDs got all the 9-tap tiles and the
    // kernel lasted exactly as long as without any skipping.)
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    tile_n = kb % p.tiles_n;
    int km = kb / p.tiles_n, c = 0;
    tile_m = -1;
    for (; c < p.nwin; ++c) {
      const int n_c = p.win_tile0[c + 1] - p.win_tile0[c];
      const int lo = (xcd * n_c) >> 3, hi = ((xcd + 1) * n_c) >> 3;
      if (km < hi - lo) { tile_m = lo + km; break; }
      km -= hi - lo;
    }
    if (tile_m < 0) return;     // (grid padded to 8 x the largest per-XCD share)
    by = 0;
    wc_h0 = p.win_h0[c]; wc_w0 = p.win_w0[c]; wc_hc = p.win_hc[c]; wc_wc = p.win_wc[c];
  } else {
    tile_n = lin % p.tiles_n;
    const int r = lin / p.tiles_n;
    by = r % p.ny; tile_m = r / p.ny;
  }
  if constexpr (BAL) {
    for (int rep = 0; rep < nrep; ++rep) {
      lean_tile<MODE, BM, BN, true>(p, smem, tile_m + rep, tile_n, by, 0, 0, 0, 0);
      if (rep + 1 < nrep) __syncthreads();   // (the epilogue's row table and the next tile's first LDS stores share smem)
    }
  } else {
    lean_tile<MODE, BM, BN, false>(p, smem, tile_m, tile_n, by, wc_h0, wc_w0, wc_hc, wc_wc);
  }
}

// smem: two buffers of A + B (<= 2 * (2112 + 2112) floats = 33 KB), or the epilogue's row table
template <int MODE, int BM, int BN>
constexpr size_t lean_smem_bytes() {
  const int a = (MODE != MODE_WGRAD) ? 4 * (BM * 4 + LEAN_QPAD) : 16 * BM;
  const int b = (MODE == MODE_DGRAD) ? 4 * (BN * 4 + LEAN_QPAD) : 16 * (BN < 64 ? 64 : BN);
  size_t main_loop = 2 * (size_t)(a + b) * sizeof(float);
  size_t epi = (size_t)BM * sizeof(long long);
  return main_loop > epi ? main_loop : epi;
}
