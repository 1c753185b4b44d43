// Implicit-GEMM convolution engine for gfx950 (MI355X): forward, data-gradient, weight-gradient.
//
//   C[M x Ncol] = A[M x Kg] * B[Kg x Ncol]  on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate; bitwise a
//   k-ordered fmaf chain, so results are fp32-exact in the cuDNN/ATen sense the reference relies on).
//
//   mode   | M rows           | Ncol | Kg (contraction)        | A gathered from        | B from
//   -------+------------------+------+-------------------------+------------------------+---------------
//   FWD    | n,ho,wo          | Cout | (kh,kw,c)               | x  (im2col, c-contig.) | Wp[k][cout]
//   DGRAD  | n,h,w (1 parity  | Cin  | (taps of class, cout)   | gy (cout-contiguous)   | Wp[(tap,c)][cout]^T
//          |  class / grid.y) |      |                         |                        |
//   WGRAD  | (kh,kw,c)        | Cout | n,ho,wo (split, grid.y) | x  (c-contiguous rows) | gy[p][cout]
//
// Two kernels share this file's host side (plans, tile choice, C ABI):
//   * igemm_lean_kernel (igemm_lean.h): the fast path for every shape whose K-tiles sit inside one filter tap
//     (channel counts % 16) -- buffer loads with hardware zero fill, scalar tap walk, quad LDS layout; see that file.
//   * igemm_kernel (below): the general path (Cin = 3, Cout = 1, 513 channels, odd WGRAD grids ...).  Block = 256
//     threads = 4 wave64 in a 2x2 arrangement, block tile BM x BN x 16; operands are staged global -> registers -> LDS
//     (double-buffered, one barrier per K-tile) with per-element predicate masks; LDS tiles are K-major ([k][row]):
//     K-contiguous operands are transposed on the LDS write (leading dimension BM+1), row-contiguous ones are written as
//     16-byte rows (leading dimension BM+4).  VEC = float4-addressable operands, !VEC = per-element gathers (64x64 only).
//
// blockIdx.x is remapped XCD-aware (common.h) with the N-tile index fastest, so the blocks that share an
// activation tile run on the same XCD/L2.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include "../../include/contrad_hip.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
#ifndef IGEMM_BK
#define IGEMM_BK 16
#endif
constexpr int BK = IGEMM_BK;          // K-tile depth.  16 (33 KB LDS, <= 128 VGPRs -> 4 blocks = 16 waves per CU) measured
                                      // +3.5 % over 32 (2 blocks per CU) on the SNDCGAN layers (tools/ab_bk.sh)
#ifndef IGEMM_MIN_WAVES
#define IGEMM_MIN_WAVES 4
#endif
constexpr int NTHREADS = 256;

struct IgemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;     // FWD
  const float* addend;   // FWD: tensor of y's shape / ldy added AFTER the activation (residual merge), or NULL
  const float* act_ref;  // DGRAD
  float* bias_ws;        // WGRAD: [splits][Ncol] partial column sums of gy (bias gradient), or NULL
  contrad_conv_desc d;
  float slope, gain;
  int M, Ncol, Kg;       // FWD / WGRAD gemm dims (DGRAD derives per class)
  int tiles_m, tiles_n;
  int P;                 // WGRAD: total positions n*ho*wo
  int ptiles_per_split;  // WGRAD
  int ny;                // lean kernels: grid.y folded into the 1-D grid (DGRAD classes / WGRAD splits)
  int cgroup;            // lean DGRAD, stride 2: M-tiles per class group of the block order (0: class-interleaved)
  // lean DGRAD, stride 2 with unequal parity classes (3x3: 4 / 2 / 2 / 1 taps), balanced order (cbal > 0): a block of class
  // c walks cb_reps[c] consecutive M-tiles, so that every block contracts the same number of K-tiles; a group covers cbal
  // M-tile indices of every class and holds cb_start[4] blocks per N-tile, class c at [cb_start[c], cb_start[c + 1])
  int cbal;
  int cb_reps[4], cb_start[5];
  int dsplits;           // lean DGRAD, stride 1: split-K count (grid.y = splits instead of parity classes), else 0 / 1
  long long slab_elems;  // lean DGRAD split-K: floats per partial slab (N * H * W * ldx)
  int px_pixels;         // lean DGRAD, pixel-major: dx pixels of the largest parity class (tiles_m = image blocks x this)
  // border classes (lean FWD / stride-1 DGRAD, nwin > 0): windows of the output / dx map whose pixels share their set of
  // non-padding taps; class c owns the M-tiles [win_tile0[c], win_tile0[c + 1]) of the launch
  short win_h0[16], win_w0[16], win_hc[16], win_wc[16];
  int win_tile0[17];
  int nwin;
  unsigned char px_order[256];   // pixel-major FWD / DGRAD: the pixel the k-th tile of an image block works on (balance, see pixel_order())
  int pixmajor;          // lean FWD / DGRAD: M-tiles are BM images at one output pixel (tiles_m = image blocks x Ho*Wo), padding taps skipped
  int px_full;           // px_order holds the order of ALL tiles of the launch (image block * pixels + pixel), pixel_order_full()
  int st_nt;             // lean FWD / DGRAD: non-temporal activation stores (outputs beyond the caches, st_nt_for())
};

// Non-temporal epilogue stores for activations that no cache will hold until their consumer runs (>= NT_STORE_BYTES = the
// 256 MB Infinity Cache), in the narrow lean instances (igemm_lean.h NT_CAP: the 128 x 128 FWD / DGRAD instances -- the
// headline's kernels, at the 128-VGPR limit -- spill with the second store loop; conv_c32 keeps plain stores).  Plain
// stores keep the written lines in the XCD's L2, where they evict operand lines the blocks are about to re-read.  Round 5,
// same-box alternations (profiles/r05_ab_stnt.txt): StyleGAN2_512 (activations of 0.1 - 1.6 GB): plain 62.81 ms per step
// (62.35 ... 63.42), non-temporal from 256 MB 62.54, every activation store non-temporal 62.25; in the final form (narrow
// instances only) 61.96 -> 61.76; the headline (outputs of 50 - 201 MB, which the NEXT layer finds in the Infinity Cache)
// LOSES with non-temporal stores: 15.681 -> 15.747 ms (5 / 5 alternations), StyleGAN2-32 15.215 -> 15.272 -- hence the
// threshold.  (The cache-policy operand of a buffer store is an immediate: the epilogue holds both store loops behind ONE
// launch-uniform branch; split-K slabs stay plain -- the reduce reads them back at once.)
constexpr long long NT_STORE_BYTES = 256ll << 20;
inline int st_nt_for(long long rows, int ld) {
#if defined(LEAN_ST_NT)          // (dev builds: -DLEAN_ST_NT=0 / 1 forces plain / non-temporal everywhere)
  if (LEAN_ST_NT >= 0) return LEAN_ST_NT;
#endif
  return rows * (long long)ld * (long long)sizeof(float) >= NT_STORE_BYTES ? 1 : 0;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Loads are BRANCH-FREE: an out-of-range element loads from a clamped (always valid) address and is zeroed by a
// predicate mask when the tile is written to LDS, one K-tile later.  (A "cond ? load : 0" select makes hipcc
// branch around every load and wait vmcnt(0) at the merge, which serialises the global-load latency with the MFMA
// phase -- cdna_hip_programming.md 5, trap (c).)  VEC = every operand row is 16-byte addressable in float4 units
// (C % 4 == 0, K % 4 == 0, leading dimensions % 4 == 0); the !VEC instantiation covers odd shapes (Cin = 3,
// Cout = 1) with per-element gathers and is only ever a tiny share of a step.
template <int MODE, int BM, int BN, bool VEC>
__global__ __launch_bounds__(NTHREADS, 2) void igemm_kernel(const IgemmArgs p) {   // (4 would spill the DGRAD 128x128 instance)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool A_KCONTIG = (MODE != MODE_WGRAD);
  constexpr bool B_KCONTIG = (MODE == MODE_DGRAD);
  constexpr int LDA = BM + (A_KCONTIG ? 1 : 4);
  constexpr int LDB = BN + (B_KCONTIG ? 1 : 4);
  constexpr int A_SZ = BK * LDA, B_SZ = BK * LDB;
  constexpr int PA = BM * BK / 1024, PB = BN * BK / 1024;  // float4 prefetch registers per operand
  constexpr int KQ = BK / 4;                 // k-quads per row of a K-contiguous tile
  constexpr int KROWS = NTHREADS / KQ;       // rows covered per pass by the K-contiguous loader
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int EPV = VEC ? 1 : 4;           // predicate bits per float4

  const contrad_conv_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = b % p.tiles_n, tile_m = b / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---------------- per-mode problem geometry ----------------
  int M = p.M, Ncol = p.Ncol, Kg = p.Kg;
  // DGRAD parity class
  int ph = 0, pw = 0, Hc = 0, Wc = 0, kh0 = 0, kw0 = 0, nth = 0, ntw = 1, bh = 0, bw = 0;
  if constexpr (MODE == MODE_DGRAD) {
    const int s = d.stride;
    ph = blockIdx.y / s;
    pw = blockIdx.y % s;
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
    Kg = nth * ntw * d.K;
    if (ntw == 0) ntw = 1;
    if (m0 >= M) return;  // uniform per block, before any barrier
  }
  // WGRAD split range over positions
  int p_begin = 0, p_end = 0;
  if constexpr (MODE == MODE_WGRAD) {
    p_begin = blockIdx.y * p.ptiles_per_split * BK;
    p_end = min(p.P, p_begin + p.ptiles_per_split * BK);
  }
  const int T = (MODE == MODE_WGRAD) ? ((p_end > p_begin) ? (p_end - p_begin + BK - 1) / BK : 0)
                                     : (Kg + BK - 1) / BK;

  // ---------------- loader state ----------------
  // K-contiguous operand: thread owns k-quad kq (4 consecutive k) of rows r + 32*i.
  const int kq = tid % KQ, krow = tid / KQ;
  // Row-contiguous operand: thread owns 4 consecutive columns c4*4 of k-rows r + RPP*i.
  constexpr int A_RPP = 1024 / BM, B_RPP = 1024 / BN;
  const int a_c4 = tid % (BM / 4), a_r = tid / (BM / 4);
  const int b_c4 = tid % (BN / 4), b_r = tid / (BN / 4);

  int a_n[PA], a_h[PA], a_w[PA];
  bool a_ok[PA];
  int wg_dn = 0, wg_dh = 0, wg_dw = 0;
  int wg_pbase = 0;   // WGRAD: first position of the NEXT tile load_a will fetch (advances with the row state)

  if constexpr (MODE == MODE_FWD) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + krow + KROWS * i;
      a_ok[i] = m < M;
      const int mm = a_ok[i] ? m : 0;
      const int wo = mm % d.Wo, t = mm / d.Wo;
      const int ho = t % d.Ho;
      a_n[i] = t / d.Ho;
      a_h[i] = ho * d.stride - d.pad;
      a_w[i] = wo * d.stride - d.pad;
    }
  } else if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + krow + KROWS * i;
      a_ok[i] = m < M;
      const int mm = a_ok[i] ? m : 0;
      const int wq = mm % Wc, t = mm / Wc;
      a_n[i] = t / Hc;
      a_h[i] = t % Hc + bh;  // ho = a_h - th
      a_w[i] = wq + bw;      // wo = a_w - tw
    }
  } else {
    const int hw = d.Ho * d.Wo;
    wg_pbase = p_begin;
    wg_dn = BK / hw;
    const int rem = BK - wg_dn * hw;
    wg_dh = rem / d.Wo;
    wg_dw = rem - wg_dh * d.Wo;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int pp = p_begin + a_r + A_RPP * i;
      const int wo = pp % d.Wo, t = pp / d.Wo;
      a_w[i] = wo;
      a_h[i] = t % d.Ho;
      a_n[i] = t / d.Ho;
      a_ok[i] = true;
    }
  }

  float4 ra[PA], rb[PB];
  unsigned amask = 0, bmask = 0;   // predicate bits of the tile currently held in ra / rb

  // element offsets (in floats) with validity, per mode ----------------------------------------------------
  // FWD   A element (row i, contraction index k): x[n, ho*s-p+kh, wo*s-p+kw, c]
  auto fwd_a = [&](int i, int k, bool& ok) -> size_t {
    const int kk = (k < Kg) ? k : 0;
    const int tap = kk / d.C, c = kk - tap * d.C;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int hi = a_h[i] + kh, wi = a_w[i] + kw;
    ok = (k < Kg) && a_ok[i] && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
    return ok ? ((size_t)(a_n[i] * d.H + hi) * d.W + wi) * d.ldx + c : 0;
  };
  // DGRAD A element: gy[n, hq+bh-th, wq+bw-tw, co];  B element (k, column c): wp[(tap, c), co]
  auto dg_decode = [&](int k, int& th, int& tw, int& co, int& tapflat) {
    const int kk = (k < Kg) ? k : 0;
    const int ti = kk / d.K;
    co = kk - ti * d.K;
    th = ti / ntw;
    tw = ti - th * ntw;
    tapflat = (kh0 + d.stride * th) * d.KW + (kw0 + d.stride * tw);
  };
  // WGRAD A element (position row i, gemm column ic): x[n, ho*s-p+kh, wo*s-p+kw, c]
  auto wg_a = [&](int i, int ic, bool pok, bool& ok) -> size_t {
    const int icc = (ic < Kg) ? ic : 0;
    const int tap = icc / d.C, c = icc - tap * d.C;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int hi = a_h[i] * d.stride - d.pad + kh, wi = a_w[i] * d.stride - d.pad + kw;
    ok = pok && (ic < Kg) && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
    return ok ? ((size_t)(a_n[i] * d.H + hi) * d.W + wi) * d.ldx + c : 0;
  };

  // ---------------- global -> register tile loads (no branches, no waits) ----------------
  // A and B are issued separately so the main loop can interleave them with MFMA k-steps.
  auto load_a_generic = [&](int t) {
    const int k0 = t * BK;
    amask = 0;
    if constexpr (MODE == MODE_FWD) {
      const int k = k0 + kq * 4;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        if constexpr (VEC) {
          bool ok;
          const size_t off = fwd_a(i, k, ok);
          ra[i] = ld4(p.A + off);
          amask |= (unsigned)ok << i;
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool ok;
            const size_t off = fwd_a(i, k + j, ok);
            v[j] = p.A[off];
            amask |= (unsigned)ok << (4 * i + j);
          }
          ra[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      const int k = k0 + kq * 4;
      if constexpr (VEC) {
        int th, tw, co, tapflat;
        dg_decode(k, th, tw, co, tapflat);
        const bool kok = k < Kg;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int ho = a_h[i] - th, wo = a_w[i] - tw;
          const bool ok = kok && a_ok[i] && (unsigned)ho < (unsigned)d.Ho && (unsigned)wo < (unsigned)d.Wo;
          ra[i] = ld4(p.A + (ok ? ((size_t)(a_n[i] * d.Ho + ho) * d.Wo + wo) * d.ldy + co : 0));
          amask |= (unsigned)ok << i;
        }
      } else {
        float va[PA][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int th, tw, co, tapflat;
          dg_decode(k + j, th, tw, co, tapflat);
          const bool kok = (k + j) < Kg;
#pragma unroll
          for (int i = 0; i < PA; ++i) {
            const int ho = a_h[i] - th, wo = a_w[i] - tw;
            const bool ok = kok && a_ok[i] && (unsigned)ho < (unsigned)d.Ho && (unsigned)wo < (unsigned)d.Wo;
            va[i][j] = p.A[ok ? ((size_t)(a_n[i] * d.Ho + ho) * d.Wo + wo) * d.ldy + co : 0];
            amask |= (unsigned)ok << (4 * i + j);
          }
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = make_float4(va[i][0], va[i][1], va[i][2], va[i][3]);
      }
    } else {  // WGRAD
      // positions come from the running state (not from t): a repeated call on the last tile then points past
      // p_end and is fully predicated off instead of pairing tile t's range with tile t+1's coordinates
      const int pbase = wg_pbase;
      wg_pbase += BK;
      const int icol = m0 + a_c4 * 4;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int pp = pbase + a_r + A_RPP * i;
        const bool pok = pp < p_end;
        if constexpr (VEC) {
          bool ok;
          const size_t off = wg_a(i, icol, pok, ok);
          ra[i] = ld4(p.A + off);
          amask |= (unsigned)ok << i;
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool ok;
            const size_t off = wg_a(i, icol + j, pok, ok);
            v[j] = p.A[off];
            amask |= (unsigned)ok << (4 * i + j);
          }
          ra[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        // advance this row's (n,ho,wo) by BK positions for the next tile
        a_w[i] += wg_dw;
        if (a_w[i] >= d.Wo) { a_w[i] -= d.Wo; a_h[i] += 1; }
        a_h[i] += wg_dh;
        if (a_h[i] >= d.Ho) { a_h[i] -= d.Ho; a_n[i] += 1; }
        a_n[i] += wg_dn;
      }
    }
  };

  auto load_b_generic = [&](int t) {
    const int k0 = t * BK;
    bmask = 0;
    if constexpr (MODE == MODE_FWD) {
      const int col = n0 + b_c4 * 4;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int kr = k0 + b_r + B_RPP * i;
        if constexpr (VEC) {
          const bool ok = kr < Kg && col < Ncol;
          rb[i] = ld4(p.B + (ok ? (size_t)kr * d.ldw + col : 0));
          bmask |= (unsigned)ok << i;
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = kr < Kg && col + j < Ncol;
            v[j] = p.B[ok ? (size_t)kr * d.ldw + col + j : 0];
            bmask |= (unsigned)ok << (4 * i + j);
          }
          rb[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      const int k = k0 + kq * 4;
      if constexpr (VEC) {
        int th, tw, co, tapflat;
        dg_decode(k, th, tw, co, tapflat);
        const bool kok = k < Kg;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int c = n0 + krow + KROWS * i;
          const bool ok = kok && c < Ncol;
          rb[i] = ld4(p.B + (ok ? ((size_t)tapflat * d.C + c) * d.ldw + co : 0));
          bmask |= (unsigned)ok << i;
        }
      } else {
        float vb[PB][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int th, tw, co, tapflat;
          dg_decode(k + j, th, tw, co, tapflat);
          const bool kok = (k + j) < Kg;
#pragma unroll
          for (int i = 0; i < PB; ++i) {
            const int c = n0 + krow + KROWS * i;
            const bool ok = kok && c < Ncol;
            vb[i][j] = p.B[ok ? ((size_t)tapflat * d.C + c) * d.ldw + co : 0];
            bmask |= (unsigned)ok << (4 * i + j);
          }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = make_float4(vb[i][0], vb[i][1], vb[i][2], vb[i][3]);
      }
    } else {  // WGRAD
      const int pbase = p_begin + k0;
      const int col = n0 + b_c4 * 4;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int pp = pbase + b_r + B_RPP * i;
        if constexpr (VEC) {
          const bool ok = pp < p_end && col < Ncol;
          rb[i] = ld4(p.B + (ok ? (size_t)pp * d.ldy + col : 0));
          bmask |= (unsigned)ok << i;
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = pp < p_end && col + j < Ncol;
            v[j] = p.B[ok ? (size_t)pp * d.ldy + col + j : 0];
            bmask |= (unsigned)ok << (4 * i + j);
          }
          rb[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  };

  // ---------------- VEC fast path: piecewise, division-free loaders ----------------
  // Per-row base offsets are computed once; the contraction index of this thread's k-quad is decoded incrementally
  // (channel offset + tap counters advanced by BK per tile), so one tile's address work is a handful of adds and
  // compares per row.  The loaders are split into per-row "pieces" that the main loop drops between MFMA k-steps.
  int a_base[PA], b_base[PB];
  int st_c = 0, st_h = 0, st_w = 0;      // decode state for the NEXT tile: channel offset, tap row, tap column
  int cu_c = 0, cu_h = 0, cu_w = 0;      // snapshot for the tile whose pieces are currently being issued
  bool cu_ok = false;
  int b_k0 = 0;                          // FWD: first packed-weight row of the next tile; generic: tile counter
  int gen_t = 0;
  if constexpr (VEC) {
    if constexpr (MODE == MODE_FWD) {
      const int k = kq * 4;
      const int tap = k / d.C;
      st_c = k - tap * d.C;
      st_h = tap / d.KW;
      st_w = tap - st_h * d.KW;
#pragma unroll
      for (int i = 0; i < PA; ++i) a_base[i] = ((a_n[i] * d.H + a_h[i]) * d.W + a_w[i]) * d.ldx;
#pragma unroll
      for (int i = 0; i < PB; ++i) b_base[i] = (b_r + B_RPP * i) * d.ldw + n0 + b_c4 * 4;
    } else if constexpr (MODE == MODE_DGRAD) {
      const int k = kq * 4;
      const int ti = k / d.K;
      st_c = k - ti * d.K;
      st_h = ti / ntw;
      st_w = ti - st_h * ntw;
#pragma unroll
      for (int i = 0; i < PA; ++i) a_base[i] = ((a_n[i] * d.Ho + a_h[i]) * d.Wo + a_w[i]) * d.ldy;
#pragma unroll
      for (int i = 0; i < PB; ++i) b_base[i] = (n0 + krow + KROWS * i) * d.ldw;
    } else {
#pragma unroll
      for (int i = 0; i < PB; ++i) b_base[i] = n0 + b_c4 * 4;
    }
  }

  // snapshot the decode of the tile about to be loaded and advance the state by BK (called once per tile, before
  // its first piece).  Past the last tile the tap counters run off the end and everything is predicated off.
  auto begin_tile = [&]() {
    if constexpr (VEC && MODE != MODE_WGRAD) {
      cu_c = st_c; cu_h = st_h; cu_w = st_w;
      const int cdim = (MODE == MODE_FWD) ? d.C : d.K;
      const int wdim = (MODE == MODE_FWD) ? d.KW : ntw;
      cu_ok = (MODE == MODE_FWD) ? (cu_h < d.KH) : (cu_h < nth);
      st_c += BK;
      while (st_c >= cdim) {
        st_c -= cdim;
        if (++st_w == wdim) { st_w = 0; ++st_h; }
      }
    }
    amask = 0;
    bmask = 0;
  };

  auto load_a_piece = [&](int i) {
    if constexpr (!VEC) {
      if (i == 0) load_a_generic(gen_t);
    } else if constexpr (MODE == MODE_FWD) {
      const int hi = a_h[i] + cu_h, wi = a_w[i] + cu_w;
      const bool ok = cu_ok && a_ok[i] && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
      const int off = a_base[i] + (cu_h * d.W + cu_w) * d.ldx + cu_c;
      ra[i] = ld4(p.A + (ok ? off : 0));
      amask |= (unsigned)ok << i;
    } else if constexpr (MODE == MODE_DGRAD) {
      const int ho = a_h[i] - cu_h, wo = a_w[i] - cu_w;
      const bool ok = cu_ok && a_ok[i] && (unsigned)ho < (unsigned)d.Ho && (unsigned)wo < (unsigned)d.Wo;
      const int off = a_base[i] - (cu_h * d.Wo + cu_w) * d.ldy + cu_c;
      ra[i] = ld4(p.A + (ok ? off : 0));
      amask |= (unsigned)ok << i;
    } else {
      const int pp = wg_pbase + a_r + A_RPP * i;
      const int icol = m0 + a_c4 * 4;
      bool ok;
      const size_t off = wg_a(i, icol, pp < p_end, ok);
      ra[i] = ld4(p.A + off);
      amask |= (unsigned)ok << i;
      a_w[i] += wg_dw;
      if (a_w[i] >= d.Wo) { a_w[i] -= d.Wo; a_h[i] += 1; }
      a_h[i] += wg_dh;
      if (a_h[i] >= d.Ho) { a_h[i] -= d.Ho; a_n[i] += 1; }
      a_n[i] += wg_dn;
    }
  };

  auto load_b_piece = [&](int i) {
    if constexpr (!VEC) {
      if (i == 0) { load_b_generic(gen_t); ++gen_t; }
    } else if constexpr (MODE == MODE_FWD) {
      const int kr = b_k0 + b_r + B_RPP * i;
      const bool ok = kr < Kg && (n0 + b_c4 * 4) < Ncol;
      rb[i] = ld4(p.B + (ok ? b_k0 * d.ldw + b_base[i] : 0));
      bmask |= (unsigned)ok << i;
      if (i == PB - 1) b_k0 += BK;
    } else if constexpr (MODE == MODE_DGRAD) {
      const int tapflat = (kh0 + d.stride * cu_h) * d.KW + (kw0 + d.stride * cu_w);
      const bool ok = cu_ok && (n0 + krow + KROWS * i) < Ncol;
      rb[i] = ld4(p.B + (ok ? tapflat * d.C * d.ldw + b_base[i] + cu_c : 0));
      bmask |= (unsigned)ok << i;
    } else {
      const int pp = wg_pbase + b_r + B_RPP * i;
      const bool ok = pp < p_end && b_base[i] < Ncol;
      rb[i] = ld4(p.B + (ok ? (size_t)pp * d.ldy + b_base[i] : 0));
      bmask |= (unsigned)ok << i;
      if (i == PB - 1) wg_pbase += BK;
    }
  };

  auto masked = [&](float4 v, unsigned mask, int i) -> float4 {
    if constexpr (VEC) {
      if (!((mask >> i) & 1u)) v = zero4();
    } else {
      if (!((mask >> (4 * i + 0)) & 1u)) v.x = 0.f;
      if (!((mask >> (4 * i + 1)) & 1u)) v.y = 0.f;
      if (!((mask >> (4 * i + 2)) & 1u)) v.z = 0.f;
      if (!((mask >> (4 * i + 3)) & 1u)) v.w = 0.f;
    }
    return v;
  };

  // ---------------- register -> LDS, one row-group per piece (predicates applied here) ----------------
  auto store_a_piece = [&](int buf, int i) {
    float* As = smem + buf * (A_SZ + B_SZ);
    const float4 v = masked(ra[i], amask, i);
    if constexpr (A_KCONTIG) {
      const int row = krow + KROWS * i;
      As[(kq * 4 + 0) * LDA + row] = v.x;
      As[(kq * 4 + 1) * LDA + row] = v.y;
      As[(kq * 4 + 2) * LDA + row] = v.z;
      As[(kq * 4 + 3) * LDA + row] = v.w;
    } else {
      *reinterpret_cast<float4*>(As + (a_r + A_RPP * i) * LDA + a_c4 * 4) = v;
    }
  };
  // WGRAD bias gradient: the B operand IS gy, so the first M-tile's blocks also accumulate its column sums
  // (flag multiply instead of a branch keeps the store pieces straight-line).
  float4 colacc = zero4();
  const float bias_flag = (MODE == MODE_WGRAD && p.bias_ws != nullptr && tile_m == 0) ? 1.f : 0.f;
  auto store_b_piece = [&](int buf, int i) {
    float* Bs = smem + buf * (A_SZ + B_SZ) + A_SZ;
    const float4 v = masked(rb[i], bmask, i);
    if constexpr (MODE == MODE_WGRAD && VEC) {
      colacc.x = fmaf(v.x, bias_flag, colacc.x); colacc.y = fmaf(v.y, bias_flag, colacc.y);
      colacc.z = fmaf(v.z, bias_flag, colacc.z); colacc.w = fmaf(v.w, bias_flag, colacc.w);
    }
    if constexpr (B_KCONTIG) {
      const int row = krow + KROWS * i;
      Bs[(kq * 4 + 0) * LDB + row] = v.x;
      Bs[(kq * 4 + 1) * LDB + row] = v.y;
      Bs[(kq * 4 + 2) * LDB + row] = v.z;
      Bs[(kq * 4 + 3) * LDB + row] = v.w;
    } else {
      *reinterpret_cast<float4*>(Bs + (b_r + B_RPP * i) * LDB + b_c4 * 4) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lhi = lane >> 5;

  if (T > 0) {
    begin_tile();
#pragma unroll
    for (int i = 0; i < PA; ++i) load_a_piece(i);
#pragma unroll
    for (int i = 0; i < PB; ++i) load_b_piece(i);
#pragma unroll
    for (int i = 0; i < PA; ++i) store_a_piece(0, i);
#pragma unroll
    for (int i = 0; i < PB; ++i) store_b_piece(0, i);
  }
  __syncthreads();

  // Main loop.  Every wave overlaps its OWN memory work with its OWN MFMA phase (the two waves that share a SIMD run
  // in lock-step through the shared matrix pipe, so relying on the partner wave to cover the load / LDS-store phases
  // left the pipe ~30 % idle): the next tile's global loads are dropped, one row-group at a time, between the first
  // k-steps, its LDS stores (into the other buffer) between the last ones; sched_barriers pin that placement.
  // There is no "is there a next tile" branch: past the last tile the loaders are predicated off (clamped addresses)
  // and the stores fill the idle buffer.  (With a branch hipcc cannot pair "loads issued" with "stores executed",
  // assumes loads may be pending at the back-edge and waits vmcnt right after issuing them.)
  constexpr int KS = BK / 2;
  constexpr int A_LD0 = 0, B_LD0 = PA, A_ST0 = KS - PA - PB, B_ST0 = KS - PB;
  static_assert(A_ST0 >= B_LD0 + PB, "not enough k-steps to schedule the load and store pieces");
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    const float* As = smem + cur * (A_SZ + B_SZ) + wm * WM + l31;
    const float* Bs = smem + cur * (A_SZ + B_SZ) + A_SZ + wn * WN + l31;
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = As[lhi * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = Bs[lhi * LDB + j * 32];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cb = ks & 1, nb = cb ^ 1;
#if !defined(IGEMM_ABLATE_LOADS)
      if (ks == A_LD0) begin_tile();
      if (ks >= A_LD0 && ks < A_LD0 + PA) load_a_piece(ks - A_LD0);
      if (ks >= B_LD0 && ks < B_LD0 + PB) load_b_piece(ks - B_LD0);
#endif
      if (ks + 1 < KS) {
        const int k = (ks + 1) * 2 + lhi;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nb][i] = As[k * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nb][j] = Bs[k * LDB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][i], bv[cb][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#if !defined(IGEMM_ABLATE_STORES)
      if (ks >= A_ST0 && ks < A_ST0 + PA) store_a_piece(cur ^ 1, ks - A_ST0);
      if (ks >= B_ST0 && ks < B_ST0 + PB) store_b_piece(cur ^ 1, ks - B_ST0);
#endif
    }
#if !defined(IGEMM_ABLATE_BARRIER)
    __syncthreads();
#endif
  }

  // ---------------- epilogue ----------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  if constexpr (MODE == MODE_DGRAD) {
    // row -> dx pixel offset table (in floats / ldx), staged in LDS (main-loop buffers are free now)
    long long* rowoff = reinterpret_cast<long long*>(smem);
    if (tid < BM) {
      const int m = m0 + tid;
      long long off = -1;
      if (m < M) {
        const int wq = m % Wc, t2 = m / Wc;
        const int hq = t2 % Hc, n = t2 / Hc;
        off = ((long long)(n * d.H + hq * d.stride + ph) * d.W + (wq * d.stride + pw)) * d.ldx;
      }
      rowoff[tid] = off;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const long long off = rowoff[row];
        if (off < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = n0 + wn * WN + j * 32 + l31;
          if (c < Ncol) {
            float v = acc[i][j][r];
            if (p.act_ref) v *= (p.act_ref[off + c] > 0.f) ? p.gain : p.gain * p.slope;
            p.C[off + c] = v;
          }
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = n0 + wn * WN + j * 32 + l31;
          if (c < Ncol) {
            float v = acc[i][j][r];
            if constexpr (MODE == MODE_FWD) {
              if (p.bias) v += p.bias[c];
              v = (v > 0.f) ? v : v * p.slope;
              v *= p.gain;
              if (p.addend) v += p.addend[(size_t)m * d.ldy + c];
              p.C[(size_t)m * d.ldy + c] = v;
            } else {
              p.C[((size_t)blockIdx.y * M + m) * Ncol + c] = v;
            }
          }
        }
      }
    if constexpr (MODE == MODE_WGRAD && VEC) {
      if (p.bias_ws != nullptr && tile_m == 0) {   // uniform per block
        // the last (redundant) store piece of the loop added only predicated-off zeros
        float* red = smem;                           // [B_RPP][BN], main-loop buffers are free after the barrier
        *reinterpret_cast<float4*>(red + b_r * BN + b_c4 * 4) = colacc;
        __syncthreads();
        if (tid < BN) {
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < B_RPP; ++r) sum += red[r * BN + tid];
          const int c = n0 + tid;
          if (c < Ncol) p.bias_ws[(size_t)blockIdx.y * Ncol + c] = sum;
        }
      }
    }
  }
}

// dwp[i][co] = sum_s ws[s][i][co]   (fixed summation order -> deterministic).
// R (power of two <= 64) consecutive lanes share one output element group and each sums the splits r, r + R, ... ; the R
// partial sums are combined by a fixed xor-butterfly.  Small outputs with hundreds of splits (the 288 x 32 weight gradient
// of a 32-channel layer at 512 x 512 has 341) otherwise run as 9 blocks of threads that each walk all splits serially:
// 55 us per call, 3.2 ms per StyleGAN2-512 step.
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int Kg, int Ncol,
                                    int ldw, int splits, const float* __restrict__ bias_ws,
                                    float* __restrict__ dbias, int R) {
  const long long total = (long long)Kg * Ncol;
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(t0 % R);
  const long long g0 = t0 / R, gstride = nthreads / R;
  if (ldw == Ncol && (total & 3) == 0) {   // dense packed rows: float4 streams, no index arithmetic
    const long long n4 = total / 4;
    for (long long q0 = g0 - (g0 % (64 / R)); q0 < n4; q0 += gstride) {   // (whole waves iterate together: shuffles below)
      const long long q = q0 + (g0 % (64 / R));
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < n4) {
        // four slabs per trip, all four loads issued before the first add: the serial "load, add, load, add" chain was
        // latency-bound (2.8 TB/s on 54 MB).  Fixed order: ((k, k+R) + (k+2R, k+3R)) per trip, trips in sequence.
        int k = r;
        for (; k + 3 * R < splits; k += 4 * R) {
          const float4 v0 = *reinterpret_cast<const float4*>(ws + (long long)k * total + q * 4);
          const float4 v1 = *reinterpret_cast<const float4*>(ws + (long long)(k + R) * total + q * 4);
          const float4 v2 = *reinterpret_cast<const float4*>(ws + (long long)(k + 2 * R) * total + q * 4);
          const float4 v3 = *reinterpret_cast<const float4*>(ws + (long long)(k + 3 * R) * total + q * 4);
          s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
          s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; k < splits; k += R) {
          const float4 v = *reinterpret_cast<const float4*>(ws + (long long)k * total + q * 4);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
      for (int o = R >> 1; o > 0; o >>= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
        s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
      }
      if (r == 0 && q < n4) *reinterpret_cast<float4*>(out + q * 4) = s;
    }
  } else {
    for (long long e0 = g0 - (g0 % (64 / R)); e0 < total; e0 += gstride) {
      const long long e = e0 + (g0 % (64 / R));
      float s = 0.f;
      if (e < total)
        for (int k = r; k < splits; k += R) s += ws[(long long)k * total + e];
      for (int o = R >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (r == 0 && e < total) {
        const int i = (int)(e / Ncol), c = (int)(e - (long long)i * Ncol);
        out[(size_t)i * ldw + c] = s;
      }
    }
  }
  if (dbias)
    for (long long c0 = g0 - (g0 % (64 / R)); c0 < Ncol; c0 += gstride) {
      const long long c = c0 + (g0 % (64 / R));
      float s = 0.f;
      if (c < Ncol)
        for (int k = r; k < splits; k += R) s += bias_ws[(long long)k * Ncol + c];
      for (int o = R >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (r == 0 && c < Ncol) dbias[c] = s;
    }
}

template <int BM, int BN>
constexpr size_t smem_bytes(int mode) {
  const int lda = BM + ((mode != MODE_WGRAD) ? 1 : 4);
  const int ldb = BN + ((mode == MODE_DGRAD) ? 1 : 4);
  size_t main_loop = 2 * (size_t)(BK * lda + BK * ldb) * sizeof(float);
  size_t epi = (size_t)BM * sizeof(long long);
  return main_loop > epi ? main_loop : epi;
}

template <int MODE, int BM, int BN, bool VEC>
int launch(const IgemmArgs& a, dim3 grid, hipStream_t stream) {
  static bool attr_set_dev[64] = {};  // per device (one process may drive several); benign race: idempotent
  const size_t smem = smem_bytes<BM, BN>(MODE);
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<MODE, BM, BN, VEC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, VEC>), grid, dim3(NTHREADS), smem, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

#include "igemm_lean.h"
#include "wgrad_c32.h"
#include "conv_c32.h"
#include "wino.h"
#include "wino22.h"
#include "wino44.h"
#include "wino44n.h"
#include "wino23.h"

// ---------------- Winograd F(2x2, 3x3) path (wino.h): 3x3 stride-1 pad-1 layers, forward and data gradient ----------------
// Can the shape run on wino_kernel<mode> at all?  (input channels % 16, output channels % 64, power-of-two maps >= 4)
bool wino_ok(const contrad_conv_desc* d, int mode) {
  if (mode != MODE_FWD && mode != MODE_DGRAD) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
  if (d->H < 4 || d->W < 4 || (d->H & (d->H - 1)) || (d->W & (d->W - 1))) return false;
  const int cin = mode == MODE_FWD ? d->C : d->K, cout = mode == MODE_FWD ? d->K : d->C;
  const int ldi = mode == MODE_FWD ? d->ldx : d->ldy, ldo = mode == MODE_FWD ? d->ldy : d->ldx;
  if ((cin & 15) || (cout & 63) || (ldi & 3) || (d->ldw & 3)) return false;
  const int th = std::min(8, d->H / 2), tw = std::min(8, d->W / 2);
  const long long nimg = 64 / (th * tw);
  if (nimg > 16) return false;
  const long long lim = 1ll << 31;
  if (nimg * d->H * d->W * std::max(ldi, ldo) * 4 >= lim) return false;     // block-relative byte offsets
  if (16ll * cin * cout * 4 >= lim) return false;
  return true;
}

wino::Args wino_args(const contrad_conv_desc* d, int mode) {
  wino::Args a{};
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = mode == MODE_FWD ? d->C : d->K;
  a.Cout = mode == MODE_FWD ? d->K : d->C;
  a.ldi = mode == MODE_FWD ? d->ldx : d->ldy;
  a.ldo = mode == MODE_FWD ? d->ldy : d->ldx;
  a.TH = std::min(8, d->H / 2); a.TW = std::min(8, d->W / 2);
  a.sh_tw = __builtin_ctz(a.TW); a.sh_thw = __builtin_ctz(a.TH * a.TW);
  a.NIMG = 64 / (a.TH * a.TW);
  a.PH = d->H / (2 * a.TH); a.PW = d->W / (2 * a.TW);
  a.NP = cdiv(d->N, a.NIMG) * a.PH * a.PW;
  a.NKB = a.Cout / 64;
  // raw box: with the halo (pixels outside the image load as zeros) when the image has several patches along the axis,
  // else the image itself (the halo reads the zero pixel)
  a.BH = a.PH > 1 ? 2 * a.TH + 2 : d->H; a.r_org = a.PH > 1 ? -1 : 0;
  a.BW = a.PW > 1 ? 2 * a.TW + 2 : d->W; a.c_org = a.PW > 1 ? -1 : 0;
  return a;
}

long long wino_items(const contrad_conv_desc* d, int mode) {
  const wino::Args a = wino_args(d, mode);
  return (long long)a.NP * a.NKB;
}

constexpr int WINO_CUS = 256;      // one persistent block per CU of the MI355X
bool wino44_planned(const contrad_conv_desc* d, int mode);     // (below: F(4x4, 3x3) goes first)

// Does the plan send the layer there?  A block is a whole CU and an item (64 tiles x 64 couts x all channels) its unit of
// work: the launch needs about a round of items, and the last round must not be mostly empty.
bool wino_planned(const contrad_conv_desc* d, int mode) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO"); return !(e && e[0] == '0'); }();
  if (!enabled || !wino_ok(d, mode)) return false;
  if (wino44_planned(d, mode)) return false;       // (F(4x4, 3x3) takes the launch)
  const long long items = wino_items(d, mode);
  const long long rounds = cdivll(items, WINO_CUS);
  static const long long min_items = []() { const char* e = contrad_dev_env("CONTRAD_WINO_MIN_ITEMS"); return e ? atoll(e) : 150ll; }();
  // (a single partial round: from 150 items.  Per-rank batches of the headline config on one GPU, profiles/r06_ab_plan_thresholds.txt:
  // 200 / 150 / 90 items -> 3.12 / 2.82 / 2.89 ms per step at batch 64, 4.39 / 4.25 / 4.24 at batch 128)
  if (items < WINO_CUS) return items >= min_items;
  return rounds * WINO_CUS * 10 <= items * 14;
}

long long wino_workspace_bytes(const contrad_conv_desc* d) { return 16ll * d->C * d->K * (long long)sizeof(float); }

int wino_grid(const wino::Args& a) {
  const int l0 = cdiv(a.NP, 8) * a.NKB;           // items of the fullest XCD
  return 8 * std::min(WINO_CUS / 8, l0);
}


// ---------------- Winograd F(4x4, 3x3) path (wino44.h): the same layers as wino.h on maps of 8x8 and larger ----------------
// (input channels % 32, output channels % 64 -- or an odd multiple of 32: wino44n.h --, power-of-two maps >= 8; 2.25 multiply-adds per
// output instead of 4)
bool wino44_ok(const contrad_conv_desc* d, int mode) {
  if (mode != MODE_FWD && mode != MODE_DGRAD) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
  // (4x4 maps: wino44n_kernel only -- a tile is an image, 32 images per item)
  if (d->H < 4 || d->W < 4 || (d->H & (d->H - 1)) || (d->W & (d->W - 1)) || (d->W < 32 ? d->H != d->W : d->H < 16)) return false;
  const int cin = mode == MODE_FWD ? d->C : d->K, cout = mode == MODE_FWD ? d->K : d->C;
  const int ldi = mode == MODE_FWD ? d->ldx : d->ldy, ldo = mode == MODE_FWD ? d->ldy : d->ldx;
  // (output channels: whole 64-wide blocks -- wino44_kernel -- or an odd number of 32-wide ones -- wino44n_kernel, wino44n.h)
  if ((cin & 31) || (cout & 31) || (ldi & 3) || (d->ldw & 3)) return false;
  const long long lim = 1ll << 31;
  const long long nimg = d->W >= 32 ? 1 : d->W == 16 ? 2 : d->W == 8 ? 8 : 32;
  if (nimg * d->H * d->W * std::max(ldi, ldo) * 4 >= lim) return false;     // block-relative byte offsets
  if (36ll * cin * cout * 4 >= lim) return false;
  return true;
}

// Does a launch of `items` whole-CU items fill the chip well enough?  A single round from 230 items, else a last round that is not
// mostly empty (rounds x CUs <= 1.4 x items).
static inline bool wino44_round_ok(long long items) {
  static const long long min_items = []() { const char* e = contrad_dev_env("CONTRAD_WINO44_MIN_ITEMS"); return e ? atoll(e) : 230ll; }();
  if (items < WINO_CUS) return items >= min_items;
  return cdivll(items, WINO_CUS) * WINO_CUS * 10 <= items * 14;
}

// patches (32 tiles each) of a launch: images / images per item x patches per image
static inline long long wino44_patches(const contrad_conv_desc* d) {
  const int TW = std::min(8, d->W / 4), TH = std::min(4, d->H / 4);
  return (long long)cdiv(d->N, 32 / (TH * TW)) * (d->H / (4 * TH)) * (d->W / (4 * TW));
}

// wino44n_kernel (items of 32 tiles x 32 couts) instead of wino44_kernel (x 64): output channels that are not whole 64-wide blocks;
// the 4x4 maps (1536 images x 512 couts: 768 items = three full rounds where 64-wide blocks give one and a half); and launches
// whose 64-wide items do not fill the chip while twice as many half items do (192 images of 8 x 8 x 512: 192 -> 384 items, 0.30 ->
// 0.24 ms; of 16 x 16 x 128: 0.095 -> 0.079 ms -- with enough items the 64-wide blocks are 10 - 25 % faster: one exchange and one
// transform of V per 64 couts instead of per 32, profiles/r06_ab_wino44n_plan.txt)
static inline bool wino44_n32(const contrad_conv_desc* d, int mode) {
  static const bool all = []() { const char* e = contrad_dev_env("CONTRAD_WINO44N_ALL"); return e && e[0] == '1'; }();      // (dev: every shape on it)
  static const bool fill = []() { const char* e = contrad_dev_env("CONTRAD_WINO44N_FILL"); return !(e && e[0] == '0'); }();  // (dev: the third rule off)
  const int cout = mode == MODE_FWD ? d->K : d->C;
  if ((cout & 63) != 0 || d->W == 4 || all) return true;
  const long long items64 = wino44_patches(d) * (cout / 64);
  return fill && !wino44_round_ok(items64) && wino44_round_ok(2 * items64);
}

wino44::Args wino44_args(const contrad_conv_desc* d, int mode) {
  wino44::Args a{};
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = mode == MODE_FWD ? d->C : d->K;
  a.Cout = mode == MODE_FWD ? d->K : d->C;
  a.ldi = mode == MODE_FWD ? d->ldx : d->ldy;
  a.ldo = mode == MODE_FWD ? d->ldy : d->ldx;
  a.TW = std::min(8, d->W / 4); a.TH = std::min(4, d->H / 4);      // 4x4-pixel tiles per image part of an item: 4 x 8, 4 x 4 (16x16 maps), 2 x 2 (8x8), 1 (4x4)
  a.sh_tw = __builtin_ctz(a.TW); a.sh_thw = __builtin_ctz(a.TH * a.TW);
  a.NIMG = 32 / (a.TH * a.TW);
  a.PH = d->H / (4 * a.TH); a.PW = d->W / (4 * a.TW);
  a.NP = cdiv(d->N, a.NIMG) * a.PH * a.PW;
  a.n32 = wino44_n32(d, mode) ? 1 : 0;
  a.NKB = a.n32 ? a.Cout / 32 : a.Cout / 64;      // 32-wide cout blocks: the items of wino44n_kernel
  a.BH = 4 * a.TH + 2; a.BW = 4 * a.TW + 2;       // raw box: always with the halo
  return a;
}

long long wino44_items(const contrad_conv_desc* d, int mode) {
  const wino44::Args a = wino44_args(d, mode);
  return (long long)a.NP * a.NKB;
}

// An item is 512 output pixels x 64 (32) couts x all channels on a whole CU: twice wino.h's.  The plan takes F(4x4, 3x3) when the
// launch has a full round of them and its last round is not mostly empty; else the layer falls through to wino_planned.
bool wino44_planned(const contrad_conv_desc* d, int mode) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO44"); return !(e && e[0] == '0'); }();
  static const bool enabled2 = []() { const char* e = contrad_dev_env("CONTRAD_WINO"); return !(e && e[0] == '0'); }();
  if (!enabled || !enabled2 || !wino44_ok(d, mode)) return false;
  static const bool enabled_n = []() { const char* e = contrad_dev_env("CONTRAD_WINO44N"); return !(e && e[0] == '0'); }();
  if (!enabled_n && wino44_n32(d, mode)) return false;       // (32-wide cout blocks: wino44n.h)
  return wino44_round_ok(wino44_items(d, mode));
}

long long wino44_workspace_bytes(const contrad_conv_desc* d) { return 36ll * d->C * d->K * (long long)sizeof(float); }

int wino44_grid(const wino44::Args& a) {
  const int l0 = cdiv(a.NP, 8) * a.NKB;           // items of the fullest XCD
  return 8 * std::min(WINO_CUS / 8, l0);
}

template <int MODE, int BOXW>
int launch_wino44_inst(const wino44::Args& a, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino44::wino44_kernel<MODE, BOXW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino44::LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((wino44::wino44_kernel<MODE, BOXW>), dim3(wino44_grid(a)), dim3(512), wino44::LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

template <int MODE, int BOXW>
int launch_wino44n_inst(const wino44::Args& a, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino44n::wino44n_kernel<MODE, BOXW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino44::LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((wino44n::wino44n_kernel<MODE, BOXW>), dim3(wino44_grid(a)), dim3(512), wino44::LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

template <int MODE>
int launch_wino44(const contrad_conv_desc* d, const float* in, const float* wp, const float* bias, const float* ref,
                  float* out, float slope, float gain, float* U, hipStream_t stream) {
  wino44::Args a = wino44_args(d, MODE);
  a.x = in; a.U = U; a.y = out; a.bias = bias; a.ref = ref; a.slope = slope; a.gain = gain;
  const int quads = (a.Cin / 4) * a.Cout;
  hipLaunchKernelGGL(wino44::wino44_filter_kernel<MODE>, dim3(cdiv(quads, 256)), dim3(256), 0, stream, wp, U, d->C, d->K, d->ldw);
  CONTRAD_CHECK_LAUNCH();
  if (a.n32)
    return a.BW == 34 ? launch_wino44n_inst<MODE, 34>(a, stream) : a.BW == 18 ? launch_wino44n_inst<MODE, 18>(a, stream)
           : a.BW == 10 ? launch_wino44n_inst<MODE, 10>(a, stream) : launch_wino44n_inst<MODE, 6>(a, stream);
  return a.BW == 34 ? launch_wino44_inst<MODE, 34>(a, stream) : a.BW == 18 ? launch_wino44_inst<MODE, 18>(a, stream) : launch_wino44_inst<MODE, 10>(a, stream);
}

// ---------------- Winograd F(2x2, 2x2) path (wino22.h): 4x4 stride-2 pad-1 layers, forward and data gradient ----------------
bool wino22_ok(const contrad_conv_desc* d, int mode) {
  if (mode != MODE_FWD && mode != MODE_DGRAD) return false;
  if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 1) return false;
  if ((d->H & 1) || (d->W & 1) || d->Ho * 2 != d->H || d->Wo * 2 != d->W) return false;
  auto grid_ok = [](int g) { return g == 4 || g == 8 || g == 16; };
  if (!grid_ok(d->Ho) || !grid_ok(d->Wo)) return false;
  const int cin = mode == MODE_FWD ? d->C : d->K, cout = mode == MODE_FWD ? d->K : d->C;
  const int ldi = mode == MODE_FWD ? d->ldx : d->ldy;
  if ((cin & (mode == MODE_FWD ? 7 : 15)) || (cout & 63) || (ldi & 3) || (d->ldw & 3)) return false;
  const long long nimg = wino22::TB / ((d->Ho / 2) * (d->Wo / 2));
  const long long lim = 1ll << 31;
  if (nimg * d->H * d->W * std::max(d->ldx, d->ldy) * 4 >= lim) return false;
  if (4ll * 9 * cin * cout * 4 >= lim) return false;
  return true;
}

wino22::Args wino22_args(const contrad_conv_desc* d, int mode) {
  wino22::Args a{};
  const bool dg = mode == MODE_DGRAD;
  a.N = d->N; a.dgrad = dg ? 1 : 0;
  a.Hi = dg ? d->Ho : d->H; a.Wi = dg ? d->Wo : d->W;
  a.Hout = dg ? d->H : d->Ho; a.Wout = dg ? d->W : d->Wo;
  a.Cin = dg ? d->K : d->C; a.Cout = dg ? d->C : d->K;
  a.ldi = dg ? d->ldy : d->ldx; a.ldo = dg ? d->ldx : d->ldy;
  a.GH = d->Ho; a.GW = d->Wo;
  const int tw = a.GW / 2, thw = (a.GH / 2) * tw;
  a.sh_tw = __builtin_ctz(tw); a.sh_thw = __builtin_ctz(thw);
  a.NIMG = wino22::TB / thw;
  a.NTB = cdiv(d->N, a.NIMG);
  a.NKB = a.Cout / 64;
  return a;
}

long long wino22_items(const contrad_conv_desc* d, int mode) {
  const wino22::Args a = wino22_args(d, mode);
  return (long long)a.NTB * a.NKB * (mode == MODE_DGRAD ? 4 : 1);
}

bool wino22_planned(const contrad_conv_desc* d, int mode) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO22"); return !(e && e[0] == '0'); }();
  if (!enabled || !wino22_ok(d, mode)) return false;
  const long long items = wino22_items(d, mode);
  const long long rounds = cdivll(items, WINO_CUS);
  // (1.78x fewer multiply-adds, not 2.25x: a last round that is a quarter empty already loses to the direct kernels -- forward of
  // 256 -> 512 channels at 1536 images, 384 items: 0.622 ms against 0.558, profiles/r06_ab_wino22_layers.txt)
  static const long long min_items = []() { const char* e = contrad_dev_env("CONTRAD_WINO22_MIN_ITEMS"); return e ? atoll(e) : 150ll; }();
  if (items < WINO_CUS) return items >= min_items;
  return rounds * WINO_CUS * 100 <= items * 125;
}

long long wino22_workspace_bytes(const contrad_conv_desc* d) { return 4ll * 9 * d->C * d->K * (long long)sizeof(float); }

template <int MODE, int NRAW>
int launch_wino22_inst(const wino22::Args& a, int blocks, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino22::wino22_kernel<MODE, NRAW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino22::LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((wino22::wino22_kernel<MODE, NRAW>), dim3(blocks), dim3(512), wino22::LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

template <int MODE>
int launch_wino22(const contrad_conv_desc* d, const float* in, const float* wp, const float* bias, const float* ref,
                  float* out, float slope, float gain, float* U, hipStream_t stream) {
  wino22::Args a = wino22_args(d, MODE);
  a.x = in; a.U = U; a.y = out; a.bias = bias; a.ref = ref; a.slope = slope; a.gain = gain;
  const int quads = 4 * (a.Cin / 4) * a.Cout;
  hipLaunchKernelGGL(wino22::wino22_filter_kernel<MODE>, dim3(cdiv(quads, 256)), dim3(256), 0, stream, wp, U, d->C, d->K, d->ldw);
  CONTRAD_CHECK_LAUNCH();
  const int l0 = cdiv(a.NTB, 8) * a.NKB * (MODE == MODE_DGRAD ? 4 : 1);
  const int blocks = 8 * std::min(WINO_CUS / 8, l0);
  const int items = 2 * a.NIMG * (a.GH + 1) * (a.GW + 1);      // raw box pieces per chunk (pixel x k-quad)
  const int nraw = cdiv(items, 256);
  if (nraw <= 5) return launch_wino22_inst<MODE, 5>(a, blocks, stream);
  if (nraw == 6) return launch_wino22_inst<MODE, 6>(a, blocks, stream);
  return launch_wino22_inst<MODE, 7>(a, blocks, stream);
}

// ---------------- F(2x2, 2x2) on the phases of the 3x3 stride-2 pad-0 layers (wino23.h): StyleGAN2's blurred conv2, forward ----------------
// (input (2 Ho + 1) x (2 Wo + 1), power-of-two output grids >= 4, input channels % 16, output channels % 64; 25 of the dense
// layer's 36 multiply-adds)
bool wino23_ok(const contrad_conv_desc* d, int mode) {
  if (mode != MODE_FWD) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 0) return false;
  if (d->H != 2 * d->Ho + 1 || d->W != 2 * d->Wo + 1) return false;
  if (d->Ho < 4 || d->Wo < 4 || (d->Ho & (d->Ho - 1)) || (d->Wo & (d->Wo - 1))) return false;
  if (d->Wo >= 32 ? d->Ho < 16 : d->Ho != d->Wo) return false;
  if ((d->C & 15) || (d->K & 63) || (d->ldx & 3) || (d->ldw & 3)) return false;
  const long long nimg = d->Wo >= 32 ? 1 : 128 / ((d->Ho / 2) * (d->Wo / 2));
  const long long lim = 1ll << 31;
  if (nimg * d->H * d->W * std::max(d->ldx, d->ldy) * 4 >= lim) return false;
  if (4ll * 9 * d->C * d->K * 4 >= lim) return false;
  return true;
}

wino23::Args wino23_args(const contrad_conv_desc* d) {
  wino23::Args a{};
  a.N = d->N; a.Hi = d->H; a.Wi = d->W; a.GH = d->Ho; a.GW = d->Wo;
  a.Cin = d->C; a.Cout = d->K; a.ldi = d->ldx; a.ldo = d->ldy;
  a.TW = d->Wo >= 32 ? 16 : d->Wo / 2; a.TH = d->Wo >= 32 ? 8 : d->Ho / 2;      // patches of 8 x 16 tiles, or whole images
  a.sh_tw = __builtin_ctz(a.TW); a.sh_thw = __builtin_ctz(a.TH * a.TW);
  a.NIMG = wino23::TB / (a.TH * a.TW);
  a.PH = d->Ho / (2 * a.TH); a.PW = d->Wo / (2 * a.TW);
  a.NP = cdiv(d->N, a.NIMG) * a.PH * a.PW;
  a.NKB = a.Cout / 64;
  return a;
}

bool wino23_planned(const contrad_conv_desc* d, int mode) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO23"); return !(e && e[0] == '0'); }();
  if (!enabled || !wino23_ok(d, mode)) return false;
  const wino23::Args a = wino23_args(d);
  const long long items = (long long)a.NP * a.NKB;
  if (items < WINO_CUS) return items >= 230;
  return cdivll(items, WINO_CUS) * WINO_CUS * 10 <= items * 14;
}

long long wino23_workspace_bytes(const contrad_conv_desc* d) { return 4ll * 9 * d->C * d->K * (long long)sizeof(float); }

int wino23_grid(const wino23::Args& a) { return 8 * std::min(WINO_CUS / 8, cdiv(a.NP, 8) * a.NKB); }

template <int NRAW>
int launch_wino23_inst(const wino23::Args& a, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino23::wino23_kernel<NRAW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino23::LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((wino23::wino23_kernel<NRAW>), dim3(wino23_grid(a)), dim3(512), wino23::LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

int launch_wino23(const contrad_conv_desc* d, const float* in, const float* wp, const float* bias, const float* ref,
                  float* out, float slope, float gain, float* U, hipStream_t stream) {
  wino23::Args a = wino23_args(d);
  a.x = in; a.U = U; a.y = out; a.bias = bias; a.ref = ref; a.slope = slope; a.gain = gain;
  const int quads = 4 * (a.Cin / 4) * a.Cout;
  hipLaunchKernelGGL(wino23::wino23_filter_kernel, dim3(cdiv(quads, 256)), dim3(256), 0, stream, wp, U, d->C, d->K, d->ldw);
  CONTRAD_CHECK_LAUNCH();
  const int nraw = cdiv(2 * a.NIMG * (2 * a.TH + 1) * (2 * a.TW + 1), 256);
  if (nraw <= 5) return launch_wino23_inst<5>(a, stream);
  if (nraw == 6) return launch_wino23_inst<6>(a, stream);
  return launch_wino23_inst<7>(a, stream);
}

// ---- weight gradient of the 4x4 stride-2 layers on F(2x2, 2x2) (wino22_wgrad_kernel) ----
bool wino22_wgrad_ok(const contrad_conv_desc* d) {
  if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 1) return false;
  if (d->Ho * 2 != d->H || d->Wo * 2 != d->W) return false;
  auto grid_ok = [](int g) { return g == 4 || g == 8 || g == 16; };
  if (!grid_ok(d->Ho) || !grid_ok(d->Wo)) return false;
  if ((d->C != 64 && (d->C & 127)) || (d->K & 63) || (d->ldx & 3) || (d->ldy & 3)) return false;
  if (2ll * d->H * d->W * std::max(d->ldx, d->ldy) * 4 >= (1ll << 31)) return false;
  return true;
}

wino22::WArgs wino22_wgrad_args(const contrad_conv_desc* d) {
  wino22::WArgs a{};
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.ldx = d->ldx; a.ldy = d->ldy; a.GH = d->Ho; a.GW = d->Wo;
  a.CTW = std::min(4, a.GW / 2);
  a.CTH = std::min(8 / a.CTW, a.GH / 2);
  a.CNIMG = 8 / (a.CTH * a.CTW);
  a.sh_ctw = __builtin_ctz(a.CTW); a.sh_cthw = __builtin_ctz(a.CTH * a.CTW);
  a.QH = a.GH / (2 * a.CTH); a.QW = a.GW / (2 * a.CTW);
  a.Q = cdiv(d->N, a.CNIMG) * a.QH * a.QW;
  a.RBN = 4 * d->C / 128; a.KB = d->K / 64;
  a.CPB = std::min(d->C, 128); a.sh_cpb = __builtin_ctz(a.CPB);
  const int splits = std::max(1, std::min(a.Q, WINO_CUS / (a.RBN * a.KB)));
  a.qps = cdiv(a.Q, splits);
  return a;
}
int wino22_wgrad_splits(const wino22::WArgs& a) { return cdiv(a.Q, a.qps); }

bool wino22_wgrad_planned(const contrad_conv_desc* d) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO22_WGRAD"); return !(e && e[0] == '0'); }();
  static const bool enabled2 = []() { const char* e = contrad_dev_env("CONTRAD_WINO22"); return !(e && e[0] == '0'); }();
  if (!enabled || !enabled2 || !wino22_wgrad_ok(d)) return false;
  const wino22::WArgs a = wino22_wgrad_args(d);
  const long long blocks = (long long)a.RBN * a.KB * wino22_wgrad_splits(a);
  static const int min_qps = []() { const char* e = contrad_dev_env("CONTRAD_WINO_MIN_QPS"); return e ? atoi(e) : 16; }();
  return a.qps >= min_qps && blocks * 10 >= WINO_CUS * 7 && blocks <= WINO_CUS;
}

long long wino22_wgrad_workspace_bytes(const contrad_conv_desc* d) {
  const wino22::WArgs a = wino22_wgrad_args(d);
  return (long long)wino22_wgrad_splits(a) * (16ll * d->C + 1) * d->K * (long long)sizeof(float);
}

// ---- weight gradient on the Winograd kernel (wino_wgrad_kernel): C % 64, K % 64 ----
bool wino_wgrad_ok(const contrad_conv_desc* d) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
  if (d->H < 4 || d->W < 4 || (d->H & (d->H - 1)) || (d->W & (d->W - 1))) return false;
  if ((d->C & 63) || (d->K & 63) || (d->ldx & 3) || (d->ldy & 3)) return false;
  const long long lim = 1ll << 31;
  if (2ll * d->H * d->W * std::max(d->ldx, d->ldy) * 4 >= lim) return false;     // chunk-relative byte offsets (<= 2 images)
  return true;
}

wino::WArgs wino_wgrad_args(const contrad_conv_desc* d) {
  wino::WArgs a{};
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.ldx = d->ldx; a.ldy = d->ldy;
  a.CTW = std::min(4, d->W / 2);
  a.CTH = std::min(8 / a.CTW, d->H / 2);
  a.CNIMG = 8 / (a.CTH * a.CTW);
  a.sh_ctw = __builtin_ctz(a.CTW); a.sh_cthw = __builtin_ctz(a.CTH * a.CTW);
  a.QH = d->H / (2 * a.CTH); a.QW = d->W / (2 * a.CTW);
  a.Q = cdiv(d->N, a.CNIMG) * a.QH * a.QW;
  a.CB = d->C / 64; a.KB = d->K / 64;
  const int splits = std::max(1, std::min(a.Q, WINO_CUS / (a.CB * a.KB)));
  a.qps = cdiv(a.Q, splits);
  a.BH = a.QH > 1 ? 2 * a.CTH + 2 : d->H; a.r_org = a.QH > 1 ? -1 : 0;
  a.BW = a.QW > 1 ? 2 * a.CTW + 2 : d->W; a.c_org = a.QW > 1 ? -1 : 0;
  return a;
}
int wino_wgrad_splits(const wino::WArgs& a) { return cdiv(a.Q, a.qps); }

// planned when every block gets a contraction long enough to pay for its prologue and its 4x4 -> 3x3 epilogue
bool wino_wgrad_planned(const contrad_conv_desc* d) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WINO_WGRAD"); return !(e && e[0] == '0'); }();
  static const bool enabled2 = []() { const char* e = contrad_dev_env("CONTRAD_WINO"); return !(e && e[0] == '0'); }();
  if (!enabled || !enabled2 || !wino_wgrad_ok(d)) return false;
  const wino::WArgs a = wino_wgrad_args(d);
  const long long blocks = (long long)a.CB * a.KB * wino_wgrad_splits(a);
  static const int min_qps = []() { const char* e = contrad_dev_env("CONTRAD_WINO_MIN_QPS"); return e ? atoi(e) : 16; }();
  return a.qps >= min_qps && blocks * 10 >= WINO_CUS * 7 && blocks <= WINO_CUS;
}

long long wino_wgrad_workspace_bytes(const contrad_conv_desc* d) {
  const wino::WArgs a = wino_wgrad_args(d);
  return (long long)wino_wgrad_splits(a) * (9ll * d->C + 1) * d->K * (long long)sizeof(float);
}

template <int MODE>
int launch_wino(const contrad_conv_desc* d, const float* in, const float* wp, const float* bias, const float* ref,
                float* out, float slope, float gain, float* U, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino::wino_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino::LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  wino::Args a = wino_args(d, MODE);
  a.x = in; a.U = U; a.y = out; a.bias = bias; a.ref = ref; a.slope = slope; a.gain = gain;
  const int quads = (a.Cin / 4) * a.Cout;
  hipLaunchKernelGGL(wino::wino_filter_kernel<MODE>, dim3(cdiv(quads, 256)), dim3(256), 0, stream, wp, U, d->C, d->K, d->ldw);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(wino::wino_kernel<MODE>, dim3(wino_grid(a)), dim3(512), wino::LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

template <int MODE, int BM, int BN, bool BAL>
int launch_lean_inst(const IgemmArgs& a, dim3 grid, hipStream_t stream) {
  static bool attr_set_dev[64] = {};  // per device (one process may drive several); benign race: idempotent
  constexpr size_t smem = lean_smem_bytes<MODE, BM, BN>();
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_lean_kernel<MODE, BM, BN, BAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  IgemmArgs b = a;
  b.ny = (int)grid.y;
  hipLaunchKernelGGL((igemm_lean_kernel<MODE, BM, BN, BAL>), dim3(grid.x * grid.y), dim3(NTHREADS), smem, stream, b);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

template <int MODE, int BM, int BN>
int launch_lean(const IgemmArgs& a, dim3 grid, hipStream_t stream) {
  if constexpr (MODE == MODE_DGRAD) {
    if (a.cbal > 0) return launch_lean_inst<MODE, BM, BN, true>(a, grid, stream);   // balanced strided order (igemm_lean.h)
  }
  return launch_lean_inst<MODE, BM, BN, false>(a, grid, stream);
}

// Does the shape fit the lean loop (igemm_lean.h)?  pps = WGRAD position tiles per split.
bool lean_ok(const contrad_conv_desc* d, int mode, long long pps) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_IGEMM_LEAN"); return !(e && e[0] == '0'); }();
  if (!enabled || BK != 16) return false;
  const long long lim = 1ll << 31;
  if (d->KH * d->KW > 32) return false;
  if ((long long)d->KH * d->KW * d->C * d->ldw * 4 >= lim) return false;   // packed weight addressed with byte offsets
  const long long img_x = (long long)d->H * d->W * d->ldx * 4, img_y = (long long)d->Ho * d->Wo * d->ldy * 4;
  if ((long long)d->ldy * 4 * 128 >= (1ll << 30) || (long long)d->K * 4 * 128 >= (1ll << 30)) return false;   // epilogue offsets
  if (mode == MODE_FWD) {
    if ((d->C & 15) || (d->ldx & 3) || (d->K & 3)) return false;
    const long long imgs = 128 / ((long long)d->Ho * d->Wo) + 2;            // images one 128-row M-tile can touch
    return imgs * img_x < lim;
  }
  if (mode == MODE_DGRAD) {
    if ((d->K & 15) || (d->ldy & 3)) return false;
    const long long imgs = 128 / ((long long)cdiv(d->H, d->stride) * cdiv(d->W, d->stride)) + 2;
    return imgs * img_y < lim && imgs * img_x < (1ll << 30);   // (dx is addressed relative to the tile's first image)
  }
  if ((d->C & 3) || (d->ldx & 3) || (d->K & 3) || (d->ldy & 3)) return false;
  const long long P = (long long)d->N * d->Ho * d->Wo;
  if (P % 16) return false;
  const int gw = d->Wo < 16 ? d->Wo : 16;
  if (gw <= 0 || 16 % gw || d->Wo % gw) return false;
  const int gh = d->Ho < 16 / gw ? d->Ho : 16 / gw;
  if ((16 / gw) % gh || d->Ho % gh) return false;
  if (d->pad > gh * d->stride || d->pad > gw * d->stride) return false;     // interior patches must never touch padding
  const long long imgs = pps * 16 / ((long long)d->Ho * d->Wo) + 2;         // images one split walks over
  return imgs * img_x < lim && pps * 16 * d->ldy * 4 < lim;
}

// float4-addressable operands?  (the !VEC kernels are instantiated for the 64x64 tile only)
bool vec_ok(const contrad_conv_desc* d, int mode) {
  const bool c4 = (d->C & 3) == 0 && (d->ldx & 3) == 0;
  const bool k4 = (d->K & 3) == 0 && (d->ldy & 3) == 0;
  if (mode == MODE_FWD) return c4 && (d->K & 3) == 0;   // A: x rows; B: packed weight rows (ldw % 4 checked)
  if (mode == MODE_DGRAD) return k4;                    // A: gy rows; B: packed weight rows along cout
  return c4 && k4;                                      // WGRAD: x rows and gy rows
}

// Dev library only (tools/tune_plans.py): the next plans take this tile / split count instead of the model's (0 = model).
struct PlanOverride { int bm, bn, splits; };
PlanOverride g_plan_override = {0, 0, 0};
#if defined(CONTRAD_DEV_SWITCHES)
}  // namespace
extern "C" void contrad_dev_plan_override(int bm, int bn, int splits) { g_plan_override = PlanOverride{bm, bn, splits}; }
namespace {
#endif

// Tile choice: the biggest tile that still yields >= 3 blocks per CU (4 are resident); measured on the lean
// kernels at 3N = 192 / 384 / 1536 images (tools/ab_tile.sh): below that fill the 64x64 tile (TM = TN = 1) wins even
// though it does 4x the LDS traffic per flop.  mult4 / 4 = independent grids of this size (DGRAD parity classes: 4; in the
// balanced order of a 3x3 stride-2 layer 9 / 4 -- its blocks walk 1 / 2 / 2 / 4 M-tiles of the four classes).
void pick_tile(long long M, int Ncol, bool vec, bool lean, int mult4, int* bm, int* bn) {   // mult4 = 4 x independent grids
  if (!vec) { *bm = 64; *bn = 64; return; }
  // lean only: 4 x 1 waves, no MFMA columns wasted.  (A 256 x 32 tile -- two MFMA tiles per wave sharing one B fragment --
  // was tried in round 2: 89.0 vs 88.3 TF/s, because all per-row work doubles with it.  What holds these K = 288 layers
  // back is the instruction overhead per block -- prologue, epilogue and loop control around only 144 MFMAs per wave --
  // plus the loop's memory instructions: ablation table in DESIGN.md section 7, tools/dev/ablate32.sh.)
  if (lean && Ncol <= 32) { *bm = 128; *bn = 32; return; }
  static const int forced = []() { const char* e = contrad_dev_env("CONTRAD_IGEMM_TILE"); return e ? atoi(e) : 0; }();  // dev: "128064"
  if (forced) { *bm = forced / 1000; *bn = forced % 1000; return; }
  if (g_plan_override.bm > 0) { *bm = g_plan_override.bm; *bn = g_plan_override.bn; return; }
  static const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  for (int i = (Ncol > 64 ? 0 : 2); i < 4; ++i) {
    *bm = cand[i][0]; *bn = cand[i][1];
    if (cdivll(M, *bm) * cdivll(Ncol, *bn) * mult4 >= 768 * 4) return;
  }
}

template <int MODE>
int dispatch(const IgemmArgs& a, int bm, int bn, bool vec, dim3 grid, hipStream_t s) {
  if (vec && lean_ok(&a.d, MODE, a.ptiles_per_split)) {
    if (bn == 32) return launch_lean<MODE, 128, 32>(a, grid, s);
    if (bm == 128 && bn == 128) return launch_lean<MODE, 128, 128>(a, grid, s);
    if (bm == 128 && bn == 64) return launch_lean<MODE, 128, 64>(a, grid, s);
    if (bm == 64 && bn == 128) return launch_lean<MODE, 64, 128>(a, grid, s);
    return launch_lean<MODE, 64, 64>(a, grid, s);
  }
  if (!vec) return launch<MODE, 64, 64, false>(a, grid, s);
  if (bm == 128 && bn == 128) return launch<MODE, 128, 128, true>(a, grid, s);
  if (bm == 128 && bn == 64) return launch<MODE, 128, 64, true>(a, grid, s);
  if (bm == 64 && bn == 128) return launch<MODE, 64, 128, true>(a, grid, s);
  return launch<MODE, 64, 64, true>(a, grid, s);
}

// The float4 / buffer-load paths need 16-byte aligned operands (torch allocations and channel slices at multiples of
// 4 floats are); anything else is a caller error, not a silent slow path.
bool aligned16(const void* a, const void* b, const void* c) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

int check_desc(const contrad_conv_desc* d) {
  CONTRAD_ARG(d != nullptr);
  CONTRAD_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0);
  CONTRAD_ARG(d->KH > 0 && d->KW > 0 && (d->stride == 1 || d->stride == 2) && d->pad >= 0);
  CONTRAD_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1);
  CONTRAD_ARG(d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1);
  CONTRAD_ARG(d->ldx >= d->C && d->ldy >= d->K && d->ldw >= d->K);
  CONTRAD_ARG((d->ldw & 3) == 0);
  // the kernels address elements with 32-bit offsets
  CONTRAD_ARG((long long)d->N * d->H * d->W * d->ldx < (1ll << 31));
  CONTRAD_ARG((long long)d->N * d->Ho * d->Wo * d->ldy < (1ll << 31));
  CONTRAD_ARG((long long)d->KH * d->KW * d->C * d->ldw < (1ll << 31));
  if ((d->C & 3) == 0) CONTRAD_ARG((d->ldx & 3) == 0);
  if ((d->K & 3) == 0) CONTRAD_ARG((d->ldy & 3) == 0);
  return 0;
}

int wgrad_plan(const contrad_conv_desc* d, int* bm, int* bn, int* tiles_m, int* tiles_n, int* splits,
               int* ptiles_per_split) {
  const int Kg = d->KH * d->KW * d->C;
  const long long P = (long long)d->N * d->Ho * d->Wo;
  *bn = (d->K > 64) ? 128 : 64;
  *bm = (Kg > 64) ? 128 : 64;
  static const int forced = []() { const char* e = contrad_dev_env("CONTRAD_WGRAD_TILE"); return e ? atoi(e) : 0; }();  // dev
  if (forced) { *bm = forced / 1000; *bn = forced % 1000; }
  if (g_plan_override.bm > 0) { *bm = g_plan_override.bm; *bn = g_plan_override.bn; }
  static const int target = []() { const char* e = contrad_dev_env("CONTRAD_WGRAD_BLOCKS"); return e ? atoi(e) : 1024; }();  // dev
  if (!vec_ok(d, MODE_WGRAD)) { *bm = 64; *bn = 64; }
  *tiles_m = cdiv(Kg, *bm);
  *tiles_n = cdiv(d->K, *bn);
  const long long ptiles = cdivll(P, BK);
  // 2 blocks are resident per CU (LDS): aim at <= 1024 blocks = two full rounds of the 256 CUs, never a ragged
  // third one (9 x 114 = 1026 blocks cost +30 % on the 3x3 layers before this was a floor)
  long long want = target / ((long long)(*tiles_m) * (*tiles_n));
  if (g_plan_override.splits > 0) want = g_plan_override.splits;   // dev: split count from the tuner
  if (want < 1) want = 1;
  long long pps = cdivll(ptiles, want);
  if (pps < 4) pps = 4;  // at least 128 positions per split
  if (pps > ptiles) pps = ptiles;
  *ptiles_per_split = (int)pps;
  *splits = (int)cdivll(ptiles, pps);
  if (d->K <= 32 && *bm == 128 && vec_ok(d, MODE_WGRAD) && lean_ok(d, MODE_WGRAD, pps)) *bn = 32;   // (tiles_n stays 1)
  return 0;
}

// Single-output 1x1 layer (the 512 -> 1 logit of the discriminator heads): y[m] = gain * lrelu(x[m] . w + bias) [+ addend].
// One wave per row, lanes stride over the channels, fixed-order butterfly sum.  On the tiled GEMM kernel this was ONE
// column of a 64-wide tile walked by 24 blocks: 44 us for 1.6 MFLOP (1.2 % of a per-rank-batch-64 step).
__global__ __launch_bounds__(256) void fwd_k1_kernel(const float* __restrict__ x, const float* __restrict__ w, long long M,
                                                     int C, int ldx, int ldw, const float* __restrict__ bias, float slope,
                                                     float gain, float* __restrict__ y, int ldy,
                                                     const float* __restrict__ addend) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + m * ldx;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc += xr[c] * w[(long long)c * ldw];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    float v = acc + (bias ? bias[0] : 0.f);
    v = (v > 0.f) ? v : v * slope;
    v *= gain;
    if (addend) v += addend[m * ldy];
    y[m * ldy] = v;
  }
}

// y[m][c] = gain * lrelu(sum_s ws[s][m][c] + bias[c])   (fixed summation order)
__global__ void fwd_reduce_kernel(const float* __restrict__ ws, int splits, long long M, int Ncol,
                                  const float* __restrict__ bias, float slope, float gain, float* __restrict__ y,
                                  int ldy, const float* __restrict__ addend) {
  const long long total = M * Ncol;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int k = 0; k < splits; ++k) v += ws[(long long)k * total + e];
    const long long m = e / Ncol;
    const int c = (int)(e - m * Ncol);
    if (bias) v += bias[c];
    v = (v > 0.f) ? v : v * slope;
    v *= gain;
    if (addend) v += addend[m * ldy + c];
    y[m * ldy + c] = v;
  }
}

// FWD plan: tile + split-K count from a small cost model (measured rates of the lean tiles, 768 blocks = a full
// chip, partial slabs priced at 4 TB/s).  Split-K only pays for small-M, deep-K GEMMs: the merged head layer
// (M = 3N rows, K = 8192) and the last conv layers at small per-rank batches.
struct FwdPlan { int bm, bn, splits, tps, pixmajor, balanced; };   // balanced: strided DGRAD, equal-work block order

// tile + split count for a lean GEMM of M x Ncol outputs over t_total K-tiles (shared by FWD and stride-1 DGRAD)
void split_plan(long long M, int Ncol, int t_total, double flops, FwdPlan* p) {
  static const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  static const double rate[4] = {135e12, 128e12, 128e12, 115e12};
  static const int sp[8] = {1, 2, 3, 4, 6, 8, 12, 16};
  if (g_plan_override.bm > 0) {   // dev: tile and split count from the tuner
    const int s_ = std::max(1, g_plan_override.splits);
    p->bm = g_plan_override.bm; p->bn = g_plan_override.bn;
    p->tps = cdiv(t_total, s_);
    p->splits = cdiv(t_total, p->tps);
    return;
  }
  double best = 1e30;
  for (int i = (Ncol > 64 ? 0 : 2); i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      const int s_ = sp[j];
      const int tps = cdiv(t_total, s_);
      if (s_ > 1 && tps < 8) break;
      const int splits = cdiv(t_total, tps);
      const long long tm = cdivll(M, cand[i][0]), tn = cdivll(Ncol, cand[i][1]);
      const double blocks = (double)(tm * tn) * splits;
      const double waste = (double)(tm * cand[i][0]) * (double)(tn * cand[i][1]) / ((double)M * Ncol);
      const double fill = blocks >= 768.0 ? 1.0 : blocks / 768.0;
      double t = flops * waste / (rate[i] * fill);
      if (splits > 1) t += (2.0 * splits + 1.0) * (double)M * Ncol * 4.0 / 4e12 + 4e-6;   // slabs + one more launch
      if (t < best) { best = t; p->bm = cand[i][0]; p->bn = cand[i][1]; p->splits = splits; p->tps = tps; }
    }
}

// 1x1 layer on 1x1 "images" with a single output channel: fwd_k1_kernel
bool fwd_k1_ok(const contrad_conv_desc* d) {
  return d->K == 1 && d->KH == 1 && d->KW == 1 && d->H == 1 && d->W == 1 && d->stride == 1 && d->pad == 0;
}

bool splitk_enabled() {
  static const bool on = []() { const char* e = contrad_dev_env("CONTRAD_IGEMM_SPLITK"); return !(e && e[0] == '0'); }();
  return on;
}

// Pixel-major M-tiles (igemm_lean.h): worth it on small maps, where a large share of the tap-positions is padding, and
// possible when a tile's worth of images exists.  frac = share of tap-positions that are NOT padding.
// Measured at 1536 images (tools/bench_conv.py, forward): 3x3 on 4x4 maps (0.69 of the tap-positions valid) -11 %, 4x4
// stride-2 onto 4x4 (0.77) -8.5 %; 3x3 on 8x8 (0.84) +-0, 4x4 stride-2 onto 8x8 (0.88) +8 % -- a pixel-major tile reads every
// input element exactly once (nothing is shared between the rows of a tile any more, the reuse between neighbouring
// pixels moves from the L1 to the L2), which costs about as much as 15 % of the MFMAs.
constexpr double PIXMAJOR_MAX_VALID = 0.80;
// WGRAD streams its operands either way (no reuse inside a K-tile to lose), so it pays earlier: 3x3 on 4x4 maps -21 %, 4x4
// stride-2 onto 4x4 -10 %, 3x3 on 8x8 (0.84 valid) -8 %; 4x4 stride-2 onto 8x8 (0.88) +3 %, 3x3 on 16x16 (0.92) +-0.
constexpr double PIXMAJOR_WGRAD_MAX_VALID = 0.85;

bool pixmajor_enabled() {
  static const bool on = []() { const char* e = contrad_dev_env("CONTRAD_PIXMAJOR"); return !(e && e[0] == '0'); }();
  return on;
}

double fwd_valid_tap_fraction(const contrad_conv_desc* d) {
  long long vh = 0, vw = 0;
  for (int ho = 0; ho < d->Ho; ++ho)
    for (int kh = 0; kh < d->KH; ++kh) vh += (unsigned)(ho * d->stride - d->pad + kh) < (unsigned)d->H;
  for (int wo = 0; wo < d->Wo; ++wo)
    for (int kw = 0; kw < d->KW; ++kw) vw += (unsigned)(wo * d->stride - d->pad + kw) < (unsigned)d->W;
  return (double)(vh * vw) / ((double)d->Ho * d->Wo * d->KH * d->KW);
}

// When the automatic plan takes border classes (igemm_lean.h, p.nwin) instead of pixel-major or image-major tiles.
// Measured at 1536 images (CONTRAD_TILEMODE=0/1/2 tools/bench_conv.py; image-major / pixel-major / border classes, ms):
//   3x3 on 4x4 (0.69 valid)        fwd 0.805 / 0.710 / 0.730   dgrad 0.807 / 0.709 / 0.729
//   4x4 s2 onto 4x4 (0.77)         fwd 0.718 / 0.649 / 0.720
//   3x3 on 8x8 (0.84)              fwd 0.805 / 0.805 / 0.726   dgrad 0.808 / 0.803 / 0.724
//   4x4 s2 onto 8x8 (0.88)         fwd 0.723 / 0.772 / 0.693
//   3x3 on 16x16 (0.92)            fwd 0.818 / 0.782 / 0.786   dgrad 0.808 / 0.823 / 0.775
//   4x4 s2 onto 16x16 (0.94)       fwd 0.925 / 0.919 / 0.897
// -> pixel-major where at most 0.80 of the tap-positions are valid (its tiles are single pixels there anyway), border
// classes from there up to 0.96 (they keep the reuse of neighbouring pixels inside a tile, which is what pixel-major
// tiles lose on the larger maps).
constexpr double BORDER_MAX_VALID = 0.96;
bool border_classes_ok(const contrad_conv_desc* d, int mode, int bm);
bool fwd_pixmajor_ok(const contrad_conv_desc* d, int bm);
bool dgrad_pixmajor_ok(const contrad_conv_desc* d, int bm);
bool fwd_border_ok(const contrad_conv_desc* d, int bm, double valid) {
  return valid <= BORDER_MAX_VALID && !fwd_pixmajor_ok(d, bm) && border_classes_ok(d, MODE_FWD, bm);
}
bool dgrad_border_ok(const contrad_conv_desc* d, int bm, double valid) {
  return valid <= BORDER_MAX_VALID && !dgrad_pixmajor_ok(d, bm) && border_classes_ok(d, MODE_DGRAD, bm);
}

bool fwd_pixmajor_ok(const contrad_conv_desc* d, int bm) {
  if (!pixmajor_enabled() || d->Ho * d->Wo > 256 || d->N < bm) return false;
  if ((long long)bm * d->H * d->W * d->ldx * 4 >= (1ll << 30)) return false;      // row offsets inside a tile (bytes)
  if ((long long)bm * d->Ho * d->Wo * d->ldy * 4 >= (1ll << 31)) return false;    // the epilogue's descriptor range
  return fwd_valid_tap_fraction(d) <= PIXMAJOR_MAX_VALID;
}

double dgrad_valid_tap_fraction(const contrad_conv_desc* d) {
  const int s = d->stride;
  long long valid = 0, all = 0;
  for (int h = 0; h < d->H; ++h) {
    long long vh = 0, ah = 0;
    for (int kh = (h + d->pad) % s; kh < d->KH; kh += s) { ++ah; vh += (unsigned)((h + d->pad - kh) / s) < (unsigned)d->Ho && h + d->pad - kh >= 0; }
    for (int w = 0; w < d->W; ++w) {
      long long vw = 0, aw = 0;
      for (int kw = (w + d->pad) % s; kw < d->KW; kw += s) { ++aw; vw += (unsigned)((w + d->pad - kw) / s) < (unsigned)d->Wo && w + d->pad - kw >= 0; }
      valid += vh * vw;
      all += ah * aw;
    }
  }
  return all ? (double)valid / (double)all : 1.0;
}

bool pixel_order_full(const int* taps, int npix, int nib, int tiles_n, unsigned char* out, int nruns);

// Pixel-major tiles inside the parity classes of a STRIDED data gradient, with the slot-balanced order (pixel_order_full):
// the launch is class-major (every class = 2 of the 8 XCD runs) and all four classes use class (0,0)'s table with the pixel
// mirrored along their odd axes -- possible when the classes are mirror images of one another (even maps, e.g. the 4x4
// stride-2 pad-1 layers).  Fills taps[] (valid taps per pixel of class (0,0)) and returns its pixel count, or 0.
// Without the balanced order these tiles LOST (round 3: 0.714 -> 0.757 ms on the 8x8 -> 4x4 layer): a class tile contracts
// over 1, 2 or 4 taps and the slots' sums were far apart.
int dgrad_strided_class0_taps(const contrad_conv_desc* d, int* taps) {
  const int s = d->stride;
  if (s != 2 || (d->H & 1) || (d->W & 1)) return 0;
  const int Hc = d->H / 2, Wc = d->W / 2;
  if (Hc * Wc > 256 || Hc > 128 || Wc > 128) return 0;      // (the per-axis bound makes wh / ww safe on their own)
  int wh[2][128], ww[2][128];
  for (int par = 0; par < 2; ++par) {
    for (int hq = 0; hq < Hc; ++hq) {
      int v = 0;
      for (int kh = (par + d->pad) % s; kh < d->KH; kh += s) {
        const int num = hq * s + par + d->pad - kh;
        v += num >= 0 && num / s < d->Ho;
      }
      wh[par][hq] = v;
    }
    for (int wq = 0; wq < Wc; ++wq) {
      int v = 0;
      for (int kw = (par + d->pad) % s; kw < d->KW; kw += s) {
        const int num = wq * s + par + d->pad - kw;
        v += num >= 0 && num / s < d->Wo;
      }
      ww[par][wq] = v;
    }
  }
  for (int hq = 0; hq < Hc; ++hq) if (wh[1][hq] != wh[0][Hc - 1 - hq]) return 0;      // odd class = mirror of the even one
  for (int wq = 0; wq < Wc; ++wq) if (ww[1][wq] != ww[0][Wc - 1 - wq]) return 0;
  for (int hq = 0; hq < Hc; ++hq)
    for (int wq = 0; wq < Wc; ++wq) taps[hq * Wc + wq] = wh[0][hq] * ww[0][wq];
  return Hc * Wc;
}

bool dgrad_strided_full_ok(const contrad_conv_desc* d, int bm, int bn) {
  int taps[256];
  const int npix = dgrad_strided_class0_taps(d, taps);
  if (!npix) return false;
  const int nib = cdiv(d->N, bm), tiles_n = cdiv(d->C, bn);
  if (((long long)nib * npix * tiles_n) % 2) return false;          // a class = exactly two XCD runs
  unsigned char tmp[256];
  return pixel_order_full(taps, npix, nib, tiles_n, tmp, 2);
}

bool dgrad_pixmajor_ok(const contrad_conv_desc* d, int bm) {
  // stride 1 only: the kernel walks pixel-major tiles inside the parity classes of a strided layer just as well (parity
  // holds, CONTRAD_PIXMAJOR_STRIDED=1), but a class tile then contracts over 1 .. 4 taps only and the 4x4 stride-2
  // layer onto 4x4 maps ran 0.714 -> 0.757 ms at 1536 images
  static const bool strided = []() { const char* e = contrad_dev_env("CONTRAD_PIXMAJOR_STRIDED"); return e && e[0] == '1'; }();
  if (!pixmajor_enabled() || d->H * d->W > 256 || d->N < bm) return false;
  if (d->stride != 1 && !strided) {
    int bm2, bn2;           // the tile the plan takes for this layer (the balanced order depends on its N-tile count)
    pick_tile((long long)d->N * cdiv(d->H, d->stride) * cdiv(d->W, d->stride), d->C, true, true, 4 * d->stride * d->stride, &bm2, &bn2);
    if (bm2 != bm || !dgrad_strided_full_ok(d, bm, bn2)) return false;
  }
  if ((long long)bm * d->Ho * d->Wo * d->ldy * 4 >= (1ll << 30)) return false;    // row offsets inside a tile (bytes)
  if ((long long)bm * d->H * d->W * d->ldx * 4 >= (1ll << 30)) return false;      // the epilogue's row offsets
  return dgrad_valid_tap_fraction(d) <= PIXMAJOR_MAX_VALID;
}

// WGRAD on pixel-major positions (igemm_lean.h): a K-tile is 16 images at one output pixel, and a row tile that lies
// inside one filter tap skips the K-tiles whose pixel is padding for that tap.  Same threshold as FWD / DGRAD.
bool wgrad_pixmajor_ok(const contrad_conv_desc* d, int bm, long long pps) {
  if (!pixmajor_enabled() || !vec_ok(d, MODE_WGRAD) || !lean_ok(d, MODE_WGRAD, pps)) return false;
  if (bm != 128 || (d->C % 128) != 0 || (d->N % 16) != 0 || d->Ho < 2 || d->Wo < 2 || d->Ho * d->Wo > 256) return false;
  if (d->pad > d->stride) return false;                                  // interior pixels must never touch padding
  static const double lim = []() { const char* e = contrad_dev_env("CONTRAD_PIXMAJOR_WGRAD_MAX"); return e ? atof(e) : PIXMAJOR_WGRAD_MAX_VALID; }();  // dev
  return fwd_valid_tap_fraction(d) <= lim;
}

// Order in which the pixels of a map are handed to consecutive pixel-major tiles.  The tiles are unequal (a corner pixel
// of a 3x3 pad-1 layer contracts over 4 taps, an edge pixel over 6, an interior one over 9) and all of them are resident
// at once (768 blocks on 1024 slots), so a CU's time is the SUM of the 3 blocks it happens to get.
//   0  row-major (as stored)     1  heaviest first (default)     2  heavy / light alternating
// Measured (3x3 on 4x4 maps, 1536 images): 0.719 / 0.709 / 0.710 ms forward, 0.716 / 0.708 / 0.707 data gradient -- the
// imbalance is NOT what keeps these tiles at 0.7 of the image-major tiles' issue rate (their operand traffic is, DESIGN.md
// section 3).  taps[p] = valid taps of pixel p (any positive weights).
void pixel_order(const int* taps, int npix, unsigned char* out) {
  static const int mode = []() { const char* e = contrad_dev_env("CONTRAD_PIXORDER"); return e ? atoi(e) : 1; }();
  int idx[256];
  for (int i = 0; i < npix; ++i) idx[i] = i;
  if (mode >= 1) {   // stable sort by taps, descending (insertion sort: npix <= 256, host, once per call)
    for (int i = 1; i < npix; ++i) {
      const int v = idx[i];
      int j = i - 1;
      while (j >= 0 && taps[idx[j]] < taps[v]) { idx[j + 1] = idx[j]; --j; }
      idx[j + 1] = v;
    }
  }
  if (mode == 2) {   // heaviest, lightest, 2nd heaviest, 2nd lightest, ...
    int tmp[256];
    for (int i = 0, lo = 0, hi = npix - 1; i < npix; ++i) tmp[i] = (i & 1) ? idx[hi--] : idx[lo++];
    for (int i = 0; i < npix; ++i) idx[i] = tmp[i];
  }
  for (int i = 0; i < npix; ++i) out[i] = (unsigned char)idx[i];
}

// Order of ALL pixel-major tiles of a launch (<= 256 of them), for launches that sit on the chip in about one round.
// How blocks reach CUs (measured through the strided data gradient, tools/dev/dgrad_cgroup2.sh): the dispatcher deals the
// blocks of an XCD round-robin over its 32 CUs, so with tiles_n N-tiles per M-tile the k-th M-tile of an XCD's run lands on
// CU slot k mod R, R = 32 / tiles_n, and a CU's time is the SUM of what its slot gets.  On a 4x4 map (3x3 pad 1: tiles of
// 9 / 6 / 4 taps) heaviest-first within an image block gave the slots 24 vs 16 tap-units where the mean is 18.75 -- the
// whole gap between these tiles' issue rate (0.68) and the image-major tiles' (0.855).  Here: every XCD gets a contiguous
// run of image blocks (operand locality in its L2) cut into balanced pieces, and inside the run the tiles are dealt to
// the R slots longest-processing-time-first with equal counts, then emitted round by round.
bool pixel_order_full(const int* taps, int npix, int nib, int tiles_n, unsigned char* out, int nruns) {
  // nruns: XCD runs the table's tiles are spread over (8: the whole launch; a strided layer's class-major launch gives
  // every parity class 2 of the 8 runs and uses one table for all four classes)
  static const bool on = []() { const char* e = contrad_dev_env("CONTRAD_PIXORDER_FULL"); return !(e && e[0] == '0'); }();
  const int nt = nib * npix;
  if (!on || nt > 256 || nt < nruns || nruns < 1 || nruns > 8 || tiles_n < 1 || tiles_n > 8 || (32 % tiles_n)) return false;
  // base order: image-block major; inside a block the pixels sorted by taps and dealt round-robin into `pieces` groups, so
  // that an XCD boundary inside a block leaves both sides the same work
  int sorted[256];
  for (int i = 0; i < npix; ++i) sorted[i] = i;
  for (int i = 1; i < npix; ++i) {
    const int v = sorted[i];
    int j = i - 1;
    while (j >= 0 && taps[sorted[j]] < taps[v]) { sorted[j + 1] = sorted[j]; --j; }
    sorted[j + 1] = v;
  }
  const int nb = nt * tiles_n, q = nb / nruns, r = nb % nruns, R = 32 / tiles_n;
  int tstart[9];
  for (int x = 0; x <= nruns; ++x) {
    const int b0 = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;      // first block of run x (xcd_remap)
    tstart[x] = x == nruns ? nt : cdiv(b0, tiles_n);                       // first M-tile whose blocks start in its run
  }
  int pieces = 1;                       // an image block is cut by XCD boundaries into `pieces` equal parts (or not at all)
  for (; pieces < npix; pieces *= 2) {
    if (npix % pieces) { pieces = 1; break; }
    bool ok = true;
    for (int x = 1; x < nruns; ++x) ok = ok && (tstart[x] % (npix / pieces) == 0);
    if (ok) break;
  }
  if (pieces >= npix || npix % pieces) pieces = 1;
  int base[256];
  for (int ib = 0; ib < nib; ++ib) {
    int k = 0;
    for (int g = 0; g < pieces; ++g)
      for (int i = g; i < npix; i += pieces) base[ib * npix + k++] = ib * npix + sorted[i];
  }
  // per XCD (xcd_remap: contiguous runs of blocks): deal its tiles to the slots
  int pos = 0;
  for (int x = 0; x < nruns; ++x) {
    const int t0 = tstart[x], t1 = tstart[x + 1];
    const int n = t1 - t0;
    if (n <= 0) continue;
    if (cdiv(n, R) > 16) return false;
    int idx[256], load[32] = {0}, cnt[32] = {0}, bin[32][16];
    for (int i = 0; i < n; ++i) idx[i] = base[t0 + i];
    for (int i = 1; i < n; ++i) {        // heaviest first (stable)
      const int v = idx[i];
      int j = i - 1;
      while (j >= 0 && taps[idx[j] % npix] < taps[v % npix]) { idx[j + 1] = idx[j]; --j; }
      idx[j + 1] = v;
    }
    const int cap = cdiv(n, R);
    for (int i = 0; i < n; ++i) {
      int best = -1;
      for (int j = 0; j < R; ++j)
        if (cnt[j] < cap && (best < 0 || load[j] < load[best])) best = j;
      bin[best][cnt[best]++] = idx[i];
      load[best] += taps[idx[i] % npix];
    }
    for (int round = 0; round < cap; ++round)
      for (int j = 0; j < R; ++j)
        if (round < cnt[j]) out[pos++] = (unsigned char)bin[j][round];
  }
  return pos == nt;
}

// ---- border classes -------------------------------------------------------------------------------------------------
// Along each axis the pixels of the map fall into runs with the same set of non-padding filter rows / columns (3x3 pad 1:
// first row, interior, last row); the products of an h-run and a w-run are rectangles whose pixels all read padding at
// the SAME taps.  An M-tile that enumerates (image, pixel) inside one rectangle can skip those taps (as a pixel-major
// tile does) and still shares the input neighbourhoods of adjacent pixels between its rows (which a pixel-major tile
// cannot).  mode 0: output pixels of the forward conv; mode 1: dx pixels of the stride-1 data gradient.
struct BorderClasses {
  int n;
  int h0[16], w0[16], hc[16], wc[16], taps[16];
};

int axis_runs(const contrad_conv_desc* d, int mode, bool along_h, int* start, int* len, int* ntaps) {
  const int extent = mode == MODE_FWD ? (along_h ? d->Ho : d->Wo) : (along_h ? d->H : d->W);
  const int in_ext = mode == MODE_FWD ? (along_h ? d->H : d->W) : (along_h ? d->Ho : d->Wo);
  const int k = along_h ? d->KH : d->KW;
  int runs = 0;
  unsigned prev = 0;
  for (int o = 0; o < extent; ++o) {
    unsigned m = 0;
    for (int t = 0; t < k; ++t) {
      const int i = mode == MODE_FWD ? o * d->stride - d->pad + t : o + d->pad - t;   // (DGRAD: stride 1)
      if ((unsigned)i < (unsigned)in_ext) m |= 1u << t;
    }
    if (o == 0 || m != prev) {
      if (runs == 16) return -1;
      start[runs] = o; len[runs] = 0; ntaps[runs] = __builtin_popcount(m);
      ++runs;
      prev = m;
    }
    ++len[runs - 1];
  }
  return runs;
}

bool border_classes(const contrad_conv_desc* d, int mode, BorderClasses* bc) {
  int hs[16], hl[16], ht[16], ws[16], wl[16], wt[16];
  const int nh = axis_runs(d, mode, true, hs, hl, ht), nw = axis_runs(d, mode, false, ws, wl, wt);
  if (nh <= 0 || nw <= 0 || nh * nw > 16 || nh * nw < 2) return false;
  bc->n = 0;
  for (int a = 0; a < nh; ++a)
    for (int b = 0; b < nw; ++b) {
      const int i = bc->n++;
      bc->h0[i] = hs[a]; bc->hc[i] = hl[a]; bc->w0[i] = ws[b]; bc->wc[i] = wl[b]; bc->taps[i] = ht[a] * wt[b];
    }
  for (int i = 1; i < bc->n; ++i)      // heaviest class first (stable insertion sort)
    for (int j = i; j > 0 && bc->taps[j - 1] < bc->taps[j]; --j) {
      std::swap(bc->h0[j], bc->h0[j - 1]); std::swap(bc->hc[j], bc->hc[j - 1]); std::swap(bc->w0[j], bc->w0[j - 1]);
      std::swap(bc->wc[j], bc->wc[j - 1]); std::swap(bc->taps[j], bc->taps[j - 1]);
    }
  return true;
}

// tile mode of a lean FWD / DGRAD launch: 0 image-major, 1 pixel-major, 2 border classes
int tile_mode_override() {
  static const int m = []() { const char* e = contrad_dev_env("CONTRAD_TILEMODE"); return e ? atoi(e) : -1; }();   // dev
  return m;
}

bool border_classes_ok(const contrad_conv_desc* d, int mode, int bm) {
  if (!pixmajor_enabled() || (mode == MODE_DGRAD && d->stride != 1)) return false;
  BorderClasses bc;
  if (!border_classes(d, mode, &bc)) return false;
  int minpix = 1 << 30;
  for (int i = 0; i < bc.n; ++i) minpix = std::min(minpix, bc.hc[i] * bc.wc[i]);
  const long long imgs = bm / minpix + 2;        // images one M-tile of the smallest class can touch
  if (imgs * d->H * d->W * d->ldx * 4 >= (1ll << 30) || imgs * d->Ho * d->Wo * d->ldy * 4 >= (1ll << 30)) return false;
  // every XCD takes an eighth of every class: the smallest class must have a tile for each of them.  (192 images on an
  // 8x8 map: 2 half-empty corner tiles, 9 edge tiles per side -- StyleGAN2-32 went 16.2 -> 16.7 ms with classes that small.)
  if (tile_mode_override() != 2 && (long long)d->N * minpix < 8ll * bm) return false;
  if ((long long)d->N * minpix < bm) return false;                     // not even one full tile in the smallest class
  return true;
}

void fill_border_classes(const contrad_conv_desc* d, int mode, int bm, IgemmArgs* a) {
  BorderClasses bc;
  border_classes(d, mode, &bc);
  a->nwin = bc.n;
  int t = 0;
  for (int i = 0; i < bc.n; ++i) {
    a->win_h0[i] = (short)bc.h0[i]; a->win_w0[i] = (short)bc.w0[i];
    a->win_hc[i] = (short)bc.hc[i]; a->win_wc[i] = (short)bc.wc[i];
    a->win_tile0[i] = t;
    t += (int)cdivll((long long)d->N * bc.hc[i] * bc.wc[i], bm);
  }
  a->win_tile0[bc.n] = t;
  // grid: every XCD (block b -> XCD b % 8) walks its eighth of every class; padded to the largest per-XCD share
  int most = 0;
  for (int x = 0; x < 8; ++x) {
    int sum = 0;
    for (int i = 0; i < bc.n; ++i) {
      const int n_c = a->win_tile0[i + 1] - a->win_tile0[i];
      sum += (((x + 1) * n_c) >> 3) - ((x * n_c) >> 3);
    }
    most = std::max(most, sum);
  }
  a->tiles_m = 8 * most;
}

int border_class_tiles(const contrad_conv_desc* d, int mode, int bm) {
  IgemmArgs a{};
  fill_border_classes(d, mode, bm, &a);
  return a.tiles_m;
}

FwdPlan fwd_plan(const contrad_conv_desc* d) {
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const bool vec = vec_ok(d, MODE_FWD);
  const bool lean = vec && lean_ok(d, MODE_FWD, 0);
  FwdPlan p{64, 64, 1, 0, 0, 0};
  pick_tile(M, d->K, vec, lean, 4, &p.bm, &p.bn);
  const int t_total = d->KH * d->KW * d->C / BK;
  p.tps = t_total;
  if (lean && splitk_enabled() && p.bn != 32 && t_total >= 32)
    split_plan(M, d->K, t_total, 2.0 * (double)M * d->K * d->C * d->KH * d->KW, &p);
  if (lean && p.splits <= 1) {
    const int force = tile_mode_override();
    const double valid = fwd_valid_tap_fraction(d);
    if (force == 2 ? border_classes_ok(d, MODE_FWD, p.bm) : (force < 0 && fwd_border_ok(d, p.bm, valid))) p.pixmajor = 2;
    else if (force == 1 || force < 0) p.pixmajor = fwd_pixmajor_ok(d, p.bm) ? 1 : 0;
  }
  return p;
}

long long dgrad_balance(const contrad_conv_desc* d, int tiles_m, IgemmArgs* a);

// Class-group size of the plain strided order: the M-tiles x N-tiles of one class inside a group are consecutive block
// ids, and 32 of them = one block for every CU of an XCD (round 2 had a fixed 8, which is this for the 4 N-tiles of the
// 512-channel layers only; with 1 / 2 / 8 N-tiles 32 / 16 / 4 measured 20 - 35 % faster on single-round launches).
int dgrad_cgroup(int tiles_m, int tiles_n) {
  static const int forced = []() { const char* e = contrad_dev_env("CONTRAD_DGRAD_CGROUP"); return e ? atoi(e) : 0; }();
  int g = forced > 0 ? forced : std::max(1, 32 / std::max(tiles_n, 1));
  return g < tiles_m ? g : tiles_m;
}

// DGRAD plan.  Stride 1 on the lean kernel may split K like FWD (small-M, deep-K layers: the 4x4 / 8x8 levels at small
// per-rank batches ran 64x64 tiles at 2/3 of the 128x128 tile's matrix-pipe utilisation just to have enough blocks);
// strided DGRAD keeps grid.y for its parity classes.  Without a workspace the plan never splits.
FwdPlan dgrad_plan(const contrad_conv_desc* d, bool may_split) {
  const int s = d->stride;
  const long long Mc = (long long)d->N * cdiv(d->H, s) * cdiv(d->W, s);   // largest class
  const bool vec = vec_ok(d, MODE_DGRAD);
  const bool lean = vec && lean_ok(d, MODE_DGRAD, 0);
  FwdPlan p{64, 64, 1, 0, 0, 0};
  pick_tile(Mc, d->C, vec, lean, 4 * s * s, &p.bm, &p.bn);
  if (lean && s == 2) {
    // Unequal parity classes (3x3 stride 2: 4 / 2 / 2 / 1 taps).  Measured on the StyleGAN2 shapes at 16 ... 192 images
    // (tools/dev/dgrad_cgroup2.sh, dgrad_balance.sh): a launch that fits the chip in about one round
    // (<= 1024 blocks = 4 per CU) is fastest in the plain class-group order with 32-block class chunks -- the dispatcher
    // deals an XCD's blocks round-robin over its 32 CUs, so every CU then holds one block of EVERY class; a longer launch
    // is fastest with equal work per block (dgrad_balance: blocks of the light classes walk 2 / 4 M-tiles).
    IgemmArgs probe{};
    static const bool force = []() { const char* e = contrad_dev_env("CONTRAD_DGRAD_BALANCE"); return e && e[0] == '2'; }();   // dev
    if (dgrad_balance(d, 8, &probe) > 0 && (force || cdivll(Mc, p.bm) * cdiv(d->C, p.bn) * s * s > 1024)) {
      p.balanced = 1;
      pick_tile(Mc, d->C, vec, lean, 4 * probe.cb_start[4] / probe.cbal, &p.bm, &p.bn);   // 9 / 4 blocks per tile index
    }
  }
  const int t_total = d->KH * d->KW * d->K / BK;
  p.tps = t_total;
  if (may_split && s == 1 && lean && splitk_enabled() && p.bn != 32 && t_total >= 32 && !(d->C & 3) && !(d->ldx & 3))
    split_plan(Mc, d->C, t_total, 2.0 * (double)Mc * d->K * d->C * d->KH * d->KW, &p);
  if (lean && p.splits <= 1) {
    const int force = tile_mode_override();
    const double valid = dgrad_valid_tap_fraction(d);
    if (force == 2 ? border_classes_ok(d, MODE_DGRAD, p.bm) : (force < 0 && dgrad_border_ok(d, p.bm, valid))) p.pixmajor = 2;
    else if (force == 1 || force < 0) p.pixmajor = dgrad_pixmajor_ok(d, p.bm) ? 1 : 0;
  }
  return p;
}

// Balanced block order of a strided lean DGRAD (igemm_lean.h, p.cbal): possible when the parity classes carry different tap
// counts and every count divides the largest (3x3 stride 2: 4, 2, 2, 1 -> runs of 1, 2, 2, 4 M-tiles per block; the 4x4
// stride-2 layers of SNDCGAN have four equal classes and keep the plain order).  Fills a->cbal / cb_reps / cb_start and
// returns the number of blocks per N-tile (M-direction), or 0 when the plain order stays.
long long dgrad_balance(const contrad_conv_desc* d, int tiles_m, IgemmArgs* a) {
  static const bool on = []() { const char* e = contrad_dev_env("CONTRAD_DGRAD_BALANCE"); return !(e && e[0] == '0'); }();
  const int s = d->stride;
  if (!on || s != 2) return 0;
  int taps[4], most = 0, least = 1 << 30;
  for (int c = 0; c < 4; ++c) {
    const int ph = c / s, pw = c % s;
    const int kh0 = (ph + d->pad) % s, kw0 = (pw + d->pad) % s;
    const int nth = kh0 < d->KH ? (d->KH - kh0 + s - 1) / s : 0, ntw = kw0 < d->KW ? (d->KW - kw0 + s - 1) / s : 0;
    taps[c] = nth * ntw;
    most = std::max(most, taps[c]); least = std::min(least, taps[c]);
  }
  if (least <= 0 || most == least) return 0;           // an empty class (1x1 stride 2) or nothing to balance
  int start = 0;
  const int R = 8;                                     // M-tile indices of every class per group (the plain order's cgroup)
  for (int c = 0; c < 4; ++c) {
    if (most % taps[c] || R % (most / taps[c])) return 0;
    a->cb_reps[c] = most / taps[c];
    a->cb_start[c] = start;
    start += R / a->cb_reps[c];
  }
  a->cb_start[4] = start;
  a->cbal = R;
  return (long long)cdiv(tiles_m, R) * start;
}

// dx[m][c] = gain * act'(act_ref[m][c]) * sum_s ws[s][m][c]   (slabs in dx's own layout, fixed summation order)
__global__ void dgrad_reduce_kernel(const float* __restrict__ ws, int splits, long long slab, long long rows, int C,
                                    int ldx, const float* __restrict__ act_ref, float slope, float gain,
                                    float* __restrict__ dx) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  const float g1 = gain, g0 = gain * slope;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long m = e / c4n;
    const long long off = m * ldx + (e - m * c4n) * 4;
    float4 v = ld4(ws + off);
    for (int k = 1; k < splits; ++k) {
      const float4 u = ld4(ws + (long long)k * slab + off);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (act_ref) {
      const float4 a = ld4(act_ref + off);
      v.x *= (a.x > 0.f) ? g1 : g0; v.y *= (a.y > 0.f) ? g1 : g0;
      v.z *= (a.z > 0.f) ? g1 : g0; v.w *= (a.w > 0.f) ? g1 : g0;
    }                                   // (no act_ref: raw sums, like the kernels' own epilogue)
    *reinterpret_cast<float4*>(dx + off) = v;
  }
}

int launch_wino22_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp, float* dbias,
                        float* workspace, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino22::wino22_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino22::W_LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  wino22::WArgs a = wino22_wgrad_args(d);
  const int splits = wino22_wgrad_splits(a);
  const long long total = 16ll * d->C * d->K;
  a.x = x; a.gy = gy; a.ws = workspace;
  a.bias_ws = dbias ? workspace + (size_t)splits * total : nullptr;
  hipLaunchKernelGGL(wino22::wino22_wgrad_kernel, dim3(a.RBN * a.KB * splits), dim3(512), wino22::W_LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  int R = 1;
  while (R < 64 && R * 2 <= splits && (total / 4) * R < 65536) R <<= 1;
  long long rb = ((total / 4) * R + 255) / 256;
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)rb), dim3(256), 0, stream, workspace, dwp, 16 * d->C, d->K, d->ldw, splits,
                     a.bias_ws, dbias, R);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

int launch_wino_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp, float* dbias,
                      float* workspace, hipStream_t stream) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)wino::wino_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     wino::W_LDS_DWORDS * 4);
  if (attr != hipSuccess) return (int)attr;
  wino::WArgs a = wino_wgrad_args(d);
  const int splits = wino_wgrad_splits(a);
  const long long total = 9ll * d->C * d->K;
  a.x = x; a.gy = gy; a.ws = workspace;
  a.bias_ws = dbias ? workspace + (size_t)splits * total : nullptr;
  hipLaunchKernelGGL(wino::wino_wgrad_kernel, dim3(a.CB * a.KB * splits), dim3(512), wino::W_LDS_DWORDS * 4, stream, a);
  CONTRAD_CHECK_LAUNCH();
  int R = 1;
  while (R < 64 && R * 2 <= splits && (total / 4) * R < 65536) R <<= 1;
  long long rb = ((total / 4) * R + 255) / 256;
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)rb), dim3(256), 0, stream, workspace, dwp, 9 * d->C, d->K, d->ldw, splits,
                     a.bias_ws, dbias, R);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int contrad_abi_version(void) { return 2; }

extern "C" long long contrad_conv2d_fwd_workspace_bytes(const contrad_conv_desc* d) {
  if (check_desc(d)) return -22;
  if (wino44_planned(d, MODE_FWD)) return wino44_workspace_bytes(d);
  if (wino_planned(d, MODE_FWD)) return wino_workspace_bytes(d);
  if (wino22_planned(d, MODE_FWD)) return wino22_workspace_bytes(d);
  if (wino23_planned(d, MODE_FWD)) return wino23_workspace_bytes(d);
  const FwdPlan p = fwd_plan(d);
  if (p.splits <= 1) return 0;
  return (long long)p.splits * d->N * d->Ho * d->Wo * d->K * (long long)sizeof(float);
}

extern "C" int contrad_conv2d_fwd_add(const contrad_conv_desc* d, const float* x, const float* wp,
                                      const float* bias, const float* addend, float* y, float slope, float gain,
                                      float* workspace, long long workspace_bytes, contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(x && wp && y);
  if (vec_ok(d, MODE_FWD)) CONTRAD_ARG(aligned16(x, wp, y) && aligned16(addend, nullptr, nullptr));
  IgemmArgs a{};
  a.A = x; a.B = wp; a.C = y; a.bias = bias; a.addend = addend; a.d = *d; a.slope = slope; a.gain = gain;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  CONTRAD_ARG(M < (1ll << 31));
  a.M = (int)M; a.Ncol = d->K; a.Kg = d->KH * d->KW * d->C;
  a.st_nt = st_nt_for(M, d->ldy);
  if (wino44_planned(d, MODE_FWD) && workspace && workspace_bytes >= wino44_workspace_bytes(d)) {   // Winograd F(4x4, 3x3), wino44.h
    CONTRAD_ARG(aligned16(x, wp, workspace));
    return launch_wino44<MODE_FWD>(d, x, wp, bias, addend, y, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino_planned(d, MODE_FWD) && workspace && workspace_bytes >= wino_workspace_bytes(d)) {   // Winograd F(2x2, 3x3), wino.h
    CONTRAD_ARG(aligned16(x, wp, workspace));
    return launch_wino<MODE_FWD>(d, x, wp, bias, addend, y, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino22_planned(d, MODE_FWD) && workspace && workspace_bytes >= wino22_workspace_bytes(d)) {   // F(2x2, 2x2) on the phases, wino22.h
    CONTRAD_ARG(aligned16(x, wp, workspace));
    return launch_wino22<MODE_FWD>(d, x, wp, bias, addend, y, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino23_planned(d, MODE_FWD) && workspace && workspace_bytes >= wino23_workspace_bytes(d)) {   // 3x3 stride 2: F(2x2, 2x2) on the phases, wino23.h
    CONTRAD_ARG(aligned16(x, wp, workspace));
    return launch_wino23(d, x, wp, bias, addend, y, slope, gain, workspace, (hipStream_t)stream);
  }
  if (conv_c32_ok(d))   // weight-stationary kernel (conv_c32.h); (alignment is an argument error above, so the dispatch is
                        // exactly what contrad_conv2d_path / _grid_blocks report)
    return launch_conv_c32<MODE_FWD>(d, x, wp, y, bias, addend, nullptr, slope, gain, (hipStream_t)stream);
  if (fwd_k1_ok(d)) {
    hipLaunchKernelGGL(fwd_k1_kernel, dim3((unsigned)cdivll(M, 4)), dim3(256), 0, (hipStream_t)stream, x, wp, M, d->C,
                       d->ldx, d->ldw, bias, slope, gain, y, d->ldy, addend);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  const bool vec = vec_ok(d, MODE_FWD);
  const FwdPlan p = fwd_plan(d);
  a.tiles_m = p.pixmajor == 1 ? cdiv(d->N, p.bm) * d->Ho * d->Wo : cdiv(a.M, p.bm);
  a.tiles_n = cdiv(a.Ncol, p.bn);
  a.pixmajor = p.pixmajor == 1;
  if (p.pixmajor == 2) fill_border_classes(d, MODE_FWD, p.bm, &a);      // (sets tiles_m)
  if (p.pixmajor == 1) {
    int taps[256];
    for (int ho = 0; ho < d->Ho; ++ho)
      for (int wo = 0; wo < d->Wo; ++wo) {
        int vh = 0, vw = 0;
        for (int kh = 0; kh < d->KH; ++kh) vh += (unsigned)(ho * d->stride - d->pad + kh) < (unsigned)d->H;
        for (int kw = 0; kw < d->KW; ++kw) vw += (unsigned)(wo * d->stride - d->pad + kw) < (unsigned)d->W;
        taps[ho * d->Wo + wo] = vh * vw;
      }
    pixel_order(taps, d->Ho * d->Wo, a.px_order);
    unsigned char full[256];
    if (pixel_order_full(taps, d->Ho * d->Wo, cdiv(d->N, p.bm), a.tiles_n, full, 8)) {
      memcpy(a.px_order, full, sizeof(full));
      a.px_full = 1;
    }
  }
  a.ptiles_per_split = p.tps;
  if (p.splits > 1) {
    CONTRAD_ARG(workspace && workspace_bytes >= contrad_conv2d_fwd_workspace_bytes(d));
    a.C = workspace;
  }
  rc = dispatch<MODE_FWD>(a, p.bm, p.bn, vec, dim3(a.tiles_m * a.tiles_n, p.splits), (hipStream_t)stream);
  if (rc || p.splits <= 1) return rc;
  const long long total = M * d->K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fwd_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, p.splits, M,
                     d->K, bias, slope, gain, y, d->ldy, addend);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_conv2d_fwd(const contrad_conv_desc* d, const float* x, const float* wp,
                                  const float* bias, float* y, float slope, float gain, float* workspace,
                                  long long workspace_bytes, contrad_stream_t stream) {
  return contrad_conv2d_fwd_add(d, x, wp, bias, nullptr, y, slope, gain, workspace, workspace_bytes, stream);
}

extern "C" long long contrad_conv2d_dgrad_workspace_bytes(const contrad_conv_desc* d) {
  if (check_desc(d)) return -22;
  if (wino44_planned(d, MODE_DGRAD)) return wino44_workspace_bytes(d);
  if (wino_planned(d, MODE_DGRAD)) return wino_workspace_bytes(d);
  if (wino22_planned(d, MODE_DGRAD)) return wino22_workspace_bytes(d);
  const FwdPlan p = dgrad_plan(d, true);
  if (p.splits <= 1) return 0;
  return (long long)p.splits * d->N * d->H * d->W * d->ldx * (long long)sizeof(float);
}

extern "C" int contrad_conv2d_dgrad_ws(const contrad_conv_desc* d, const float* gy, const float* wp,
                                       float* dx, const float* act_ref, float slope, float gain,
                                       float* workspace, long long workspace_bytes, contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(gy && wp && dx);
  if (vec_ok(d, MODE_DGRAD)) CONTRAD_ARG(aligned16(gy, wp, dx) && aligned16(act_ref, nullptr, nullptr));
  if (wino44_planned(d, MODE_DGRAD) && workspace && workspace_bytes >= wino44_workspace_bytes(d)) {   // wino44.h: mirrored filter
    CONTRAD_ARG(aligned16(gy, wp, workspace));
    return launch_wino44<MODE_DGRAD>(d, gy, wp, nullptr, act_ref, dx, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino_planned(d, MODE_DGRAD) && workspace && workspace_bytes >= wino_workspace_bytes(d)) {   // wino.h: mirrored filter
    CONTRAD_ARG(aligned16(gy, wp, workspace));
    return launch_wino<MODE_DGRAD>(d, gy, wp, nullptr, act_ref, dx, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino22_planned(d, MODE_DGRAD) && workspace && workspace_bytes >= wino22_workspace_bytes(d)) {   // wino22.h: one item per dx phase
    CONTRAD_ARG(aligned16(gy, wp, workspace));
    return launch_wino22<MODE_DGRAD>(d, gy, wp, nullptr, act_ref, dx, slope, gain, workspace, (hipStream_t)stream);
  }
  if (conv_c32_ok(d))   // stride-1 pad-1 3x3: the same weight-stationary kernel with the filter mirrored (conv_c32.h)
    return launch_conv_c32<MODE_DGRAD>(d, gy, wp, dx, nullptr, nullptr, act_ref, slope, gain, (hipStream_t)stream);
  // every input pixel must be covered by at least one tap of its parity class, otherwise the class
  // (whose gradient is exactly zero) still writes zeros: handled by Kg == 0 -> T == 0 -> acc = 0.
  IgemmArgs a{};
  a.A = gy; a.B = wp; a.C = dx; a.act_ref = act_ref; a.d = *d; a.slope = slope; a.gain = gain;
  a.st_nt = st_nt_for((long long)d->N * d->H * d->W, d->ldx);
  const int s = d->stride;
  const long long Mc = (long long)d->N * cdiv(d->H, s) * cdiv(d->W, s);  // largest class
  CONTRAD_ARG(Mc < (1ll << 31));
  const bool vec = vec_ok(d, MODE_DGRAD);
  const FwdPlan pl = dgrad_plan(d, workspace != nullptr);
  const int bm = pl.bm, bn = pl.bn;
  a.pixmajor = pl.pixmajor == 1;
  a.px_pixels = cdiv(d->H, s) * cdiv(d->W, s);
  a.tiles_m = pl.pixmajor == 1 ? cdiv(d->N, bm) * a.px_pixels : cdiv((int)Mc, bm);
  if (pl.pixmajor == 2) fill_border_classes(d, MODE_DGRAD, bm, &a);     // (sets tiles_m)
  if (pl.pixmajor == 1) {
    int taps[256];
    for (int i = 0; i < a.px_pixels; ++i) taps[i] = 1;
    if (s == 1)     // (strided: per-class pixel sets, left in row-major order)
      for (int h = 0; h < d->H; ++h)
        for (int w = 0; w < d->W; ++w) {
          int vh = 0, vw = 0;
          for (int kh = 0; kh < d->KH; ++kh) vh += (unsigned)(h + d->pad - kh) < (unsigned)d->Ho;
          for (int kw = 0; kw < d->KW; ++kw) vw += (unsigned)(w + d->pad - kw) < (unsigned)d->Wo;
          taps[h * d->W + w] = vh * vw;
        }
    pixel_order(taps, a.px_pixels, a.px_order);
    unsigned char full[256];
    if (s == 1 && pixel_order_full(taps, a.px_pixels, cdiv(d->N, bm), cdiv(d->C, bn), full, 8)) {
      memcpy(a.px_order, full, sizeof(full));
      a.px_full = 1;
    } else if (s == 2 && dgrad_strided_full_ok(d, bm, bn)) {
      int t0[256];
      const int npix = dgrad_strided_class0_taps(d, t0);
      pixel_order_full(t0, npix, cdiv(d->N, bm), cdiv(d->C, bn), full, 2);
      memcpy(a.px_order, full, sizeof(full));
      a.px_full = 2;
    }
  }
  a.tiles_n = cdiv(d->C, bn);
  if (pl.splits > 1) {
    // stride-1 split-K: every split writes raw partial sums into its own slab (dx's layout), dgrad_reduce_kernel sums
    // them in a fixed order and applies the fused act' epilogue
    CONTRAD_ARG(workspace_bytes >= contrad_conv2d_dgrad_workspace_bytes(d));
    a.dsplits = pl.splits;
    a.ptiles_per_split = pl.tps;
    a.slab_elems = (long long)d->N * d->H * d->W * d->ldx;
    CONTRAD_ARG(a.slab_elems * 4 < (1ll << 40));
    a.C = workspace; a.act_ref = nullptr;
    rc = dispatch<MODE_DGRAD>(a, bm, bn, vec, dim3(a.tiles_m * a.tiles_n, pl.splits), (hipStream_t)stream);
    if (rc) return rc;
    const long long rows = (long long)d->N * d->H * d->W;
    long long blocks = (rows * (d->C / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, workspace,
                       pl.splits, a.slab_elems, rows, d->C, d->ldx, act_ref, slope, gain, dx);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  // Block order of the strided lean DGRAD: the parity classes of a 3x3 stride-2 conv contract over 4, 2, 2 and 1 taps.
  // With the classes of one M-tile on consecutive block ids (tile_n, class, tile_m) the kernel took exactly as long as
  // its heaviest class alone (measured by launching single classes: 1.03 ms for class (0,0) vs 1.10 ms for all four;
  // k3 / k4 / k5 kernels: 72 / 128 / 96 TF/s = 9/16, 16/16, 25/36 of the balanced rate) -- blocks that become resident
  // together should carry equal work.  Groups of 8 M-tiles, class-major inside a group: neighbours are equal, and the
  // gy rows a group's classes share are still cache-resident when the next class reads them.  3x3 s2 at batch 32:
  // 72 -> 99, 84 -> 102, 78 -> 117, 82 -> 99 TF/s; groups of 32 / 64 lose again on layers with < ~100 M-tiles (few
  // groups -> a tail of light classes).  tools/bench_conv.py; CONTRAD_DGRAD_CGROUP=0 restores the old order.
  a.cgroup = (s > 1) ? dgrad_cgroup(a.tiles_m, a.tiles_n) : 0;
  if (a.px_full == 2) a.cgroup = a.tiles_m;           // class-major over the whole launch: class c = XCD runs 2c, 2c + 1
  // ... and equal work per block where the classes are unequal (3x3 stride 2: 4 / 2 / 2 / 1 taps): a block of a light
  // class walks 2 / 4 consecutive M-tiles (dgrad_balance(); igemm_lean.h).  Neighbours are then equal AND every block of
  // the launch carries the same number of K-tiles, so the tail of the launch is not a few 4-tap blocks running alone.
  if (pl.balanced && vec && lean_ok(d, MODE_DGRAD, 0) && !a.pixmajor && a.nwin == 0) {
    const long long blocks = dgrad_balance(d, a.tiles_m, &a);
    if (blocks > 0) {
      a.cgroup = 0;
      return dispatch<MODE_DGRAD>(a, bm, bn, vec, dim3((unsigned)(blocks * a.tiles_n), 1), (hipStream_t)stream);
    }
  }
  const int tm_pad = a.cgroup > 0 ? cdiv(a.tiles_m, a.cgroup) * a.cgroup : a.tiles_m;
  return dispatch<MODE_DGRAD>(a, bm, bn, vec, dim3(tm_pad * a.tiles_n, s * s), (hipStream_t)stream);
}

extern "C" int contrad_conv2d_dgrad(const contrad_conv_desc* d, const float* gy, const float* wp,
                                    float* dx, const float* act_ref, float slope, float gain,
                                    contrad_stream_t stream) {
  return contrad_conv2d_dgrad_ws(d, gy, wp, dx, act_ref, slope, gain, nullptr, 0, stream);   // never splits K
}

extern "C" int contrad_conv2d_wino_ok(const contrad_conv_desc* d, int mode) {
  if (check_desc(d)) return -22;
  if (mode == MODE_WGRAD) return (wino_wgrad_ok(d) || wino22_wgrad_ok(d)) ? 1 : 0;
  return (wino_ok(d, mode) || wino22_ok(d, mode) || wino23_ok(d, mode)) ? 1 : 0;
}

extern "C" long long contrad_conv2d_wino_workspace_bytes(const contrad_conv_desc* d, int mode) {
  if (check_desc(d) || mode < 0 || mode > 2) return -22;
  if (mode == MODE_WGRAD) return wino_wgrad_ok(d) ? wino_wgrad_workspace_bytes(d) : wino22_wgrad_ok(d) ? wino22_wgrad_workspace_bytes(d) : -22;
  return d->KH == 4 ? wino22_workspace_bytes(d) : d->stride == 2 ? wino23_workspace_bytes(d) : wino_workspace_bytes(d);
}

extern "C" int contrad_conv2d_wino_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp,
                                         float* dbias, float* workspace, long long workspace_bytes,
                                         contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(x && gy && dwp && workspace && aligned16(x, gy, workspace));
  if (wino22_wgrad_ok(d)) {
    CONTRAD_ARG(workspace_bytes >= wino22_wgrad_workspace_bytes(d));
    return launch_wino22_wgrad(d, x, gy, dwp, dbias, workspace, (hipStream_t)stream);
  }
  CONTRAD_ARG(wino_wgrad_ok(d) && workspace_bytes >= wino_wgrad_workspace_bytes(d));
  return launch_wino_wgrad(d, x, gy, dwp, dbias, workspace, (hipStream_t)stream);
}

extern "C" int contrad_conv2d_wino(const contrad_conv_desc* d, int mode, const float* in, const float* wp,
                                   const float* bias, const float* ref, float* out, float slope, float gain,
                                   float* workspace, long long workspace_bytes, contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(in && wp && out && workspace && (mode == MODE_FWD || mode == MODE_DGRAD));
  CONTRAD_ARG(aligned16(in, wp, workspace));
  if (wino22_ok(d, mode)) {        // 4x4 stride 2: F(2x2, 2x2) on the phases (wino22.h)
    CONTRAD_ARG(workspace_bytes >= wino22_workspace_bytes(d) && (mode == MODE_FWD || bias == nullptr));
    if (mode == MODE_FWD) return launch_wino22<MODE_FWD>(d, in, wp, bias, ref, out, slope, gain, workspace, (hipStream_t)stream);
    return launch_wino22<MODE_DGRAD>(d, in, wp, nullptr, ref, out, slope, gain, workspace, (hipStream_t)stream);
  }
  if (wino23_ok(d, mode)) {        // 3x3 stride 2 pad 0 (forward only): F(2x2, 2x2) on the phases, zero planes skipped (wino23.h)
    CONTRAD_ARG(workspace_bytes >= wino23_workspace_bytes(d));
    return launch_wino23(d, in, wp, bias, ref, out, slope, gain, workspace, (hipStream_t)stream);
  }
  CONTRAD_ARG(wino_ok(d, mode) && workspace_bytes >= wino_workspace_bytes(d));
  if (mode == MODE_FWD) return launch_wino<MODE_FWD>(d, in, wp, bias, ref, out, slope, gain, workspace, (hipStream_t)stream);
  CONTRAD_ARG(bias == nullptr);
  return launch_wino<MODE_DGRAD>(d, in, wp, nullptr, ref, out, slope, gain, workspace, (hipStream_t)stream);
}

extern "C" int contrad_conv2d_wino44_ok(const contrad_conv_desc* d, int mode) {
  if (check_desc(d)) return -22;
  return wino44_ok(d, mode) ? 1 : 0;
}

extern "C" long long contrad_conv2d_wino44_workspace_bytes(const contrad_conv_desc* d) {
  if (check_desc(d)) return -22;
  return wino44_workspace_bytes(d);
}

extern "C" int contrad_conv2d_wino44(const contrad_conv_desc* d, int mode, const float* in, const float* wp,
                                     const float* bias, const float* ref, float* out, float slope, float gain,
                                     float* workspace, long long workspace_bytes, contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(in && wp && out && workspace && (mode == MODE_FWD || mode == MODE_DGRAD));
  CONTRAD_ARG(aligned16(in, wp, workspace) && aligned16(ref, nullptr, nullptr));
  CONTRAD_ARG(wino44_ok(d, mode) && workspace_bytes >= wino44_workspace_bytes(d));
  if (mode == MODE_FWD) return launch_wino44<MODE_FWD>(d, in, wp, bias, ref, out, slope, gain, workspace, (hipStream_t)stream);
  CONTRAD_ARG(bias == nullptr);
  return launch_wino44<MODE_DGRAD>(d, in, wp, nullptr, ref, out, slope, gain, workspace, (hipStream_t)stream);
}

extern "C" int contrad_conv2d_tile(const contrad_conv_desc* d, int mode, int* bm, int* bn) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(bm && bn && mode >= 0 && mode <= 2);
  if (mode == MODE_FWD) {
    const FwdPlan p = fwd_plan(d);
    *bm = p.bm; *bn = p.bn;
  } else if (mode == MODE_DGRAD) {
    const FwdPlan p = dgrad_plan(d, true);
    *bm = p.bm; *bn = p.bn;
  } else {
    int tm, tn, sp, pps;
    wgrad_plan(d, bm, bn, &tm, &tn, &sp, &pps);
  }
  return 0;
}

extern "C" int contrad_conv2d_path(const contrad_conv_desc* d, int mode) {
  if (check_desc(d) || mode < 0 || mode > 2) return -22;
  if (mode == MODE_FWD && fwd_k1_ok(d)) return 5;
  if (mode != MODE_WGRAD && wino44_planned(d, mode)) return wino44_n32(d, mode) ? 11 : 9;
  if (mode != MODE_WGRAD && wino_planned(d, mode)) return 7;
  if (mode != MODE_WGRAD && wino22_planned(d, mode)) return 8;
  if (wino23_planned(d, mode)) return 10;
  if (mode == MODE_WGRAD && wino_wgrad_planned(d)) return 7;
  if (mode == MODE_WGRAD && wino22_wgrad_planned(d)) return 8;
  if (mode != MODE_WGRAD && conv_c32_ok(d)) return 6;
  if (!vec_ok(d, mode)) return 0;
  if (mode == MODE_WGRAD && wgrad_c32_ok(d)) return 4;
  long long pps = 0;
  if (mode == MODE_WGRAD) {
    int bm, bn, tm, tn, sp, p;
    wgrad_plan(d, &bm, &bn, &tm, &tn, &sp, &p);
    pps = p;
  }
  if (!lean_ok(d, mode, pps)) return 1;
  if (mode == MODE_FWD && fwd_plan(d).pixmajor) return 3;
  if (mode == MODE_DGRAD && dgrad_plan(d, true).pixmajor) return 3;
  if (mode == MODE_WGRAD) {
    int bm, bn, tm, tn, sp, p;
    wgrad_plan(d, &bm, &bn, &tm, &tn, &sp, &p);
    if (wgrad_pixmajor_ok(d, bm, p)) return 3;
  }
  return 2;
}

extern "C" double contrad_conv2d_executed_fraction(const contrad_conv_desc* d, int mode) {
  if (check_desc(d) || mode < 0 || mode > 2) return -22.0;
  if (contrad_conv2d_path(d, mode) == 9 || contrad_conv2d_path(d, mode) == 11) return 0.25;        // 36 transform-domain multiply-adds per 4x4 tile instead of 144
  if (contrad_conv2d_path(d, mode) == 7) return 4.0 / 9.0;   // 16 transform-domain multiply-adds per 2x2 tile instead of 36
  if (contrad_conv2d_path(d, mode) == 8) return 9.0 / 16.0;  // four phases x 9 per 2x2 tile instead of 64
  if (contrad_conv2d_path(d, mode) == 10) return 25.0 / 36.0; // 9 + 6 + 6 + 4 planes of the four phases per 2x2 tile instead of 36
  if (contrad_conv2d_path(d, mode) != 3) return 1.0;
  return mode == MODE_DGRAD ? dgrad_valid_tap_fraction(d) : fwd_valid_tap_fraction(d);
}

extern "C" long long contrad_conv2d_grid_blocks(const contrad_conv_desc* d, int mode, int with_workspace) {
  if (check_desc(d) || mode < 0 || mode > 2) return -22;
  if (mode != MODE_WGRAD && with_workspace && wino44_planned(d, mode)) return wino44_grid(wino44_args(d, mode));   // (512 threads each)
  if (mode != MODE_WGRAD && with_workspace && wino_planned(d, mode)) return wino_grid(wino_args(d, mode));
  if (with_workspace && wino23_planned(d, mode)) return wino23_grid(wino23_args(d));
  if (mode != MODE_WGRAD && with_workspace && wino22_planned(d, mode)) {
    const wino22::Args a = wino22_args(d, mode);
    return 8 * std::min(WINO_CUS / 8, cdiv(a.NTB, 8) * a.NKB * (mode == MODE_DGRAD ? 4 : 1));
  }
  if (mode == MODE_FWD) {
    const FwdPlan p = fwd_plan(d);
    const long long M = (long long)d->N * d->Ho * d->Wo;
    if (fwd_k1_ok(d)) return cdivll(M, 4);
    if (conv_c32_ok(d)) return conv_c32_blocks(d);
    if (p.pixmajor == 1) return (long long)cdiv(d->N, p.bm) * d->Ho * d->Wo * cdiv(d->K, p.bn);
    if (p.pixmajor == 2) return (long long)border_class_tiles(d, MODE_FWD, p.bm) * cdiv(d->K, p.bn);
    return cdivll(M, p.bm) * cdiv(d->K, p.bn) * (with_workspace ? p.splits : 1);
  }
  if (mode == MODE_DGRAD) {
    if (conv_c32_ok(d)) return conv_c32_blocks(d);
    const int s = d->stride;
    const long long Mc = (long long)d->N * cdiv(d->H, s) * cdiv(d->W, s);
    const FwdPlan p = dgrad_plan(d, with_workspace != 0);
    const int tiles_m = p.pixmajor == 2 ? border_class_tiles(d, MODE_DGRAD, p.bm)
                        : p.pixmajor == 1 ? cdiv(d->N, p.bm) * cdiv(d->H, s) * cdiv(d->W, s) : cdiv((int)Mc, p.bm);
    const int tiles_n = cdiv(d->C, p.bn);
    if (p.splits > 1) return (long long)tiles_m * tiles_n * p.splits;
    if (p.balanced && vec_ok(d, MODE_DGRAD) && lean_ok(d, MODE_DGRAD, 0) && p.pixmajor == 0) {
      IgemmArgs a{};
      const long long blocks = dgrad_balance(d, tiles_m, &a);
      if (blocks > 0) return blocks * tiles_n;
    }
    if (p.pixmajor == 1 && s == 2) return (long long)tiles_m * tiles_n * s * s;     // class-major, one group (px_full == 2)
    const int cgroup = (s > 1) ? dgrad_cgroup(tiles_m, tiles_n) : 0;
    const int tm_pad = cgroup > 0 ? cdiv(tiles_m, cgroup) * cgroup : tiles_m;
    return (long long)tm_pad * tiles_n * s * s;
  }
  if (wino_wgrad_planned(d)) { const wino::WArgs a = wino_wgrad_args(d); return (long long)a.CB * a.KB * wino_wgrad_splits(a); }
  if (wino22_wgrad_planned(d)) { const wino22::WArgs a = wino22_wgrad_args(d); return (long long)a.RBN * a.KB * wino22_wgrad_splits(a); }
  if (wgrad_c32_ok(d)) return wgrad_c32_blocks(d);
  int bm, bn, tm, tn, splits, pps;
  wgrad_plan(d, &bm, &bn, &tm, &tn, &splits, &pps);
  return (long long)tm * tn * splits;
}

extern "C" int contrad_conv2d_tile_order(const contrad_conv_desc* d, int mode, unsigned char* out, int capacity) {
  if (check_desc(d) || (mode != MODE_FWD && mode != MODE_DGRAD) || !out || capacity <= 0) return -22;
  int taps[256];
  unsigned char full[256];
  int nt = 0;
  if (mode == MODE_FWD) {
    const FwdPlan p = fwd_plan(d);
    if (p.pixmajor != 1 || !vec_ok(d, MODE_FWD)) return 0;
    for (int ho = 0; ho < d->Ho; ++ho)
      for (int wo = 0; wo < d->Wo; ++wo) {
        int vh = 0, vw = 0;
        for (int kh = 0; kh < d->KH; ++kh) vh += (unsigned)(ho * d->stride - d->pad + kh) < (unsigned)d->H;
        for (int kw = 0; kw < d->KW; ++kw) vw += (unsigned)(wo * d->stride - d->pad + kw) < (unsigned)d->W;
        taps[ho * d->Wo + wo] = vh * vw;
      }
    if (!pixel_order_full(taps, d->Ho * d->Wo, cdiv(d->N, p.bm), cdiv(d->K, p.bn), full, 8)) return 0;
    nt = cdiv(d->N, p.bm) * d->Ho * d->Wo;
  } else {
    const FwdPlan p = dgrad_plan(d, true);
    if (p.pixmajor != 1 || p.splits > 1 || !vec_ok(d, MODE_DGRAD)) return 0;
    if (d->stride == 1) {
      for (int h = 0; h < d->H; ++h)
        for (int w = 0; w < d->W; ++w) {
          int vh = 0, vw = 0;
          for (int kh = 0; kh < d->KH; ++kh) vh += (unsigned)(h + d->pad - kh) < (unsigned)d->Ho;
          for (int kw = 0; kw < d->KW; ++kw) vw += (unsigned)(w + d->pad - kw) < (unsigned)d->Wo;
          taps[h * d->W + w] = vh * vw;
        }
      if (!pixel_order_full(taps, d->H * d->W, cdiv(d->N, p.bm), cdiv(d->C, p.bn), full, 8)) return 0;
      nt = cdiv(d->N, p.bm) * d->H * d->W;
    } else {
      const int npix = dgrad_strided_class0_taps(d, taps);
      if (!npix || !dgrad_strided_full_ok(d, p.bm, p.bn)) return 0;
      pixel_order_full(taps, npix, cdiv(d->N, p.bm), cdiv(d->C, p.bn), full, 2);
      nt = cdiv(d->N, p.bm) * npix;
    }
  }
  if (nt > capacity) return -22;
  memcpy(out, full, (size_t)nt);
  return nt;
}

extern "C" long long contrad_conv2d_wgrad_workspace_bytes(const contrad_conv_desc* d) {
  if (check_desc(d)) return -22;
  if (wino_wgrad_planned(d)) return wino_wgrad_workspace_bytes(d);
  if (wino22_wgrad_planned(d)) return wino22_wgrad_workspace_bytes(d);
  if (wgrad_c32_ok(d))
    return (long long)wgrad_c32_blocks(d) * ((long long)d->KH * d->KW * d->C + 1) * d->K * (long long)sizeof(float);
  int bm, bn, tm, tn, splits, pps;
  wgrad_plan(d, &bm, &bn, &tm, &tn, &splits, &pps);
  return (long long)splits * ((long long)d->KH * d->KW * d->C + 1) * d->K * (long long)sizeof(float);
}

extern "C" int contrad_conv2d_wgrad(const contrad_conv_desc* d, const float* x, const float* gy,
                                    float* dwp, float* dbias, float* workspace, long long workspace_bytes,
                                    contrad_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CONTRAD_ARG(x && gy && dwp && workspace);
  if (vec_ok(d, MODE_WGRAD)) CONTRAD_ARG(aligned16(x, gy, workspace));
  CONTRAD_ARG(workspace_bytes >= contrad_conv2d_wgrad_workspace_bytes(d));
  if (wino_wgrad_planned(d)) return launch_wino_wgrad(d, x, gy, dwp, dbias, workspace, (hipStream_t)stream);   // wino.h, F(3x3, 2x2)
  if (wino22_wgrad_planned(d)) return launch_wino22_wgrad(d, x, gy, dwp, dbias, workspace, (hipStream_t)stream);   // wino22.h
  if (wgrad_c32_ok(d)) {   // (C = K = 32: vec_ok holds, so the operands were checked for 16-byte alignment above)
    // accumulator-stationary kernel for the 32 -> 32 channel 3x3 layers (wgrad_c32.h): one partial per block, summed
    // by the same fixed-order reduce as the split-K slabs
    const int blocks = wgrad_c32_blocks(d);
    const long long total = (long long)d->KH * d->KW * d->C * d->K;
    float* bias_ws = dbias ? workspace + (size_t)blocks * total : nullptr;
    rc = launch_wgrad_c32(d, x, gy, workspace, bias_ws, (hipStream_t)stream);
    if (rc) return rc;
    int R = 1;
    while (R < 64 && R * 2 <= blocks && (total / 4) * R < 65536) R <<= 1;
    long long rb = ((total / 4) * R + 255) / 256;
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)rb), dim3(256), 0, (hipStream_t)stream, workspace, dwp,
                       d->KH * d->KW * d->C, d->K, d->ldw, blocks, bias_ws, dbias, R);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  int bm, bn, splits, pps;
  IgemmArgs a{};
  wgrad_plan(d, &bm, &bn, &a.tiles_m, &a.tiles_n, &splits, &pps);
  a.A = x; a.B = gy; a.C = workspace; a.d = *d;
  a.M = d->KH * d->KW * d->C; a.Ncol = d->K; a.Kg = a.M;
  const bool fused_bias = dbias != nullptr && vec_ok(d, MODE_WGRAD);
  CONTRAD_ARG(dbias == nullptr || fused_bias);   // the fused bias gradient needs the float4 path
  a.bias_ws = fused_bias ? workspace + (size_t)splits * a.M * a.Ncol : nullptr;
  const long long P = (long long)d->N * d->Ho * d->Wo;
  CONTRAD_ARG(P < (1ll << 31) - 4096);
  a.P = (int)P; a.ptiles_per_split = pps;
  a.pixmajor = wgrad_pixmajor_ok(d, bm, pps) ? 1 : 0;
  rc = dispatch<MODE_WGRAD>(a, bm, bn, vec_ok(d, MODE_WGRAD), dim3(a.tiles_m * a.tiles_n, splits),
                            (hipStream_t)stream);
  if (rc) return rc;
  const long long total = (long long)a.M * a.Ncol;
  // split-lanes per element group: enough threads (~64 k) to cover the memory latency when the output is small
  int R = 1;
  while (R < 64 && R * 2 <= splits && (total / 4) * R < 65536) R <<= 1;
  long long blocks = ((total / 4) * R + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, workspace, dwp,
                     a.M, a.Ncol, d->ldw, splits, a.bias_ws, fused_bias ? dbias : nullptr, R);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
