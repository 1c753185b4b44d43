// Forward and data gradient of the 32 -> 32 channel 3x3 stride-1 convolutions (StyleGAN2's 512 x 512 level), included by
// igemm.hip.  Weight-stationary: the whole filter (9 x 32 x 32 values) lives in the registers of every wave for the life
// of the block, 144 values per lane in the B-operand layout of the 32x32x2 MFMA.
//
// Why.  On the implicit-GEMM engine this layer is a GEMM with 32 output columns: a 128 x 32 tile is ONE MFMA tile per
// wave, every 16-deep K-tile costs a global load, an LDS round trip and a barrier for 8 MFMAs per wave, and the input
// pixels are gathered once per filter tap (FWD 107, DGRAD 110 TF/s at 48 images, rocprofv3 rows of round 3; the
// counters showed a busy texture-address path and a low effective clock).  Here a block stages a
// (4 + 2) x (32 + 2) pixel halo tile ONCE, channel-major in LDS (Xs[c][row][col]: the 32 lanes of an MFMA operand read 32
// consecutive pixels of one channel = 32 consecutive banks, a filter tap is an address offset), each wave owns one tile
// row and issues 144 MFMAs (9 taps x 16 channel pairs) between two barriers with ONE ds_read_b32 each and nothing else.
// The data gradient of a stride-1 pad-1 3x3 conv is the same convolution of gy with the filter flipped and its channel
// roles swapped: same kernel, a different gather when the weight registers are filled, the act' epilogue instead of
// bias + LeakyReLU.  The k order of the contraction is (tap, channel pair).
#pragma once

// Tile walk of the 32-channel kernels (conv_c32.h, wgrad_c32.h): 1 = a block walks DOWN a column of tiles, so the two upper
// halo rows of the next tile are the rows it has just read (0 = along rows: the vertical halo, half of a 4-row tile again,
// came back 16 tiles later and missed the L2).  Round 5 (profiles/r05_pmc_c32_walk.txt): FETCH_SIZE per launch 710 -> 522
// (FWD), 918 -> 663 (DGRAD), 1 589 -> 1 383 MiB (WGRAD), L2 hit rate 0.30 -> 0.41 / 0.51 -> 0.58; durations and the step
// are unchanged (61.87 vs 61.85 ms, five alternations) -- these kernels are not bound by their fabric traffic -- so the
// order is kept for the bytes it saves, not for time.
#ifndef C32_YFAST
#define C32_YFAST 1
#endif

constexpr int CC_TH = 4, CC_TW = 32, CC_C = 32;
constexpr int CC_PIX = (CC_TH + 2) * (CC_TW + 2);              // pixels of the halo tile
constexpr int CC_CS = CC_PIX + 1;                              // channel stride in LDS: odd, so that the transposing
                                                               // ds_write_b32 of 8 channel quads x 4 pixels per lane group
                                                               // (bank = (4q + k) * 13 + pixel mod 32) never conflicts
constexpr int CC_MAX_BLOCKS = 512;                             // 2 blocks per CU resident

struct ConvC32Args {
  const float* x;        // (N, H, W, 32) dense: the input (FWD) or gy (DGRAD)
  const float* wp;       // packed weight [(tap * 32 + cin)][ldw], cout contiguous
  float* y;              // (N, H, W, 32) dense: the output (FWD) or dx (DGRAD)
  const float* bias;     // FWD: [32] or NULL
  const float* addend;   // FWD: y's layout or NULL
  const float* act_ref;  // DGRAD: dx's layout or NULL
  float slope, gain;
  int N, H, W, ldw;
  int tiles_x, tiles_y;
  long long ntiles;
};

template <int MODE>   // MODE_FWD or MODE_DGRAD
__global__ __launch_bounds__(256, 2) void conv_c32_kernel(const ConvC32Args a) {
  __shared__ __attribute__((aligned(16))) float Xs[CC_C * CC_CS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int NX = (CC_PIX * (CC_C / 4) + 255) / 256;        // float4 of the halo tile per thread

  // B operand of MFMA (tap, pair j): B[k = hi][col = l31] = w[tap][cin = 2j + hi][cout = l31]; for the data gradient
  // "cin" is gy's channel (the filter's cout), "cout" is dx's channel (the filter's cin), and the tap is mirrored
  float wr[9 * 16];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      wr[t * 16 + j] = (MODE == MODE_FWD) ? a.wp[(size_t)(t * CC_C + 2 * j + hi) * a.ldw + l31]
                                          : a.wp[(size_t)((8 - t) * CC_C + l31) * a.ldw + 2 * j + hi];

  const long long per = (a.ntiles + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per;
  const long long t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
  const int tpi = a.tiles_x * a.tiles_y;

  float4 rx[NX];
  auto origin = [&](long long t, int& n, int& ty, int& tx) {
    n = (int)(t / tpi);
    const int r = (int)(t - (long long)n * tpi);
#if C32_YFAST        // a block walks DOWN a column of tiles: the next tile's two upper halo rows are the rows just read
    tx = r / a.tiles_y;
    ty = r - tx * a.tiles_y;
#else
    ty = r / a.tiles_x;
    tx = r - ty * a.tiles_x;
#endif
  };
  auto fetch = [&](long long t) {      // the next tile's global loads, issued before the current tile's MFMAs
    int n, ty, tx;
    origin(t, n, ty, tx);
    const float* xo = a.x + ((size_t)n * a.H * a.W + (size_t)(ty * CC_TH) * a.W + tx * CC_TW) * CC_C;
    const unsigned edge = (ty == 0 ? 1u : 0u) | (ty == a.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) |
                          (tx == a.tiles_x - 1 ? 8u : 0u) | 16u;                      // wave-uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      // piece e of the halo tile: channel quad e & 7 of pixel e >> 3 = (pr, pc); which image borders make it padding
      const int e = tid + 256 * i;
      const int p = e >> 3;
      const int pr = p / (CC_TW + 2), pc = p - pr * (CC_TW + 2);
      const unsigned pad = (p >= CC_PIX) ? 16u
                           : ((pr == 0 ? 1u : 0u) | (pr == CC_TH + 1 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == CC_TW + 1 ? 8u : 0u));
      rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((pad & edge) == 0) rx[i] = *reinterpret_cast<const float4*>(xo + ((pr - 1) * a.W + (pc - 1)) * CC_C + (e & 7) * 4);
    }
  };
  const float bj = (MODE == MODE_FWD && a.bias) ? a.bias[l31] : 0.f;
  const float g1 = a.gain, g0 = a.gain * a.slope;

  if (t0 < t1) fetch(t0);
  for (long long t = t0; t < t1; ++t) {
    __syncthreads();                                   // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i;
      if ((e >> 3) < CC_PIX) {
        float* dst = Xs + (e & 7) * 4 * CC_CS + (e >> 3);
        dst[0] = rx[i].x; dst[CC_CS] = rx[i].y; dst[2 * CC_CS] = rx[i].z; dst[3 * CC_CS] = rx[i].w;
      }
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);

    f32x16 acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
    // A[row = pixel l31][k = hi] = x[pixel + tap][cin = 2j + hi]
    const float* xw = Xs + hi * CC_CS + wave * (CC_TW + 2) + l31;
    // instruction order pinned: operand reads stay two MFMA pairs ahead of their use and no further (left alone the
    // scheduler hoists dozens of the 144 independent ds_reads and spills the weight registers)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const float a0 = xw[(2 * j) * CC_CS + kh * (CC_TW + 2) + kw];
          const float a1 = xw[(2 * j + 2) * CC_CS + kh * (CC_TW + 2) + kw];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wr[(kh * 3 + kw) * 16 + j], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wr[(kh * 3 + kw) * 16 + j + 1], acc0, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }

    // epilogue: acc[r] is pixel (r & 3) + 8 (r >> 2) + 4 hi of this wave's row, output channel l31: 128 B per pixel
    int n, ty, tx;
    origin(t, n, ty, tx);
    const size_t row0 = (((size_t)n * a.H + ty * CC_TH + wave) * a.W + tx * CC_TW + 4 * hi) * CC_C + l31;
    float* yo = a.y + row0;
    const float* ro = (MODE == MODE_FWD) ? (a.addend ? a.addend + row0 : nullptr) : (a.act_ref ? a.act_ref + row0 : nullptr);
    // (plain stores: a non-temporal form -- buffer stores with the nt policy, round 5 -- took this kernel from 241 to 255
    // VGPRs and doubled its branches; not worth the risk for outputs the 256 MB Infinity Cache cannot hold anyway)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = ((r & 3) + 8 * (r >> 2)) * CC_C;             // compile-time: an immediate offset of the access
      float v = acc0[r];
      if constexpr (MODE == MODE_FWD) {
        v += bj;
        v = (v > 0.f) ? v : v * a.slope;
        v *= a.gain;
        if (ro) v += ro[off];
      } else {
        if (ro) v *= (ro[off] > 0.f) ? g1 : g0;                    // (no act_ref: raw sums, like the engine's own epilogue)
      }
      yo[off] = v;
    }
  }
}

// 3x3, stride 1, pad 1, 32 -> 32 channels, dense NHWC in and out, the map divisible into 4 x 32 pixel tiles
inline bool conv_c32_ok(const contrad_conv_desc* d) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_CONV_C32"); return !(e && e[0] == '0'); }();
  return enabled && d->C == 32 && d->K == 32 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         d->ldx == 32 && d->ldy == 32 && d->ldw >= 32 && (d->W % CC_TW) == 0 && (d->H % CC_TH) == 0 &&
         (long long)d->N * d->H * d->W >= 1 << 16;       // (small maps: the engine's tiles fill the chip better)
}

inline int conv_c32_blocks(const contrad_conv_desc* d) {
  const long long ntiles = (long long)d->N * (d->H / CC_TH) * (d->W / CC_TW);
  return (int)(ntiles < CC_MAX_BLOCKS ? ntiles : CC_MAX_BLOCKS);
}

template <int MODE>
inline int launch_conv_c32(const contrad_conv_desc* d, const float* in, const float* wp, float* out, const float* bias,
                           const float* addend, const float* act_ref, float slope, float gain, hipStream_t stream) {
  ConvC32Args a{};
  a.x = in; a.wp = wp; a.y = out; a.bias = bias; a.addend = addend; a.act_ref = act_ref;
  a.slope = slope; a.gain = gain;
  a.N = d->N; a.H = d->H; a.W = d->W; a.ldw = d->ldw;
  a.tiles_x = d->W / CC_TW; a.tiles_y = d->H / CC_TH;
  a.ntiles = (long long)d->N * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((conv_c32_kernel<MODE>), dim3(conv_c32_blocks(d)), dim3(256), 0, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
