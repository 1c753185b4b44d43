// Weight gradient of the 32 -> 32 channel 3x3 stride-1 convolutions (StyleGAN2's 512 x 512 level), included by
// igemm.hip.  Accumulator-stationary: dW is only 9 x 32 x 32 values, the whole of it lives in the accumulators of every
// wave for the life of the block.
//
// Why a second weight-gradient kernel.  On the implicit-GEMM engine this layer is a 288 x 32 GEMM over 12.6 M positions
// (48 images): with a 128 x 32 tile every 16-position K-tile needs 8 KB of x (gathered once per filter tap) + 2 KB of gy
// for 8 MFMAs per wave -- 12.8 FLOP per byte through the L1 / L2 path instead of the 32 of a 128 x 128 tile -- and the
// matrix pipe was busy 0.59 of the time at a normal clock (rocprofv3, profiles/r03_sg2_512_n1_pmc.json: 83 TF/s, 8.6 GB
// fetched for 3.2 GB of operands).  Here a block stages a (4 + 2) x (32 + 2) pixel halo tile of x and the 4 x 32 tile of
// gy in LDS ONCE (42.5 KB); each wave takes one tile row, and for every pair of neighbouring positions issues 9 MFMAs
// (one per tap: A = x at the shifted pixels, B = gy, both straight out of LDS as 64 consecutive floats per wave) into 9
// accumulator tiles: 1.6 loads of x per position instead of 9, 4.6 instead of 20 bytes per cycle and CU.
// Deterministic: a block walks a fixed contiguous range of tiles, writes ONE partial [288][32] (+ the bias-gradient
// column sums) that the engine's wgrad_reduce_kernel sums over blocks in fixed order.
#pragma once

#ifndef C32_YFAST
#define C32_YFAST 1     // tile walk: 0 = along rows, 1 = down columns (see conv_c32.h)
#endif

constexpr int WC_TH = 4, WC_TW = 32, WC_C = 32;
constexpr int WC_XS = 7 * 256 * 4;                            // floats of the x halo tile: 6 x 34 pixels x 32 channels = 6 528, padded to the
                                                             // 7 float4 pieces per thread the block stores without a branch
constexpr int WC_GS = WC_TH * WC_TW * WC_C;                  // floats of the gy tile
constexpr int WC_MAX_BLOCKS = 512;                           // 2 blocks per CU resident (144 AGPRs + the prefetch registers)

struct WgradC32Args {
  const float* x;      // (N, H, W, 32) dense
  const float* gy;     // (N, H, W, 32) dense
  float* ws;           // [blocks][288 * 32]
  float* bias_ws;      // [blocks][32] or NULL
  int N, H, W;
  int tiles_x, tiles_y;
  long long ntiles;
};

__global__ __launch_bounds__(256, 2) void wgrad_c32_kernel(const WgradC32Args a) {   // 2 blocks per CU: 144 AGPRs + <= 112 VGPRs
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Gs = smem + WC_XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NX = ((WC_TH + 2) * (WC_TW + 2) * (WC_C / 4) + 255) / 256;    // float4 of the x halo tile per thread
  constexpr int NG = (WC_TH * WC_TW * (WC_C / 4)) / 256;                      // float4 of the gy tile per thread
  constexpr int RPW = WC_TH / 4;                                              // tile rows per wave

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const long long per = (a.ntiles + gridDim.x - 1) / gridDim.x;
  const long long t0 = (long long)blockIdx.x * per;
  const long long t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
  const int tpi = a.tiles_x * a.tiles_y;

  // The next tile's global loads are issued into registers BEFORE the current tile is consumed from LDS: the ~4 us of
  // MFMAs per tile cover the memory latency inside the block (with load -> barrier -> compute -> barrier the matrix pipe
  // idled a third of the time: 102 TF/s).
  float4 rx[NX], rg[NG];
  // per-thread pieces of a tile, fixed for the life of the block: byte offset relative to the tile's top-left HALO pixel and
  // which image borders would make the piece padding (4 bits per piece: 0 top, 1 bottom, 2 left, 3 right).  The loads are
  // buffer loads whose offset becomes out-of-range (hardware zero fill) for padding: no branch per piece -- with `if (...)
  // load` the fetch was eleven basic blocks, the compiler waited for the queue inside it and spilled four of the prefetched
  // float4 right behind their loads (i.e. waited for them: the prefetch hid nothing)
  unsigned offx[NX];
  unsigned padx = 0;
  constexpr unsigned WC_OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + 256 * i;
    const int q = e & 7, p = e >> 3;
    const int pr = p / (WC_TW + 2), pc = p - pr * (WC_TW + 2);
    offx[i] = (e >= (WC_TH + 2) * (WC_TW + 2) * (WC_C / 4)) ? WC_OOB : (unsigned)(((pr * a.W + pc) * WC_C + q * 4) * 4);
    padx |= ((pr == 0 ? 1u : 0u) | (pr == WC_TH + 1 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == WC_TW + 1 ? 8u : 0u)) << (4 * i);
  }
  // gy pieces: e = tid + 256 i -> pixel (row i, column tid >> 3), channel quad tid & 7: one offset + i rows
  const unsigned offg0 = (unsigned)(((tid >> 3) * WC_C + (tid & 7) * 4) * 4);
  const unsigned grow = (unsigned)(a.W * WC_C * 4);
  auto fetch = [&](long long t) {
    const int n = (int)(t / tpi);
    const int r = (int)(t - (long long)n * tpi);
#if C32_YFAST
    const int tx = r / a.tiles_y, ty = r - tx * a.tiles_y;
#else
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
#endif
    const size_t org = ((size_t)n * a.H * a.W + (size_t)(ty * WC_TH) * a.W + tx * WC_TW) * WC_C;
    const unsigned edge = (ty == 0 ? 1u : 0u) | (ty == a.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) |
                          (tx == a.tiles_x - 1 ? 8u : 0u);                            // wave-uniform
    // (base of x: the tile's top-left halo pixel -- for the first tile of the tensor a pointer just below it, never dereferenced)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + org) - (size_t)(a.W + 1) * WC_C, 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy + org), 0, (int)0x80000000u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const unsigned v = ((padx >> (4 * i)) & edge) ? WC_OOB : offx[i];
      rx[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)v, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NG; ++i)
      rg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsg, (int)offg0, (int)((unsigned)i * grow), 0));
  };
  if (t0 < t1) fetch(t0);
  for (long long t = t0; t < t1; ++t) {
    __syncthreads();                                   // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<float4*>(Xs + (e >> 3) * WC_C + (e & 7) * 4) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<float4*>(Gs + (e >> 3) * WC_C + (e & 7) * 4) = rg[i];
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
    // this wave: tile rows wave * RPW ...; MFMA 32x32x2: A[c][pos] = x[pos + tap][c], B[pos][k] = gy[pos][k], pos = 2 neighbours
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr;
      const float* Grow = Gs + row * WC_TW * WC_C;
#pragma unroll 4
      for (int pp = 0; pp < WC_TW / 2; ++pp) {
        const float b = Grow[(2 * pp) * WC_C + lane];     // lanes 0..31: position 2pp, lanes 32..63: position 2pp + 1
        bsum += b;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float av = Xs[((row + kh) * (WC_TW + 2) + 2 * pp + kw) * WC_C + lane];
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[kh * 3 + kw], 0, 0, 0);
          }
      }
    }
  }

  // ---- block partial: sum the four waves' accumulators through LDS, tap by tap ----
  constexpr int TILE = WC_C * WC_C;                       // 1024 floats per tap
  float* out = a.ws + (size_t)blockIdx.x * 9 * TILE;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;   // C/D layout of the 32x32 MFMA: row = c, col = k
      smem[wave * TILE + row * WC_C + l31] = acc[t][r];
    }
    __syncthreads();
    for (int e = tid; e < TILE; e += 256)
      out[t * TILE + e] = (smem[e] + smem[TILE + e]) + (smem[2 * TILE + e] + smem[3 * TILE + e]);
  }
  if (a.bias_ws) {
    bsum += __shfl_xor(bsum, 32, 64);                     // even + odd positions of the pair
    __syncthreads();
    if (lane < 32) smem[wave * 32 + lane] = bsum;
    __syncthreads();
    if (tid < 32) a.bias_ws[(size_t)blockIdx.x * 32 + tid] = (smem[tid] + smem[32 + tid]) + (smem[64 + tid] + smem[96 + tid]);
  }
}

// 3x3, stride 1, pad 1, 32 -> 32 channels, dense NHWC, grid divisible into 4 x 32 tiles
inline bool wgrad_c32_ok(const contrad_conv_desc* d) {
  static const bool enabled = []() { const char* e = contrad_dev_env("CONTRAD_WGRAD_C32"); return !(e && e[0] == '0'); }();
  return enabled && d->C == 32 && d->K == 32 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
         d->ldx == 32 && d->ldy == 32 && d->ldw == 32 && (d->W % WC_TW) == 0 && (d->H % WC_TH) == 0 &&
         (long long)d->N * d->H * d->W >= 1 << 16;       // (small maps: the engine's split-K GEMM has enough reuse)
}

inline int wgrad_c32_blocks(const contrad_conv_desc* d) {
  const long long ntiles = (long long)d->N * (d->H / WC_TH) * (d->W / WC_TW);
  return (int)(ntiles < WC_MAX_BLOCKS ? ntiles : WC_MAX_BLOCKS);
}

inline int launch_wgrad_c32(const contrad_conv_desc* d, const float* x, const float* gy, float* ws, float* bias_ws,
                            hipStream_t stream) {
  static bool attr_set_dev[64] = {};   // per device; benign race: idempotent
  constexpr size_t smem = (size_t)(WC_XS + WC_GS) * sizeof(float);
  static_assert(smem >= 4 * WC_C * WC_C * sizeof(float), "the epilogue reuses the tile buffers");
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  bool& attr_set = attr_set_dev[dev_id & 63];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_c32_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  WgradC32Args a{};
  a.x = x; a.gy = gy; a.ws = ws; a.bias_ws = bias_ws;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.tiles_x = d->W / WC_TW; a.tiles_y = d->H / WC_TH;
  a.ntiles = (long long)d->N * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(wgrad_c32_kernel, dim3(wgrad_c32_blocks(d)), dim3(256), smem, stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
