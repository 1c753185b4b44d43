// EXPERIMENT, NOT PART OF THE LIBRARY (round 2).  Correct (7e-7 against the implicit-GEMM engine, forward and data
// gradient) and NOT faster: 100 vs 104 TF/s on the 48 x 512 x 512 x 32 layer although it executes 4.6 instead of 12.4
// non-MFMA instructions per MFMA -- which is the finding: instruction overhead is not what limits this layer
// (DESIGN.md section 7).  To try it again: copy into contrad_amd/csrc/, declare the two extern "C" entry points in
// include/contrad_hip.h, rebuild, run tools/experiments/strip_test.py.
//
// 3x3 / stride 1 / pad 1 convolution with 32 input and 32 output channels (StyleGAN2_512's 512x512 level: the first
// ResBlock conv of the discriminator, the last styled conv of the generator), forward and data gradient.
//
// Why not the implicit-GEMM engine: with Cin = 32 the contraction is only 9 * 32 = 288 deep.  A block of the lean loop
// is 18 K-tiles = 144 MFMAs per wave wrapped in ~2100 other instructions (prologue, per-tile scalar walk, epilogue), and
// every input pixel is pulled through the memory pipeline nine times, once per tap: 99 / 103 TF/s where the deep layers
// reach 130-145 (DESIGN.md section 7 has the ablation and the counters).  Here the block stays on the layer instead:
//
//   * a block owns a strip of 32 output columns and walks DOWN the image, four output rows per step (one per wave);
//   * the 3 x 3 x 32 x 32 filter lives in REGISTERS for the whole block (144 VGPRs per lane: lane (k = lane & 31,
//     half = lane >> 5) holds W[tap][8h + 4 half + j][k] -- exactly the B fragments the MFMAs want), so the B operand
//     costs no LDS traffic at all;
//   * per step the six input rows the four output rows touch (34 pixels with the halo, 32 channels) are staged ONCE into
//     LDS in the quad layout [row][c / 4][pixel][4]; a tap is just an address offset into that tile, and one
//     ds_read_b128 per lane feeds four MFMAs (k order permuted like the lean loop: k = 8h + 4 (lane >> 5) + j);
//   * the next step's rows are fetched into registers before the MFMAs of the current step and stored to the other LDS
//     buffer after them: one barrier per step of 4 x 144 MFMAs.
//
// Per MFMA: 0.25 LDS reads, ~0.05 global loads, 0.11 stores -- against 1.1 / 0.5 / 0.1 and 5 scalar instructions in the
// lean loop's 128 x 32 tile.  The data gradient of this conv is the same conv over gy with the taps flipped and the
// filter transposed, i.e. the same kernel with a different gather when the filter is loaded.
#include "common.h"
#include <stdlib.h>
#include "../../include/contrad_hip.h"

namespace {

constexpr int SC = 32;            // channels in = channels out
constexpr int SW = 32;            // output columns per strip
constexpr int SPX = SW + 2;       // staged pixels per row (halo)
constexpr int SROWS = 6;          // staged rows per step (4 output rows + halo)
constexpr int SQ = SC / 4;        // channel quads
constexpr int SBUF = SROWS * SQ * SPX * 4;   // floats per LDS buffer (6528 = 26 KB)
constexpr int SPIECES = (SROWS * SPX * SQ + 255) / 256;   // float4 per thread and step (7)

struct StripArgs {
  const float* x;        // FWD: input (N,H,W,32) ld ldx;  DGRAD: gy
  const float* wp;       // packed filter [(tap * 32 + cin)][cout], ld ldw
  const float* bias;     // FWD only, may be NULL
  const float* act_ref;  // DGRAD only, may be NULL (same layout as the output)
  float* y;
  int N, H, W, ldx, ldy, ldw;
  int rows_per_block;    // multiple of 4
  int dgrad;
  float slope, gain;
};

__global__ __launch_bounds__(256, 2) void conv3x3_c32_strip_kernel(const StripArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int strips = a.W / SW, segs = a.H / a.rows_per_block;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int strip = b % strips; b /= strips;
  const int seg = b % segs;
  const int n = b / segs;
  const int c0 = strip * SW, r_begin = seg * a.rows_per_block, r_end = r_begin + a.rows_per_block;

  // ---- the filter, as MFMA B fragments, in registers for the whole strip ----
  float wr[9][4][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 8 * h + 4 * lhi + j;                      // contraction channel of (h, half, j)
        wr[t][h][j] = a.dgrad ? a.wp[(size_t)((8 - t) * SC + l31) * a.ldw + c]      // W'[t][k][c_out] = W[8 - t][c_out][k]
                              : a.wp[(size_t)(t * SC + c) * a.ldw + l31];
      }

  // ---- staging: piece i of this thread is float4 number tid + 256 i of the 6 x 34 x 8 tile (quad fastest, so that a
  //      pixel's 128 bytes are read by 8 neighbouring lanes); offsets are recomputed per step instead of kept in
  //      registers -- the filter already holds 144 of them ----
  const float* img = a.x + (size_t)n * a.H * a.W * a.ldx;
  const int rowpitch = a.W * a.ldx;
  float4 pre[SPIECES];
  auto fetch = [&](int R) {                                     // rows R - 1 .. R + 4 -> registers (zeros outside the image)
#pragma unroll
    for (int i = 0; i < SPIECES; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (SPX * SQ), rem = idx - row * (SPX * SQ);
      const int px = rem >> 3, cq = rem & 7;
      const int gr = R - 1 + row, gc = c0 - 1 + px;
      const bool ok = idx < SROWS * SPX * SQ && gr >= 0 && gr < a.H && gc >= 0 && gc < a.W;
      pre[i] = ok ? *reinterpret_cast<const float4*>(img + (gr * rowpitch + gc * a.ldx + cq * 4))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < SPIECES; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (SPX * SQ), rem = idx - row * (SPX * SQ);
      const int px = rem >> 3, cq = rem & 7;
      if (idx < SROWS * SPX * SQ)
        *reinterpret_cast<float4*>(lds + buf * SBUF + ((row * SQ + cq) * SPX + px) * 4) = pre[i];
    }
  };

  // ---- fragment addresses: output row `wave` of the step, pixel l31, quad 2h + half; a tap adds (kh rows, kw pixels) ----
  const int rdA = ((wave * SQ + lhi) * SPX + l31) * 4;
  const float g1 = a.gain, g0 = a.gain * a.slope;
  const float bj = (!a.dgrad && a.bias) ? a.bias[l31] : 0.f;

  fetch(r_begin);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int R = r_begin; R < r_end; R += 4, buf ^= 1) {
    const bool more = R + 4 < r_end;
    if (more) fetch(R + 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* base = lds + buf * SBUF + rdA;
    // 36 groups of (one ds_read_b128, four MFMAs).  A single wave saturates the matrix pipe with a dependent MFMA chain
    // (tools/micro/mfma_chain.hip: 154 TF/s at one wave per SIMD), but only if it never waits for LDS: the read of
    // group g + 2 is issued before the MFMAs of group g.
    auto frag = [&](int g) {
      const int t = g >> 2, h = g & 3, kh = t / 3, kw = t - 3 * kh;
      return *reinterpret_cast<const float4*>(base + ((kh * SQ + 2 * h) * SPX + kw) * 4);
    };
    float4 q0 = frag(0), q1 = frag(1);
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      const float4 q = q0;
      q0 = q1;
      if (g + 2 < 36) q1 = frag(g + 2);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, wr[g >> 2][g & 3][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.y, wr[g >> 2][g & 3][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.z, wr[g >> 2][g & 3][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q.w, wr[g >> 2][g & 3][3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next step's rows go to LDS BEFORE this step's output is stored: vmcnt counts loads and stores alike, so a
    // wait for the prefetch placed after the stores would also wait for every store of the epilogue to be acknowledged
    if (more) stash(buf ^ 1);
    // ---- epilogue of this wave's output row: C layout col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 half ----
    {
      const size_t rowoff = ((size_t)(n * a.H + R + wave) * a.W + c0) * a.ldy + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pix = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const size_t off = rowoff + (size_t)pix * a.ldy;
        float v = acc[r];
        if (a.dgrad) {
          if (a.act_ref) v *= (a.act_ref[off] > 0.f) ? g1 : g0;
        } else {
          v += bj;
          v *= (v > 0.f) ? g1 : g0;
        }
        a.y[off] = v;
      }
    }
    __syncthreads();
  }
}

}  // namespace

// 1 when (C, K, kernel, stride, pad, extents, leading dimensions) fit the strip kernel
extern "C" int contrad_conv3x3_c32_strip_ok(const contrad_conv_desc* d) {
  static const bool enabled = []() { const char* e = getenv("CONTRAD_CONV_STRIP"); return !(e && e[0] == '0'); }();
  if (!enabled || !d) return 0;
  if (d->C != SC || d->K != SC || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
  if (d->W % SW || d->H % 4 || d->W < 64 || d->H < 64) return 0;
  if ((d->ldx & 3) || d->ldx < SC || d->ldy < SC || d->ldw < SC) return 0;
  if ((long long)d->N * d->H * d->W * (d->ldx > d->ldy ? d->ldx : d->ldy) >= (1ll << 31)) return 0;
  return 1;
}

// mode 0: y = gain * lrelu(conv(x) + bias);  mode 1: dx = conv^T(gy) * gain * lrelu'(act_ref)  (act_ref may be NULL)
extern "C" int contrad_conv3x3_c32_strip(const contrad_conv_desc* d, int mode, const float* in, const float* wp,
                                         const float* bias, const float* act_ref, float* out, float slope, float gain,
                                         contrad_stream_t stream) {
  CONTRAD_ARG(d && in && wp && out && (mode == 0 || mode == 1));
  CONTRAD_ARG(contrad_conv3x3_c32_strip_ok(d));
  CONTRAD_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
  StripArgs a{};
  a.x = in; a.wp = wp; a.bias = bias; a.act_ref = act_ref; a.y = out;
  a.N = d->N; a.H = d->H; a.W = d->W; a.ldw = d->ldw;
  a.ldx = mode == 0 ? d->ldx : d->ldy;       // the tensor the kernel READS (x for the forward, gy for the data gradient)
  a.ldy = mode == 0 ? d->ldy : d->ldx;       // ... and writes
  a.dgrad = mode; a.slope = slope; a.gain = gain;
  // rows per block: as long as possible (the filter load and the first fetch are per block) while the grid still covers
  // the chip a few times over: 2 blocks per CU resident
  int rpb = d->H;
  while (rpb > 32 && (rpb % 8) == 0 && (long long)d->N * (d->W / SW) * (d->H / rpb) < 2048) rpb /= 2;
  a.rows_per_block = rpb;
  const int grid = d->N * (d->W / SW) * (d->H / rpb);
  static bool attr_set = false;
  const size_t smem = 2 * (size_t)SBUF * sizeof(float);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c32_strip_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(conv3x3_c32_strip_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
