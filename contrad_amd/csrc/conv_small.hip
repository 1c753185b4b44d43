// Direct (VALU) convolutions for the RGB ends of the networks, where one GEMM dimension is 3 and the
// layer is HBM-bound, not MFMA-bound (first conv / FromRGB AI 1.4-13 FLOP/B):
//
//   rgb_conv_fwd   : image NCHW (N,Cin<=4,H,W) -> NHWC (N,H,W,K); k x k, stride 1, "same" padding; fuses the
//                    discriminators' input rescale x*2-1 (sndcgan.py:123, stylegan2/discriminator.py:226),
//                    bias, leaky-relu and gain.  One read of the image, one write of the activation.
//   rgb_conv_wgrad : its weight + bias gradient (packed layout), two-stage deterministic reduction.
//   rgb_conv_dgrad : data gradient / transposed conv onto <=4 channels, NHWC (N,H,W,K) -> NCHW (N,C,H,W) with an
//                    optional bias + tanh + affine epilogue: the generator's last ConvTranspose2d + Tanh +
//                    0.5x+0.5 (sndcgan.py:37-38,47) and d(loss)/d(image) of the first D layer (G-step, R1).
//
// Thread mapping (fwd / wgrad): a pixel's K output channels are spread over K/4 lanes (float4 per lane, so a
// pixel's channels are written as one contiguous 4*K-byte run); a 256-thread block therefore works on
// 1024/K pixels at a time and walks a tile of TH rows x W columns whose input halo sits in LDS.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

#ifndef RGB_MIN_WAVES
#define RGB_MIN_WAVES 3   // 168 VGPRs, no spill -> 3 waves per SIMD (the weights of all 27 taps live in registers)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// v_pk_fma_f32: two fp32 FMAs per lane per issue slot -- these kernels sit at the VALU roofline as much as at the HBM one
// (27 taps x 4 channels of FMA per 16 B stored), so the packed form is what brings them under the store time
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int MAX_TAPS = 9;  // k <= 3
constexpr int MAX_CIN = 4;

struct RgbArgs {
  const float* img;  // NCHW
  const float* wp;   // packed [(tap*Cin + ci)][ldw]
  const float* bias;
  float* y;          // NHWC, ld = ldy
  int N, Cin, H, W, K, ldy, ldw, k, pad, TH;
  float in_scale, in_shift, slope, gain;
};

// stage rows [h0-pad, h0+TH+pad) x cols [-pad, W+pad) of image n, all Cin channels, transformed, zero outside
__device__ __forceinline__ void stage_halo(float* lds, const RgbArgs& a, int n, int h0) {
  const int HW = a.W + 2 * a.pad, HH = a.TH + 2 * a.pad;
  for (int e = threadIdx.x; e < a.Cin * HH * HW; e += blockDim.x) {
    const int ci = e / (HH * HW), r = (e / HW) % HH, c = e % HW;
    const int h = h0 - a.pad + r, w = c - a.pad;
    float v = 0.f;
    if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
      v = a.img[((size_t)(n * a.Cin + ci) * a.H + h) * a.W + w] * a.in_scale + a.in_shift;
    lds[e] = v;
  }
}

template <int KSZ, int CIN>
__global__ __launch_bounds__(256, RGB_MIN_WAVES) void rgb_conv_fwd_kernel(RgbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TAPS = KSZ * KSZ;
  const int tpp = a.K >> 2;              // lanes per pixel
  const int slots = blockDim.x / tpp;    // pixels in flight
  const int cg = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  const int tiles_h = cdiv_dev(a.H, a.TH);
  const int HW = a.W + 2 * a.pad, HH = a.TH + 2 * a.pad;

  float4 w[TAPS * CIN];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
      w[t * CIN + ci] = *reinterpret_cast<const float4*>(a.wp + (size_t)(t * CIN + ci) * a.ldw + cg * 4);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + cg * 4);

  for (int tile = blockIdx.x; tile < a.N * tiles_h; tile += gridDim.x) {
    const int n = tile / tiles_h, h0 = (tile % tiles_h) * a.TH;
    __syncthreads();
    stage_halo(lds, a, n, h0);
    __syncthreads();
    const int rows = min(a.TH, a.H - h0);
    for (int pix = slot; pix < rows * a.W; pix += slots) {
      const int r = pix / a.W, c = pix - r * a.W;
      f32x2 a01 = {b4.x, b4.y}, a23 = {b4.z, b4.w};
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int kh = t / KSZ, kw = t % KSZ;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float xv = lds[(ci * HH + r + kh) * HW + c + kw];
          const float4 ww = w[t * CIN + ci];
          const f32x2 xx = {xv, xv};
          a01 = pk_fma(xx, f32x2{ww.x, ww.y}, a01);
          a23 = pk_fma(xx, f32x2{ww.z, ww.w}, a23);
        }
      }
      float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
      acc.x = (acc.x > 0.f ? acc.x : acc.x * a.slope) * a.gain;
      acc.y = (acc.y > 0.f ? acc.y : acc.y * a.slope) * a.gain;
      acc.z = (acc.z > 0.f ? acc.z : acc.z * a.slope) * a.gain;
      acc.w = (acc.w > 0.f ? acc.w : acc.w * a.slope) * a.gain;
      *reinterpret_cast<float4*>(a.y + ((size_t)(n * a.H + h0 + r) * a.W + c) * a.ldy + cg * 4) = acc;
    }
  }
}

struct RgbWgradArgs {
  const float* img;
  const float* gy;   // NHWC gradient wrt the PRE-activation output, ld = ldy
  float* partial;    // [gridDim.x][(TAPS*Cin + 1) * K]   (last K entries: bias gradient)
  int N, Cin, H, W, K, ldy, k, pad, TH;
  float in_scale, in_shift;
};

template <int KSZ, int CIN>
__global__ __launch_bounds__(256, RGB_MIN_WAVES) void rgb_conv_wgrad_kernel(RgbWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TAPS = KSZ * KSZ;
  const int tpp = a.K >> 2;
  const int slots = blockDim.x / tpp;
  const int cg = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  const int tiles_h = cdiv_dev(a.H, a.TH);
  const int HW = a.W + 2 * a.pad, HH = a.TH + 2 * a.pad;
  RgbArgs h{};  // reuse the halo stager
  h.img = a.img; h.Cin = a.Cin; h.H = a.H; h.W = a.W; h.pad = a.pad; h.TH = a.TH;
  h.in_scale = a.in_scale; h.in_shift = a.in_shift;

  f32x2 acc[TAPS * CIN][2];
#pragma unroll
  for (int i = 0; i < TAPS * CIN; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
  float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int tile = blockIdx.x; tile < a.N * tiles_h; tile += gridDim.x) {
    const int n = tile / tiles_h, h0 = (tile % tiles_h) * a.TH;
    __syncthreads();
    stage_halo(lds, h, n, h0);
    __syncthreads();
    const int rows = min(a.TH, a.H - h0);
    const int npix = rows * a.W;
    // gy is the only stream from HBM here (one float4 per thread and pixel): fetched one iteration ahead, otherwise every
    // iteration stalls on the full memory latency with only 2 - 4 waves per SIMD to cover it (rocprofv3: 60 % of the
    // wave cycles waiting, 2.2 TB/s)
    const float* gbase = a.gy + (size_t)(n * a.H + h0) * a.W * a.ldy + cg * 4;
    float4 g_next = (slot < npix) ? *reinterpret_cast<const float4*>(gbase + (size_t)slot * a.ldy) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pix = slot; pix < npix; pix += slots) {
      const int r = pix / a.W, c = pix - r * a.W;
      const float4 g = g_next;
      if (pix + slots < npix) g_next = *reinterpret_cast<const float4*>(gbase + (size_t)(pix + slots) * a.ldy);
      accb.x += g.x; accb.y += g.y; accb.z += g.z; accb.w += g.w;
      const f32x2 g01 = {g.x, g.y}, g23 = {g.z, g.w};
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int kh = t / KSZ, kw = t % KSZ;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float xv = lds[(ci * HH + r + kh) * HW + c + kw];
          const f32x2 xx = {xv, xv};
          acc[t * CIN + ci][0] = pk_fma(xx, g01, acc[t * CIN + ci][0]);
          acc[t * CIN + ci][1] = pk_fma(xx, g23, acc[t * CIN + ci][1]);
        }
      }
    }
  }
  // reduce over the pixel slots of the block through LDS (fixed order), then one partial per block
  __syncthreads();
  float* out = a.partial + (size_t)blockIdx.x * (TAPS * a.Cin + 1) * a.K;
  float4* red = reinterpret_cast<float4*>(lds);  // [slots][tpp]
#pragma unroll
  for (int q = 0; q <= TAPS * CIN; ++q) {   // fully unrolled: acc[] stays in registers
    const bool is_bias = (q == TAPS * CIN);
    const int qi = q < TAPS * CIN ? q : 0;
    const float4 v = is_bias ? accb : make_float4(acc[qi][0].x, acc[qi][0].y, acc[qi][1].x, acc[qi][1].y);
    red[slot * tpp + cg] = v;
    __syncthreads();
    if (slot == 0) {
      float4 s = red[cg];
      for (int sl = 1; sl < slots; ++sl) {
        const float4 o = red[sl * tpp + cg];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      *reinterpret_cast<float4*>(out + (size_t)q * a.K + cg * 4) = s;
    }
    __syncthreads();
  }
}

// dwp[row][k] (row < TAPS*Cin, leading dim ldw) and db[k] from the per-block partials, fixed order.  16 elements x 16
// partial lanes per block: with 512 partials each lane walks 32 of them (4 lanes walking 128 each took 34 us for 7 MB).
__global__ __launch_bounds__(256) void rgb_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                               int rows, int K, float* __restrict__ dwp, int ldw,
                                                               float* __restrict__ db) {
  __shared__ float red[16][17];
  const int ex = threadIdx.x & 15, py = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + ex;
  const int E = (rows + 1) * K;
  float s = 0.f;
  if (e < E)
    for (int b = py; b < nblocks; b += 16) s += partial[(size_t)b * E + e];
  red[py][ex] = s;
  __syncthreads();
  if (py == 0 && e < E) {
    s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][ex];
    const int row = e / K, k = e - row * K;
    if (row < rows) dwp[(size_t)row * ldw + k] = s;
    else if (db) db[k] = s;
  }
}

struct RgbDgradArgs {
  const float* gy;   // NHWC (N,H,W,K), ld = ldy
  const float* wp;   // packed [(tap*C + c)][ldw]
  const float* bias; // [C] or NULL
  float* out;        // NCHW (N,C,H,W)
  const float* mod;      // optional [N][K] per-sample modulation of the input channels (ToRGB)
  const float* residual; // optional NCHW (N,C,H,W) added before the activation (ToRGB skip)
  int N, C, H, W, K, ldy, ldw, k, pad;
  int act;           // 0: none, 1: tanh
  float out_scale, out_shift;  // out = f(acc + bias) * out_scale + out_shift
};

// stride-1 transposed conv onto C <= 4 channels: out[n,c,h,w] = sum_{kh,kw,k} gy[n,h+p-kh,w+p-kw,k] wp[(kh,kw,c),k]
// K/4 lanes cooperate on one output pixel: each lane owns 4 input channels (one coalesced float4 per tap, so a
// pixel's K channels are read as one contiguous run), keeps its taps x C x 4 weights in registers, and the C partial
// sums are combined with xor-shuffles inside the lane group.  256/(K/4) pixels per block pass.
template <int KSZ, int CH>   // CH: channel chunks of 4 per lane (K = 4 * CH * lanes-per-pixel)
__global__ __launch_bounds__(256) void rgb_conv_dgrad_kernel(RgbDgradArgs a) {
  constexpr int TAPS = KSZ * KSZ;
  const int tpp = a.K / (4 * CH);           // lanes per pixel (power of two, <= 64)
  const int slots = blockDim.x / tpp;
  const int cg = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  float4 w[CH][TAPS * MAX_CIN];
#pragma unroll
  for (int q = 0; q < CH; ++q)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int c = 0; c < MAX_CIN; ++c)
        w[q][t * MAX_CIN + c] =
            (c < a.C) ? *reinterpret_cast<const float4*>(a.wp + (size_t)(t * a.C + c) * a.ldw + (q * tpp + cg) * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long total = (long long)a.N * a.H * a.W;
  for (long long pix0 = (long long)blockIdx.x * slots; pix0 < total; pix0 += (long long)gridDim.x * slots) {
    const long long pix = pix0 + slot;
    const bool live = pix < total;
    const long long pp = live ? pix : 0;
    const int wq = (int)(pp % a.W);
    const int h = (int)((pp / a.W) % a.H);
    const int n = (int)(pp / ((long long)a.W * a.H));
    float acc[MAX_CIN] = {0.f, 0.f, 0.f, 0.f};
    float4 md[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q)
      md[q] = a.mod ? *reinterpret_cast<const float4*>(a.mod + (size_t)n * a.K + (q * tpp + cg) * 4)
                    : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int kh = 0; kh < KSZ; ++kh) {
      const int ho = h + a.pad - kh;
#pragma unroll
      for (int kw = 0; kw < KSZ; ++kw) {
        const int wo = wq + a.pad - kw;
        const bool ok = live && (unsigned)ho < (unsigned)a.H && (unsigned)wo < (unsigned)a.W;
        const float m = ok ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const float4 g = *reinterpret_cast<const float4*>(
              a.gy + (ok ? ((size_t)(n * a.H + ho) * a.W + wo) * a.ldy + (q * tpp + cg) * 4 : 0));
#pragma unroll
          for (int c = 0; c < MAX_CIN; ++c) {
            const float4 ww = w[q][(kh * KSZ + kw) * MAX_CIN + c];
            acc[c] += m * (g.x * md[q].x * ww.x + g.y * md[q].y * ww.y + g.z * md[q].z * ww.z + g.w * md[q].w * ww.w);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < MAX_CIN; ++c)
      for (int o = tpp >> 1; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    if (live && cg < a.C) {
      float v = (cg == 0) ? acc[0] : (cg == 1) ? acc[1] : (cg == 2) ? acc[2] : acc[3];
      v += a.bias ? a.bias[cg] : 0.f;
      const size_t o = ((size_t)(n * a.C + cg) * a.H + h) * a.W + wq;
      if (a.residual) v += a.residual[o];
      if (a.act == 1) v = tanhf(v);
      a.out[o] = v * a.out_scale + a.out_shift;
    }
  }
}

// The 3x3 transposed conv onto <= 3 channels with ONE THREAD PER OUTPUT PIXEL (no per-sample modulation / residual: the
// generator's last ConvTranspose2d + Tanh and d loss / d image of D's first layer).  A block owns TH rows x W columns
// (TH * W <= 256 pixels); the gy halo tile goes through LDS in chunks of 32 channels with coalesced float4 loads (a
// pixel's channels are one contiguous run), pixel stride 36 floats = conflict-free ds_read_b128 across the lanes; the
// weights are wave-uniform (scalar loads); the NCHW stores of a wave are contiguous per channel.  The lane-group
// formulation above spends its time in xor-shuffle reductions and holds 144 weight registers per lane (192 us for the
// 512-image batch of the headline config = 9 % of the HBM roofline; a thread-per-pixel version reading gy straight
// from global memory is bound by the vector-L1 line rate -- 64 lines per load instruction -- at the same 165 us).
constexpr int DG_CK = 32;            // channels per LDS pass
constexpr int DG_PS = DG_CK + 4;     // padded pixel stride (floats)
__global__ __launch_bounds__(256) void rgb_dgrad3_tile_kernel(const float* __restrict__ gy, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int N, int C, int H, int W, int K, int ldy, int ldw, int TH,
                                                              int act, float out_scale, float out_shift) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [(TH + 2)][(W + 2)][DG_PS] + weights [9][3][DG_CK]
  const int tiles_h = cdiv_dev(H, TH);
  const int n = blockIdx.x / tiles_h, h0 = (blockIdx.x % tiles_h) * TH;
  const int HW2 = W + 2, HH2 = TH + 2;
  float* wl = tile + (size_t)HH2 * HW2 * DG_PS;     // this pass's weights: ds_read broadcasts stay in order with the
                                                     // tile reads (a scalar-load stream shares lgkmcnt with them and
                                                     // returns out of order: the kernel waited on every group)
  const int r = threadIdx.x / W, c = threadIdx.x - r * W;
  const bool live = threadIdx.x < TH * W && h0 + r < H;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += DG_CK) {
    __syncthreads();
    for (int e = threadIdx.x; e < HH2 * HW2 * (DG_CK / 4); e += blockDim.x) {
      const int q = e % (DG_CK / 4), p = e / (DG_CK / 4);
      const int pr = p / HW2, pc = p - pr * HW2;
      const int h = h0 - 1 + pr, w = pc - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
        v = *reinterpret_cast<const float4*>(gy + ((size_t)(n * H + h) * W + w) * ldy + k0 + q * 4);
      *reinterpret_cast<float4*>(tile + (size_t)p * DG_PS + q * 4) = v;
    }
    for (int e = threadIdx.x; e < 9 * 3 * (DG_CK / 4); e += blockDim.x) {
      const int q = e % (DG_CK / 4), tc = e / (DG_CK / 4);      // tc = tap * 3 + cc
      const int tap = tc / 3, cc = tc - tap * 3;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cc < C) v = *reinterpret_cast<const float4*>(wp + (size_t)(tap * C + cc) * ldw + k0 + q * 4);
      *reinterpret_cast<float4*>(wl + tc * DG_CK + q * 4) = v;
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          // out[h][w] += gy[h + 1 - kh][w + 1 - kw] * w[kh][kw]; tile row of gy row (h + 1 - kh) is (r + 2 - kh)
          const float4* g4 = reinterpret_cast<const float4*>(tile + (size_t)((r + 2 - kh) * HW2 + (c + 2 - kw)) * DG_PS);
          const float4* w4 = reinterpret_cast<const float4*>(wl + (kh * 3 + kw) * 3 * DG_CK);
#pragma unroll
          for (int q = 0; q < DG_CK / 4; ++q) {
            const float4 g = g4[q];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
              const float4 ww = w4[cc * (DG_CK / 4) + q];       // same address in every lane: LDS broadcast
              acc[cc] = fmaf(g.x, ww.x, fmaf(g.y, ww.y, fmaf(g.z, ww.z, fmaf(g.w, ww.w, acc[cc]))));
            }
          }
        }
    }
  }
  if (live) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      if (cc < C) {
        float v = acc[cc] + (bias ? bias[cc] : 0.f);
        if (act == 1) v = tanhf(v);
        out[((size_t)(n * C + cc) * H + h0 + r) * W + c] = v * out_scale + out_shift;
      }
    }
  }
}

int rgb_common_check(int N, int Cin, int H, int W, int K, int k, int ldy, int ldw) {
  CONTRAD_ARG(N > 0 && H > 0 && W > 0 && Cin == 3);  /* RGB images (nc = 3 everywhere in the reference) */
  CONTRAD_ARG(k == 1 || k == 3);
  CONTRAD_ARG(K >= 4 && (K & 3) == 0 && K <= 1024 && (1024 % K) == 0);
  CONTRAD_ARG(ldy >= K && (ldy & 3) == 0 && (ldw & 3) == 0 && ldw >= K);
  return 0;
}

// rows per tile: ~1024 output pixels per staging pass (a whole CIFAR image: one halo load + one barrier pair per image
// instead of four; at W = 512 two rows instead of one, which halves the 3-row halo traffic of the k3 kernels)
int pick_th(int W) { int th = 1024 / W; return th < 1 ? 1 : (th > 32 ? 32 : th); }

}  // namespace

extern "C" int contrad_rgb_conv_fwd(const float* img, const float* wp, const float* bias, float* y, int N,
                                    int Cin, int H, int W, int K, int k, int ldy, int ldw, float in_scale,
                                    float in_shift, float slope, float gain, contrad_stream_t stream) {
  int rc = rgb_common_check(N, Cin, H, W, K, k, ldy, ldw);
  if (rc) return rc;
  CONTRAD_ARG(img && wp && y);
  RgbArgs a{};
  a.img = img; a.wp = wp; a.bias = bias; a.y = y;
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.K = K; a.ldy = ldy; a.ldw = ldw; a.k = k; a.pad = k / 2;
  a.TH = pick_th(W);
  a.in_scale = in_scale; a.in_shift = in_shift; a.slope = slope; a.gain = gain;
  const int tiles = N * cdiv(H, a.TH);
  const int grid = tiles < 4096 ? tiles : 4096;
  const size_t smem = (size_t)Cin * (a.TH + 2 * a.pad) * (W + 2 * a.pad) * sizeof(float);
  CONTRAD_ARG(smem <= 64 * 1024);
  if (k == 3) hipLaunchKernelGGL((rgb_conv_fwd_kernel<3, 3>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((rgb_conv_fwd_kernel<1, 3>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

static int rgb_wgrad_grid(int N, int H, int W, int k) {
  const int tiles = N * cdiv(H, pick_th(W));
  // resident blocks per CU: 3 for the 3x3 instance (156 VGPRs), 8 for the 1x1 one -- one full round, no ragged second
  // one (512 blocks = 2 waves per SIMD could not cover the latency of the gy stream: 2.2 TB/s)
  const int cap = (k == 3) ? 768 : 2048;
  return tiles < cap ? tiles : cap;
}

extern "C" long long contrad_rgb_conv_wgrad_workspace_bytes(int N, int Cin, int H, int W, int K, int k) {
  return (long long)rgb_wgrad_grid(N, H, W, k) * (k * k * Cin + 1) * K * (long long)sizeof(float);
}

extern "C" int contrad_rgb_conv_wgrad(const float* img, const float* gy, float* dwp, float* dbias, int N,
                                      int Cin, int H, int W, int K, int k, int ldy, int ldw, float in_scale,
                                      float in_shift, float* workspace, long long workspace_bytes,
                                      contrad_stream_t stream) {
  int rc = rgb_common_check(N, Cin, H, W, K, k, ldy, ldw);
  if (rc) return rc;
  CONTRAD_ARG(img && gy && dwp && workspace);
  CONTRAD_ARG(workspace_bytes >= contrad_rgb_conv_wgrad_workspace_bytes(N, Cin, H, W, K, k));
  RgbWgradArgs a{};
  a.img = img; a.gy = gy; a.partial = workspace;
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.K = K; a.ldy = ldy; a.k = k; a.pad = k / 2; a.TH = pick_th(W);
  a.in_scale = in_scale; a.in_shift = in_shift;
  const int grid = rgb_wgrad_grid(N, H, W, k);
  size_t smem = (size_t)Cin * (a.TH + 2 * a.pad) * (W + 2 * a.pad) * sizeof(float);
  const size_t red = 256 * sizeof(float4);
  if (smem < red) smem = red;
  CONTRAD_ARG(smem <= 64 * 1024);
  if (k == 3) hipLaunchKernelGGL((rgb_conv_wgrad_kernel<3, 3>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((rgb_conv_wgrad_kernel<1, 3>), dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  CONTRAD_CHECK_LAUNCH();
  const int rows = k * k * Cin;
  hipLaunchKernelGGL(rgb_wgrad_reduce_kernel, dim3(cdiv((rows + 1) * K, 16)), dim3(256), 0,
                     (hipStream_t)stream, workspace, grid, rows, K, dwp, ldw, dbias);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_rgb_conv_dgrad(const float* gy, const float* wp, const float* bias, const float* mod,
                                      const float* residual, float* out, int N, int C, int H, int W, int K,
                                      int k, int ldy, int ldw, int act, float out_scale, float out_shift,
                                      contrad_stream_t stream) {
  CONTRAD_ARG(gy && wp && out && N > 0 && H > 0 && W > 0 && C >= 1 && C <= MAX_CIN);
  CONTRAD_ARG((k == 1 || k == 3) && K >= 16 && K <= 512 && (K & (K - 1)) == 0 && ldy >= K && (ldy & 3) == 0);
  CONTRAD_ARG(K <= 256 || k == 1);
  CONTRAD_ARG(ldw >= K && (ldw & 3) == 0 && (act == 0 || act == 1));
  RgbDgradArgs a{};
  a.gy = gy; a.wp = wp; a.bias = bias; a.out = out; a.mod = mod; a.residual = residual;
  a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.ldy = ldy; a.ldw = ldw; a.k = k; a.pad = k / 2;
  a.act = act; a.out_scale = out_scale; a.out_shift = out_shift;
  const long long total = (long long)N * H * W;
  if (k == 3 && !mod && !residual && C <= 3 && (K % DG_CK) == 0 && W <= 256) {   // LDS-tiled thread-per-pixel form
    const int TH = 256 / W < 1 ? 1 : (256 / W > H ? H : 256 / W);
    const size_t smem = ((size_t)(TH + 2) * (W + 2) * DG_PS + 9 * 3 * DG_CK) * sizeof(float);
    if (smem <= 64 * 1024) {
      hipLaunchKernelGGL(rgb_dgrad3_tile_kernel, dim3(N * cdiv(H, TH)), dim3(256), smem, (hipStream_t)stream, gy, wp, bias,
                         out, N, C, H, W, K, ldy, ldw, TH, act, out_scale, out_shift);
      CONTRAD_CHECK_LAUNCH();
      return 0;
    }
  }
  const int ch = (K > 256) ? 2 : 1;
  const int slots = 256 / (K / (4 * ch));
  long long grid = (total + slots - 1) / slots;
  if (grid > 4096) grid = 4096;
  if (k == 3) hipLaunchKernelGGL((rgb_conv_dgrad_kernel<3, 1>), dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (ch == 1) hipLaunchKernelGGL((rgb_conv_dgrad_kernel<1, 1>), dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((rgb_conv_dgrad_kernel<1, 2>), dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
