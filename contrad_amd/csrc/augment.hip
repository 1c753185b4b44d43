// Fused SimCLR augmentation (augment/__init__.py:106-122): RandomResizeCrop + HorizontalFlip (one bilinear
// gather; the flip is an exact column permutation composed into the crop's sampling grid), RandomApply(
// ColorJitter) (contrast + the reference's atan2-HSV jitter with its 255/360 hue quirk, in either order),
// RandomApply(gray) -- ONE read and ONE write of the 3N-image batch instead of ~25 elementwise launches.
// All random parameters are sampled on the host in the reference's RNG draw order and passed per sample:
//   params[n][0..3]  = theta00, theta11, theta02, theta12   (augment/spatial.py:140-143)
//   params[n][4]     = flip sign (+1 / -1)                  (spatial.py:89-90)
//   params[n][5]     = colour-jitter mask, [6] = contrast factor, [7..9] = f_h, f_s, f_v
//   params[n][10]    = gray mask, [11] = blur mask (used by the blur kernels only)
//   params[n][12..14]= cutout mask, cutout row centre, cutout column centre (used by the cutout kernel only)
//
// Small images (3*H*W*4 B <= 64 KiB: CIFAR 32x32, up to 64x64): one block per image, the image lives in LDS
// between stages so the per-channel contrast mean costs no extra HBM pass.
// Large images (AFHQ 512x512): a statistics pass (gather [+HSV] -> channel sums, the source stays hot in the
// 256 MiB Infinity Cache) followed by an apply pass that recomputes the gather; optional separable Gaussian
// blur (GaussianBlur, augment/__init__.py:53-78) with reflect padding, tiles staged in LDS.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

constexpr int NPARAM = CONTRAD_AUG_NPARAM;
constexpr float TWO_PI = 6.283185307179586f;
constexpr float SQRT3 = 1.7320508075688772f;

// ATen remainder for floats: fmod, then shift into the divisor's sign
__device__ __forceinline__ float py_mod(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
  return m;
}

// grid_sample(..., padding_mode='reflection', align_corners=False) coordinate handling (ATen GridSampler.h)
__device__ __forceinline__ float reflect_coord(float in, int size) {
  // reflect over [-0.5, size-0.5]  (twice_low = -1, twice_high = 2*size - 1)
  const float mn = -0.5f, span = (float)size;
  in = fabsf(in - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  const float r = (flips & 1) ? (span - extra + mn) : (extra + mn);
  return fminf((float)(size - 1), fmaxf(r, 0.f));
}

struct AugArgs {
  const float* x;       // NCHW (B,3,H,W)
  float* y;             // NCHW
  const float* params;  // [B][NPARAM]
  int B, H, W;
  int contrast_first, has_contrast;
};

// op order of ColorJitterLayer.transform (color_jitter.py:65-75): a launch argument, or -- contrast_first < 0 -- read
// from the parameter block (column 15), so that a captured hipGraph replays with each step's own draw
#define CF(a, pr) ((a).contrast_first < 0 ? ((pr)[15] != 0.f) : ((a).contrast_first != 0))

// bilinear sample of channel plane `src` (H x W) at output pixel (i, j) under the crop+flip of sample n
struct Sampler {
  int x0, y0, x1, y1;
  float wx0, wx1, wy0, wy1;
  bool vx0, vx1, vy0, vy1;
};
__device__ __forceinline__ Sampler make_sampler(const float* pr, int i, int j, int H, int W) {
  const int jj = (pr[4] < 0.f) ? (W - 1 - j) : j;
  const float xn = (2.f * jj + 1.f) / (float)W - 1.f;
  const float yn = (2.f * i + 1.f) / (float)H - 1.f;
  const float gx = pr[0] * xn + pr[2];
  const float gy = pr[1] * yn + pr[3];
  float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  ix = reflect_coord(ix, W);
  iy = reflect_coord(iy, H);
  Sampler s;
  const float fx = floorf(ix), fy = floorf(iy);
  s.x0 = (int)fx; s.y0 = (int)fy; s.x1 = s.x0 + 1; s.y1 = s.y0 + 1;
  s.wx1 = ix - fx; s.wx0 = 1.f - s.wx1;
  s.wy1 = iy - fy; s.wy0 = 1.f - s.wy1;
  s.vx0 = (unsigned)s.x0 < (unsigned)W; s.vx1 = (unsigned)s.x1 < (unsigned)W;
  s.vy0 = (unsigned)s.y0 < (unsigned)H; s.vy1 = (unsigned)s.y1 < (unsigned)H;
  return s;
}
__device__ __forceinline__ float sample_plane(const float* src, const Sampler& s, int W) {
  float v = 0.f;
  if (s.vy0 && s.vx0) v += src[s.y0 * W + s.x0] * (s.wx0 * s.wy0);   // nw
  if (s.vy0 && s.vx1) v += src[s.y0 * W + s.x1] * (s.wx1 * s.wy0);   // ne
  if (s.vy1 && s.vx0) v += src[s.y1 * W + s.x0] * (s.wx0 * s.wy1);   // sw
  if (s.vy1 && s.vx1) v += src[s.y1 * W + s.x1] * (s.wx1 * s.wy1);   // se
  return v;
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// RandomHSVFunction.forward (augment/color_jitter.py:83-95) with rgb2hsv / hsv2rgb of augment/utils.py
__device__ __forceinline__ void hsv_jitter(float& r, float& g, float& b, float fh, float fs, float fv) {
  const float cmax = fmaxf(r, fmaxf(g, b)), cmin = fminf(r, fminf(g, b));
  float hue = atan2f(SQRT3 * (g - b), 2.f * r - g - b);
  hue = py_mod(hue, TWO_PI) / TWO_PI;
  float sat = 1.f - cmin / (cmax + 1e-8f);
  float val = cmax;
  if (!isfinite(hue)) hue = 0.f;
  if (!isfinite(sat)) sat = 0.f;
  if (!isfinite(val)) val = 0.f;
  float h = hue + (fh * 255.f) / 360.f;
  h = py_mod(h, 1.f);
  h = clamp01(h);
  const float s = clamp01(sat * fs), v = clamp01(val * fv);
  const float c = v * s;
  const float h6 = h * 6.f;
  float k, t;
  k = py_mod(5.f + h6, 6.f); t = clamp01(fminf(k, 4.f - k)); r = v - c * t;
  k = py_mod(3.f + h6, 6.f); t = clamp01(fminf(k, 4.f - k)); g = v - c * t;
  k = py_mod(1.f + h6, 6.f); t = clamp01(fminf(k, 4.f - k)); b = v - c * t;
}

__device__ __forceinline__ void gray3(float& r, float& g, float& b) {
  const float l = 0.299f * r + 0.587f * g + 0.114f * b;
  r = l; g = l; b = l;
}

// ---------------- small images: one block per image, image resident in LDS ----------------
__global__ __launch_bounds__(256) void simclr_small_kernel(AugArgs a) {
  extern __shared__ __attribute__((aligned(16))) float img[];  // [3][H*W]
  __shared__ float red[16];
  __shared__ float mean[3];
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const float* pr = a.params + (size_t)n * NPARAM;
  const float* src = a.x + (size_t)n * 3 * HW;
  const bool jitter = pr[5] != 0.f, gray = pr[10] != 0.f;
  const float fc = pr[6], fh = pr[7], fs = pr[8], fv = pr[9];

  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int i = p / a.W, j = p - i * a.W;
    const Sampler s = make_sampler(pr, i, j, a.H, a.W);
    float r = sample_plane(src, s, a.W), g = sample_plane(src + HW, s, a.W), b = sample_plane(src + 2 * HW, s, a.W);
    if (jitter && !CF(a, pr)) hsv_jitter(r, g, b, fh, fs, fv);
    img[p] = r; img[HW + p] = g; img[2 * HW + p] = b;
  }
  __syncthreads();
  if (jitter) {
    if (a.has_contrast) {
      for (int c = 0; c < 3; ++c) {
        float s = 0.f;
        for (int p = threadIdx.x; p < HW; p += blockDim.x) s += img[c * HW + p];
        s = block_sum(s, red);
        if (threadIdx.x == 0) mean[c] = s / (float)HW;
      }
      __syncthreads();
    }
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
      float r = img[p], g = img[HW + p], b = img[2 * HW + p];
      if (a.has_contrast) {
        r = (r - mean[0]) * fc + mean[0]; g = (g - mean[1]) * fc + mean[1]; b = (b - mean[2]) * fc + mean[2];
      }
      r = clamp01(r); g = clamp01(g); b = clamp01(b);
      if (CF(a, pr)) hsv_jitter(r, g, b, fh, fs, fv);
      img[p] = r; img[HW + p] = g; img[2 * HW + p] = b;
    }
    __syncthreads();
  }
  float* dst = a.y + (size_t)n * 3 * HW;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    float r = img[p], g = img[HW + p], b = img[2 * HW + p];
    if (gray) gray3(r, g, b);
    dst[p] = r; dst[HW + p] = g; dst[2 * HW + p] = b;
  }
}

// ---------------- backward of the small-image pipeline (generator step: d loss / d input images) ----------------
// One block per image.  The crop+flip gather is separable (axis-aligned affine grid), so its transpose is
// dX = Wy^T * dC * Wx with sparse H x H / W x W interpolation matrices built in LDS -- deterministic, no atomics.
// Colour stages: RandomApply blends are selects; HSV is straight-through (RandomHSVFunction.backward,
// color_jitter.py:97-104); contrast y = clamp((x-mu)f+mu) gives dx = f*gm + (1-f)*mean(gm), gm = g*[0<=pre<=1];
// gray gives dx_c = w_c * sum_c' g_c'.  Forward intermediates are recomputed from the saved input.
__global__ __launch_bounds__(256) void simclr_small_bwd_kernel(AugArgs a, const float* __restrict__ gout,
                                                               float* __restrict__ gin) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[16];
  __shared__ float mean[3], gmean[3];
  const int n = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W;
  float* cimg = lds;               // [3][HW] forward crop(+hsv) output = contrast input
  float* g = cimg + 3 * HW;        // [3][HW] running gradient
  float* Wy = g + 3 * HW;          // [H][H]  Wy[i][y]
  float* Wx = Wy + H * H;          // [W][W]  Wx[j][x]
  float* T = Wx + W * W;           // [HW]
  const float* pr = a.params + (size_t)n * NPARAM;
  const float* src = a.x + (size_t)n * 3 * HW;
  const bool jitter = pr[5] != 0.f, gray = pr[10] != 0.f;
  const float fc = pr[6], fh = pr[7], fs = pr[8], fv = pr[9];

  for (int e = threadIdx.x; e < H * H + W * W; e += blockDim.x) Wy[e] = 0.f;
  __syncthreads();
  // interpolation matrices (row i of Wy: the <= 2 source rows of output row i; likewise columns)
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    const Sampler s = make_sampler(pr, i, 0, H, W);
    if (s.vy0) Wy[i * H + s.y0] += s.wy0;
    if (s.vy1) Wy[i * H + s.y1] += s.wy1;
  }
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    const Sampler s = make_sampler(pr, 0, j, H, W);
    if (s.vx0) Wx[j * W + s.x0] += s.wx0;
    if (s.vx1) Wx[j * W + s.x1] += s.wx1;
  }
  // forward recompute + load the incoming gradient (through the gray blend)
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int i = p / W, j = p - i * W;
    const Sampler s = make_sampler(pr, i, j, H, W);
    float r = sample_plane(src, s, W), gg = sample_plane(src + HW, s, W), b = sample_plane(src + 2 * HW, s, W);
    if (jitter && !CF(a, pr)) hsv_jitter(r, gg, b, fh, fs, fv);
    cimg[p] = r; cimg[HW + p] = gg; cimg[2 * HW + p] = b;
    const float* go = gout + (size_t)n * 3 * HW;
    float g0 = go[p], g1 = go[HW + p], g2 = go[2 * HW + p];
    if (gray) {
      const float sum = g0 + g1 + g2;
      g0 = 0.299f * sum; g1 = 0.587f * sum; g2 = 0.114f * sum;
    }
    g[p] = g0; g[HW + p] = g1; g[2 * HW + p] = g2;
  }
  __syncthreads();
  if (jitter) {   // uniform per block
    // HSV is straight-through in either order, so only the contrast stage transforms the gradient
    for (int c = 0; c < 3; ++c) {
      float sm = 0.f;
      if (a.has_contrast)
        for (int p = threadIdx.x; p < HW; p += blockDim.x) sm += cimg[c * HW + p];
      sm = block_sum(sm, red);
      if (threadIdx.x == 0) mean[c] = sm / (float)HW;
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
      float sm = 0.f;
      for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const float x = cimg[c * HW + p];
        const float pre = a.has_contrast ? (x - mean[c]) * fc + mean[c] : x;
        const float gm = (pre >= 0.f && pre <= 1.f) ? g[c * HW + p] : 0.f;   // torch.clamp backward
        g[c * HW + p] = gm;
        sm += gm;
      }
      sm = block_sum(sm, red);
      if (threadIdx.x == 0) gmean[c] = sm / (float)HW;
    }
    __syncthreads();
    if (a.has_contrast)
      for (int e = threadIdx.x; e < 3 * HW; e += blockDim.x) {
        const int c = e / HW;
        g[e] = fc * g[e] + (1.f - fc) * gmean[c];
      }
    __syncthreads();
  }
  // dX_c = Wy^T * G_c * Wx
  float* gi = gin + (size_t)n * 3 * HW;
  for (int c = 0; c < 3; ++c) {
    for (int e = threadIdx.x; e < HW; e += blockDim.x) {      // T[i][x] = sum_j G[i][j] Wx[j][x]
      const int i = e / W, x = e - i * W;
      float acc = 0.f;
      for (int j = 0; j < W; ++j) acc = fmaf(g[c * HW + i * W + j], Wx[j * W + x], acc);
      T[e] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HW; e += blockDim.x) {      // dX[y][x] = sum_i Wy[i][y] T[i][x]
      const int y = e / W, x = e - y * W;
      float acc = 0.f;
      for (int i = 0; i < H; ++i) acc = fmaf(Wy[i * H + y], T[i * W + x], acc);
      gi[c * HW + e] = acc;
    }
    __syncthreads();
  }
}

// ---------------- large images ----------------
// pass 1: per-(image, channel) sums of the contrast input; partial[n][blockIdx.y][3]
__global__ __launch_bounds__(256) void simclr_stats_kernel(AugArgs a, float* __restrict__ partial) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const float* pr = a.params + (size_t)n * NPARAM;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (pr[5] != 0.f && a.has_contrast) {   // only jittered samples need the means
    const float* src = a.x + (size_t)n * 3 * HW;
    for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < HW; p += gridDim.y * blockDim.x) {
      const int i = p / a.W, j = p - i * a.W;
      const Sampler s = make_sampler(pr, i, j, a.H, a.W);
      float r = sample_plane(src, s, a.W), g = sample_plane(src + HW, s, a.W), b = sample_plane(src + 2 * HW, s, a.W);
      if (!CF(a, pr)) hsv_jitter(r, g, b, pr[7], pr[8], pr[9]);
      s0 += r; s1 += g; s2 += b;
    }
  }
  s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    float* o = partial + ((size_t)n * gridDim.y + blockIdx.y) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}

// pass 2: recompute the gather and finish the colour pipeline
__global__ __launch_bounds__(256) void simclr_apply_kernel(AugArgs a, const float* __restrict__ partial, int nparts) {
  __shared__ float mean[3];
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const float* pr = a.params + (size_t)n * NPARAM;
  const bool jitter = pr[5] != 0.f, gray = pr[10] != 0.f;
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += partial[((size_t)n * nparts + q) * 3 + threadIdx.x];
    mean[threadIdx.x] = s / (float)HW;
  }
  __syncthreads();
  const float* src = a.x + (size_t)n * 3 * HW;
  float* dst = a.y + (size_t)n * 3 * HW;
  const float fc = pr[6], fh = pr[7], fs = pr[8], fv = pr[9];
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < HW; p += gridDim.y * blockDim.x) {
    const int i = p / a.W, j = p - i * a.W;
    const Sampler s = make_sampler(pr, i, j, a.H, a.W);
    float r = sample_plane(src, s, a.W), g = sample_plane(src + HW, s, a.W), b = sample_plane(src + 2 * HW, s, a.W);
    if (jitter) {
      if (!CF(a, pr)) hsv_jitter(r, g, b, fh, fs, fv);
      if (a.has_contrast) {
        r = (r - mean[0]) * fc + mean[0]; g = (g - mean[1]) * fc + mean[1]; b = (b - mean[2]) * fc + mean[2];
      }
      r = clamp01(r); g = clamp01(g); b = clamp01(b);
      if (CF(a, pr)) hsv_jitter(r, g, b, fh, fs, fv);
    }
    if (gray) gray3(r, g, b);
    dst[p] = r; dst[HW + p] = g; dst[2 * HW + p] = b;
  }
}

// ---------------- separable Gaussian blur with reflect padding (masked per sample) ----------------
// horizontal: tmp[n,c,i,j] = sum_t g[t] x[n,c,i,reflect(j+t-R)];  vertical likewise on tmp -> y.
// Un-masked samples are copied through (RandomApply select).  One block per (plane row-tile).
__device__ __forceinline__ int reflect_idx(int i, int n) {  // F.pad(mode='reflect'): no edge repeat
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

constexpr int BLUR_MAXK = 129;

__global__ __launch_bounds__(256) void blur_h_kernel(const float* __restrict__ x, float* __restrict__ tmp,
                                                     const float* __restrict__ params, int H, int W, int R,
                                                     const float* __restrict__ gk) {
  extern __shared__ __attribute__((aligned(16))) float row[];  // [W + 2R]
  __shared__ float g[BLUR_MAXK];
  const int plane = blockIdx.y, n = plane / 3, i = blockIdx.x;
  if (params[(size_t)n * NPARAM + 11] == 0.f) return;  // uniform per block
  const float* src = x + ((size_t)plane * H + i) * W;
  for (int t = threadIdx.x; t < 2 * R + 1; t += blockDim.x) g[t] = gk[t];
  for (int j = threadIdx.x; j < W + 2 * R; j += blockDim.x) row[j] = src[reflect_idx(j - R, W)];
  __syncthreads();
  float* dst = tmp + ((size_t)plane * H + i) * W;
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < 2 * R + 1; ++t) s = fmaf(g[t], row[j + t], s);
    dst[j] = s;
  }
}

// vertical pass over a column strip of 64 columns: block (64 x 4), LDS tile [(TH + 2R)][64]
constexpr int BLUR_TH = 64;
__global__ __launch_bounds__(256) void blur_v_kernel(const float* __restrict__ tmp, const float* __restrict__ x,
                                                     float* __restrict__ y, const float* __restrict__ params,
                                                     int H, int W, int R, const float* __restrict__ gk) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [(BLUR_TH + 2R)][64]
  __shared__ float g[BLUR_MAXK];
  const int plane = blockIdx.z, n = plane / 3;
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * BLUR_TH;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const bool masked = params[(size_t)n * NPARAM + 11] != 0.f;
  const size_t pbase = (size_t)plane * H * W;
  if (!masked) {
    if (x != y)
      for (int r = ty; r < BLUR_TH; r += 4) {
        const int i = i0 + r, j = j0 + tx;
        if (i < H && j < W) y[pbase + (size_t)i * W + j] = x[pbase + (size_t)i * W + j];
      }
    return;
  }
  for (int t = threadIdx.x; t < 2 * R + 1; t += blockDim.x) g[t] = gk[t];
  for (int r = ty; r < BLUR_TH + 2 * R; r += 4) {
    const int i = reflect_idx(i0 + r - R, H), j = j0 + tx;
    tile[r * 64 + tx] = (j < W) ? tmp[pbase + (size_t)i * W + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < BLUR_TH; r += 4) {
    const int i = i0 + r, j = j0 + tx;
    if (i < H && j < W) {
      float s = 0.f;
      for (int t = 0; t < 2 * R + 1; ++t) s = fmaf(g[t], tile[(r + t) * 64 + tx], s);
      y[pbase + (size_t)i * W + j] = s;
    }
  }
}

// ---------------- backward of the large-image pipeline (generator step at AFHQ size) ----------------
// Same mathematics as simclr_small_bwd_kernel, spread over four launches because nothing fits LDS:
//   (stats pass of the forward: channel means of the contrast input)
//   gm pass      : recompute the forward per pixel, gray backward, clamp mask -> GM, partial channel sums of GM
//   gather^T (x) : T[i][x]  = sum_j  G'[i][j] Wx[j][x],   G' = fc*GM + (1-fc)*mean(GM)  (contrast backward)
//   gather^T (y) : dX[y][x] = sum_i  Wy[i][y] T[i][x]
// The crop+flip gather is an axis-aligned affine grid, so Wx / Wy have <= 2 entries per row and the rows that touch a
// given source column form a short contiguous run: each thread inverts the affine map for its run (deterministic,
// no atomics).
struct Axis { int p0, p1; float w0, w1; bool v0, v1; };
__device__ __forceinline__ Axis axis_sample(float scale, float bias, int idx, int size) {
  const float n = (2.f * idx + 1.f) / (float)size - 1.f;
  const float g = scale * n + bias;
  float c = ((g + 1.f) * (float)size - 1.f) * 0.5f;
  c = reflect_coord(c, size);
  Axis a;
  const float f = floorf(c);
  a.p0 = (int)f; a.p1 = a.p0 + 1;
  a.w1 = c - f; a.w0 = 1.f - a.w1;
  a.v0 = (unsigned)a.p0 < (unsigned)size; a.v1 = (unsigned)a.p1 < (unsigned)size;
  return a;
}
// candidate output indices whose (unclamped) source coordinate lies within (src-1, src+1), widened by 2
__device__ __forceinline__ void axis_inverse_range(float scale, float bias, int src, int size, int& lo, int& hi) {
  if (!(scale > 1e-6f)) { lo = 0; hi = size - 1; return; }
  const float c0 = scale * 0.5f + (float)size * (bias + 1.f - scale) * 0.5f - 0.5f;
  lo = (int)floorf(((float)src - 1.f - c0) / scale) - 2;
  hi = (int)ceilf(((float)src + 1.f - c0) / scale) + 2;
  lo = lo < 0 ? 0 : lo;
  hi = hi > size - 1 ? size - 1 : hi;
}

__global__ __launch_bounds__(256) void simclr_bwd_gm_kernel(AugArgs a, const float* __restrict__ gout,
                                                            const float* __restrict__ partial, int nparts,
                                                            float* __restrict__ GM, float* __restrict__ partial2) {
  __shared__ float mean[3];
  __shared__ float red[16];
  const int n = blockIdx.x;
  const int HW = a.H * a.W;
  const float* pr = a.params + (size_t)n * NPARAM;
  const bool jitter = pr[5] != 0.f, gray = pr[10] != 0.f;
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += partial[((size_t)n * nparts + q) * 3 + threadIdx.x];
    mean[threadIdx.x] = s / (float)HW;
  }
  __syncthreads();
  const float* src = a.x + (size_t)n * 3 * HW;
  const float* go = gout + (size_t)n * 3 * HW;
  float* gm = GM + (size_t)n * 3 * HW;
  const float fc = pr[6];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < HW; p += gridDim.y * blockDim.x) {
    float g0 = go[p], g1 = go[HW + p], g2 = go[2 * HW + p];
    if (gray) {
      const float sum = g0 + g1 + g2;
      g0 = 0.299f * sum; g1 = 0.587f * sum; g2 = 0.114f * sum;
    }
    if (jitter) {
      const int i = p / a.W, j = p - i * a.W;
      const Sampler s = make_sampler(pr, i, j, a.H, a.W);
      float r = sample_plane(src, s, a.W), g = sample_plane(src + HW, s, a.W), b = sample_plane(src + 2 * HW, s, a.W);
      if (!CF(a, pr)) hsv_jitter(r, g, b, pr[7], pr[8], pr[9]);
      if (a.has_contrast) {
        r = (r - mean[0]) * fc + mean[0]; g = (g - mean[1]) * fc + mean[1]; b = (b - mean[2]) * fc + mean[2];
      }
      if (!(r >= 0.f && r <= 1.f)) g0 = 0.f;        // torch.clamp backward
      if (!(g >= 0.f && g <= 1.f)) g1 = 0.f;
      if (!(b >= 0.f && b <= 1.f)) g2 = 0.f;
      s0 += g0; s1 += g1; s2 += g2;
    }
    gm[p] = g0; gm[HW + p] = g1; gm[2 * HW + p] = g2;
  }
  s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    float* o = partial2 + ((size_t)n * gridDim.y + blockIdx.y) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}

// T[n,c,i,x] = sum_j G'[n,c,i,j] Wx[j][x]
__global__ __launch_bounds__(256) void simclr_bwd_gather_x_kernel(AugArgs a, const float* __restrict__ GM,
                                                                  const float* __restrict__ partial2, int nparts,
                                                                  float* __restrict__ T) {
  __shared__ float gmean[3];
  const int n = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W;
  const float* pr = a.params + (size_t)n * NPARAM;
  const bool contrast = pr[5] != 0.f && a.has_contrast;
  if (threadIdx.x < 3) {
    float s = 0.f;
    if (contrast)
      for (int q = 0; q < nparts; ++q) s += partial2[((size_t)n * nparts + q) * 3 + threadIdx.x];
    gmean[threadIdx.x] = s / (float)HW;
  }
  __syncthreads();
  const float fc = contrast ? pr[6] : 1.f;
  const bool flip = pr[4] < 0.f;
  for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < 3 * HW; e += gridDim.y * blockDim.x) {
    const int c = e / HW, rem = e - c * HW, i = rem / W, x = rem - i * W;
    int lo, hi;
    axis_inverse_range(pr[0], pr[2], x, W, lo, hi);
    const float* grow = GM + ((size_t)n * 3 + c) * HW + (size_t)i * W;
    const float add = (1.f - fc) * gmean[c];
    float acc = 0.f;
    for (int jj = lo; jj <= hi; ++jj) {
      const Axis ax = axis_sample(pr[0], pr[2], jj, W);
      float w = 0.f;
      if (ax.v0 && ax.p0 == x) w += ax.w0;
      if (ax.v1 && ax.p1 == x) w += ax.w1;
      if (w != 0.f) acc = fmaf(fc * grow[flip ? (W - 1 - jj) : jj] + add, w, acc);
    }
    T[((size_t)n * 3 + c) * HW + rem] = acc;
  }
}

// dX[n,c,y,x] = sum_i Wy[i][y] T[n,c,i,x]
__global__ __launch_bounds__(256) void simclr_bwd_gather_y_kernel(AugArgs a, const float* __restrict__ T,
                                                                  float* __restrict__ gin) {
  const int n = blockIdx.x;
  const int H = a.H, W = a.W, HW = H * W;
  const float* pr = a.params + (size_t)n * NPARAM;
  for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < 3 * HW; e += gridDim.y * blockDim.x) {
    const int c = e / HW, rem = e - c * HW, y = rem / W, x = rem - y * W;
    int lo, hi;
    axis_inverse_range(pr[1], pr[3], y, H, lo, hi);
    const float* tcol = T + ((size_t)n * 3 + c) * HW + x;
    float acc = 0.f;
    for (int i = lo; i <= hi; ++i) {
      const Axis ay = axis_sample(pr[1], pr[3], i, H);
      float w = 0.f;
      if (ay.v0 && ay.p0 == y) w += ay.w0;
      if (ay.v1 && ay.p1 == y) w += ay.w1;
      if (w != 0.f) acc = fmaf(tcol[(size_t)i * W], w, acc);
    }
    gin[((size_t)n * 3 + c) * HW + rem] = acc;
  }
}

// RandomApply(CutOut) (augment/spatial.py:152-181): zero a (length x length) window, clipped at the borders
__global__ void cutout_kernel(float* __restrict__ y, const float* __restrict__ params, int H, int W, int half) {
  const int n = blockIdx.y;
  const float* pr = params + (size_t)n * NPARAM;
  if (pr[12] == 0.f) return;
  const int hc = (int)pr[13], wc = (int)pr[14];
  const int side = 2 * half + 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 3 * side * side) return;
  const int c = e / (side * side), r = e - c * side * side;
  const int i = hc - half + r / side, j = wc - half + r % side;
  if ((unsigned)i < (unsigned)H && (unsigned)j < (unsigned)W) y[(((size_t)n * 3 + c) * H + i) * W + j] = 0.f;
}

// ---------------- adjoint of the masked separable Gaussian blur (reflect padding) ----------------
// forward (one axis): out[i] = sum_t g[t] in[reflect(i + t - R)].  Adjoint: din[y] = sum over the pre-images p of y
// under the reflection (p = y, p = -y for 1 <= y <= R, p = 2(n-1) - y for n-1-R <= y <= n-2) of the zero-padded
// correlation  F(p) = sum_{i in [0,n), |i-p| <= R} g[p - i + R] dout[i].  Applied along y first (the forward ran x
// then y), then along x.  Un-masked samples pass the gradient through.
__device__ __forceinline__ float blur_adj_at(const float* __restrict__ v, long long stride, int n, int p, int R,
                                             const float* g) {
  int i0 = p - R, i1 = p + R;
  i0 = i0 < 0 ? 0 : i0;
  i1 = i1 > n - 1 ? n - 1 : i1;
  float s = 0.f;
  for (int i = i0; i <= i1; ++i) s = fmaf(g[p - i + R], v[(long long)i * stride], s);
  return s;
}
__global__ __launch_bounds__(256) void blur_adj_kernel(const float* __restrict__ gin_, float* __restrict__ gout_,
                                                       const float* __restrict__ params, int H, int W, int R,
                                                       const float* __restrict__ gk, int vertical) {
  __shared__ float g[BLUR_MAXK];
  const int plane = blockIdx.y, n = plane / 3;
  const bool masked = params[(size_t)n * NPARAM + 11] != 0.f;
  for (int t = threadIdx.x; t < 2 * R + 1; t += blockDim.x) g[t] = gk[t];
  __syncthreads();
  const size_t pbase = (size_t)plane * H * W;
  const int HW = H * W;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < HW; e += gridDim.x * blockDim.x) {
    if (!masked) { gout_[pbase + e] = gin_[pbase + e]; continue; }
    const int y = e / W, x = e - y * W;
    const int len = vertical ? H : W, pos = vertical ? y : x;
    const float* base = vertical ? (gin_ + pbase + x) : (gin_ + pbase + (size_t)y * W);
    const long long stride = vertical ? W : 1;
    float s = blur_adj_at(base, stride, len, pos, R, g);
    if (pos >= 1 && pos <= R) s += blur_adj_at(base, stride, len, -pos, R, g);
    if (pos >= len - 1 - R && pos <= len - 2) s += blur_adj_at(base, stride, len, 2 * (len - 1) - pos, R, g);
    gout_[pbase + e] = s;
  }
}

}  // namespace

extern "C" long long contrad_simclr_workspace_bytes(int B, int H, int W) {
  if ((long long)3 * H * W * 4 <= 64 * 1024) return 16;
  const int nparts = cdiv(H * W, 256 * 16);
  return (long long)B * nparts * 3 * (long long)sizeof(float);
}

extern "C" int contrad_simclr_augment(const float* x, float* y, const float* params, int B, int H, int W,
                                      int contrast_first, int has_contrast, float* workspace,
                                      long long workspace_bytes, contrad_stream_t stream) {
  CONTRAD_ARG(x && y && params && B > 0 && H > 1 && W > 1 && x != y);
  AugArgs a{x, y, params, B, H, W, contrast_first, has_contrast};
  hipStream_t s = (hipStream_t)stream;
  const size_t img_bytes = (size_t)3 * H * W * sizeof(float);
  if (img_bytes <= 64 * 1024) {
    hipLaunchKernelGGL(simclr_small_kernel, dim3(B), dim3(256), img_bytes, s, a);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  CONTRAD_ARG(workspace && workspace_bytes >= contrad_simclr_workspace_bytes(B, H, W));
  const int nparts = cdiv(H * W, 256 * 16);
  hipLaunchKernelGGL(simclr_stats_kernel, dim3(B, nparts), dim3(256), 0, s, a, workspace);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(simclr_apply_kernel, dim3(B, nparts), dim3(256), 0, s, a, workspace, nparts);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" long long contrad_simclr_augment_bwd_workspace_bytes(int B, int H, int W) {
  const size_t smem = (size_t)(7 * H * W + H * H + W * W) * sizeof(float);
  if (smem <= 64 * 1024) return 16;
  const int nparts = cdiv(H * W, 256 * 16);
  return ((long long)B * nparts * 3 * 2 + (long long)B * 3 * H * W * 2) * (long long)sizeof(float);
}

extern "C" int contrad_simclr_augment_bwd(const float* x, const float* params, const float* grad_out,
                                          float* grad_in, int B, int H, int W, int contrast_first,
                                          int has_contrast, float* workspace, long long workspace_bytes,
                                          contrad_stream_t stream) {
  CONTRAD_ARG(x && params && grad_out && grad_in && B > 0 && H > 1 && W > 1);
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = (size_t)(7 * H * W + H * H + W * W) * sizeof(float);
  AugArgs a{x, nullptr, params, B, H, W, contrast_first, has_contrast};
  if (smem <= 64 * 1024) {   // small-image path (<= 44x44): one block per image, everything in LDS
    hipLaunchKernelGGL(simclr_small_bwd_kernel, dim3(B), dim3(256), smem, s, a, grad_out, grad_in);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  CONTRAD_ARG(workspace && workspace_bytes >= contrad_simclr_augment_bwd_workspace_bytes(B, H, W));
  const int nparts = cdiv(H * W, 256 * 16);
  float* partial = workspace;
  float* partial2 = partial + (size_t)B * nparts * 3;
  float* GM = partial2 + (size_t)B * nparts * 3;
  float* T = GM + (size_t)B * 3 * H * W;
  hipLaunchKernelGGL(simclr_stats_kernel, dim3(B, nparts), dim3(256), 0, s, a, partial);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(simclr_bwd_gm_kernel, dim3(B, nparts), dim3(256), 0, s, a, grad_out, partial, nparts, GM, partial2);
  CONTRAD_CHECK_LAUNCH();
  const int gy = cdiv(3 * H * W, 256 * 4);
  hipLaunchKernelGGL(simclr_bwd_gather_x_kernel, dim3(B, gy), dim3(256), 0, s, a, GM, partial2, nparts, T);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(simclr_bwd_gather_y_kernel, dim3(B, gy), dim3(256), 0, s, a, T, grad_in);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_cutout_masked(float* y, const float* params, int B, int H, int W, int length,
                                     contrad_stream_t stream) {
  CONTRAD_ARG(y && params && B > 0 && H > 0 && W > 0 && length > 0 && (length & 1));
  const int half = (length - 1) / 2;
  hipLaunchKernelGGL(cutout_kernel, dim3(cdiv(3 * length * length, 256), B), dim3(256), 0, (hipStream_t)stream, y, params,
                     H, W, half);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_gaussian_blur_masked_bwd(const float* grad_out, float* tmp, float* grad_in, const float* params,
                                                const float* kernel1d, int B, int H, int W, int radius,
                                                contrad_stream_t stream) {
  CONTRAD_ARG(grad_out && tmp && grad_in && params && kernel1d && B > 0 && H > 0 && W > 0);
  CONTRAD_ARG(radius >= 0 && 2 * radius + 1 <= BLUR_MAXK && radius < H && radius < W);
  hipStream_t s = (hipStream_t)stream;
  const int gx = cdiv(H * W, 256 * 2);
  hipLaunchKernelGGL(blur_adj_kernel, dim3(gx, B * 3), dim3(256), 0, s, grad_out, tmp, params, H, W, radius, kernel1d, 1);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(blur_adj_kernel, dim3(gx, B * 3), dim3(256), 0, s, tmp, grad_in, params, H, W, radius, kernel1d, 0);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_gaussian_blur_masked(const float* x, float* tmp, float* y, const float* params,
                                            const float* kernel1d, int B, int H, int W, int radius,
                                            contrad_stream_t stream) {
  CONTRAD_ARG(x && tmp && y && params && kernel1d && B > 0 && H > 0 && W > 0);
  CONTRAD_ARG(radius >= 0 && 2 * radius + 1 <= BLUR_MAXK && radius < H && radius < W);
  hipStream_t s = (hipStream_t)stream;
  const size_t smem_h = (size_t)(W + 2 * radius) * sizeof(float);
  hipLaunchKernelGGL(blur_h_kernel, dim3(H, B * 3), dim3(256), smem_h, s, x, tmp, params, H, W, radius, kernel1d);
  CONTRAD_CHECK_LAUNCH();
  const size_t smem_v = (size_t)(BLUR_TH + 2 * radius) * 64 * sizeof(float);
  hipLaunchKernelGGL(blur_v_kernel, dim3(cdiv(W, 64), cdiv(H, BLUR_TH), B * 3), dim3(256), smem_v, s, tmp, x, y,
                     params, H, W, radius, kernel1d);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
