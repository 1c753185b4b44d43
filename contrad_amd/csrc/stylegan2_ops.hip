// HIP counterparts of the reference's two native CUDA ops (models/gan/stylegan2/op/):
//
//   contrad_upfirdn2d      <-> upfirdn2d_op.upfirdn2d   (op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:49-369)
//   contrad_fused_bias_act <-> fused.fused_bias_act     (op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-98)
//
// Same tensor contract as the reference: input [major, in_h, in_w, minor], FIR kernel [kh, kw], zero-insertion
// upsampling by (up_x, up_y), padding (may be negative), correlation with the FLIPPED kernel, decimation by
// (down_x, down_y).  The reference reshapes NCHW to (major = B*C, minor = 1); this build keeps activations NHWC
// and calls it with (major = B, minor = C), which makes the innermost dimension contiguous and float4-wide --
// the op is HBM-bound (a 4x4 FIR: 32 FLOP per 8 B), so the design goal is one coalesced read (taps served from
// L1/L2) and one coalesced write, not the reference's shared-memory tiling per (B*C) plane.
#include "common.h"
#include "../../include/contrad_hip.h"

namespace {

// Streaming accesses of the FIR kernels: activations of hundreds of MB read once and written once.  UF_NT (dev build
// knob: bit 0 loads, bit 1 stores) forces them non-temporal.  Round 5 (profiles/r05_hbm_nt.txt): non-temporal STORES lift
// the blur at 48 x 512^2 x 32 from 4.12 to 5.04 TB/s and its transpose with the act' epilogue from 4.17 to 5.24 (plain
// stores keep the written lines in the XCD's L2, where they evict the input rows the neighbouring threads re-read); in
// the StyleGAN2_512 step that is 62.79 -> 62.04 ms (five same-box alternations, profiles/r05_ab_stnt.txt) -- used for
// outputs of >= 256 MB (UpfirdnArgs.nt_store); smaller outputs are found in the caches by their consumer and stay plain.
// Non-temporal LOADS lose everywhere (the taps' re-reads miss): 4.12 -> 3.22 TB/s.
typedef float uf_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 uf_ld4(const float* p) {
#if defined(UF_NT) && (UF_NT & 1)
  const uf_v4f v = __builtin_nontemporal_load(reinterpret_cast<const uf_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void uf_st4(float* p, float4 v, bool nt) {
#if defined(UF_NT) && (UF_NT & 2)
  nt = true;
#endif
  if (nt) {           // (uniform over the launch)
    uf_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<uf_v4f*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

struct UpfirdnArgs {
  const float* in;
  const float* kernel;
  float* out;
  int major, in_h, in_w, minor;
  int kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int out_h, out_w;
  // fused epilogue (all optional): v += addend;  out = v;  out2 = v * (act_ref > 0 ? gain : slope * gain)
  const float* addend;
  const float* act_ref;
  float* out2;
  float slope, gain;
  // modulated-conv epilogue (contrad_upfirdn2d_modconv; mc_bias != NULL selects it):
  // out = sqrt2 * lrelu_0.2(v * mc_demod[m,c] + mc_noise_w[0] * mc_noise[m,oy,ox] + mc_bias[c]) * mc_post[m,c]
  const float* mc_demod;
  const float* mc_noise;
  const float* mc_noise_w;
  const float* mc_bias;
  const float* mc_post;
  int nt_store;      // output beyond the caches (>= 256 MB): non-temporal stores, see upfirdn2d_launch
};

// epilogue store of VW consecutive channels at flat element offset `off` of the output
template <int VW>
__device__ __forceinline__ void uf_store(const UpfirdnArgs& a, size_t off, float* v) {
  if (a.addend) {
    if (VW == 4) {
      const float4 t = uf_ld4(a.addend + off);
      v[0] += t.x; v[1 % VW] += t.y; v[2 % VW] += t.z; v[3 % VW] += t.w;
    } else {
      v[0] += a.addend[off];
    }
  }
  if (a.out) {
    if (VW == 4) uf_st4(a.out + off, make_float4(v[0], v[1 % VW], v[2 % VW], v[3 % VW]), a.nt_store != 0);
    else a.out[off] = v[0];
  }
  if (a.out2) {
    const float neg = a.slope * a.gain;
    float w[VW], rf[VW];
    if (VW == 4) {
      const float4 t = uf_ld4(a.act_ref + off);
      rf[0] = t.x; rf[1 % VW] = t.y; rf[2 % VW] = t.z; rf[3 % VW] = t.w;
    } else {
      rf[0] = a.act_ref[off];
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) w[i] = v[i] * (rf[i] > 0.f ? a.gain : neg);
    if (VW == 4) uf_st4(a.out2 + off, make_float4(w[0], w[1 % VW], w[2 % VW], w[3 % VW]), a.nt_store != 0);
    else a.out2[off] = w[0];
  }
}
__device__ __forceinline__ void uf_store4(const UpfirdnArgs& a, size_t off, float4 v) {
  float t[4] = {v.x, v.y, v.z, v.w};
  uf_store<4>(a, off, t);
}
// the modulated-conv epilogue on 4 channels c..c+3 of output pixel `pix` (= (m * out_h + oy) * out_w + ox): the same
// operations in the same order as modconv_epilogue_kernel (+ its post_scale), so the fused launch is bitwise the two passes
__device__ __forceinline__ void mc_store4(const UpfirdnArgs& a, int m, size_t pix, int c, float4 v) {
  const float g = 1.4142135623730951f;
  if (a.mc_demod) {
    const float4 d = *reinterpret_cast<const float4*>(a.mc_demod + (size_t)m * a.minor + c);
    v.x *= d.x; v.y *= d.y; v.z *= d.z; v.w *= d.w;
  }
  const float nz = (a.mc_noise && a.mc_noise_w) ? a.mc_noise_w[0] * a.mc_noise[pix] : 0.f;
  const float4 b = *reinterpret_cast<const float4*>(a.mc_bias + c);
  v.x += nz + b.x; v.y += nz + b.y; v.z += nz + b.z; v.w += nz + b.w;
  v.x = (v.x > 0.f ? v.x : 0.2f * v.x) * g; v.y = (v.y > 0.f ? v.y : 0.2f * v.y) * g;
  v.z = (v.z > 0.f ? v.z : 0.2f * v.z) * g; v.w = (v.w > 0.f ? v.w : 0.2f * v.w) * g;
  if (a.mc_post) {
    const float4 q = *reinterpret_cast<const float4*>(a.mc_post + (size_t)m * a.minor + c);
    v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w;
  }
  uf_st4(a.out + pix * a.minor + c, v, a.nt_store != 0);
}

constexpr int MAX_FIR = 8;

template <int VW>  // vector width over `minor`: 4 (float4) or 1
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirdnArgs a) {
  __shared__ float fir[MAX_FIR * MAX_FIR];  // flipped
  for (int e = threadIdx.x; e < a.kh * a.kw; e += blockDim.x) {
    const int ky = e / a.kw, kx = e - ky * a.kw;
    fir[e] = a.kernel[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)];
  }
  __syncthreads();
  const int mv = a.minor / VW;
  const long long total = (long long)a.major * a.out_h * a.out_w * mv;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % mv) * VW;
    long long t = e / mv;
    const int ox = (int)(t % a.out_w);
    t /= a.out_w;
    const int oy = (int)(t % a.out_h);
    const int m = (int)(t / a.out_h);
    float acc[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) acc[v] = 0.f;
    for (int ky = 0; ky < a.kh; ++ky) {
      const int py = oy * a.down_y + ky - a.pad_y0;     // coordinate in the zero-inserted image
      if (py < 0 || py % a.up_y != 0) continue;
      const int iy = py / a.up_y;
      if (iy >= a.in_h) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int px = ox * a.down_x + kx - a.pad_x0;
        if (px < 0 || px % a.up_x != 0) continue;
        const int ix = px / a.up_x;
        if (ix >= a.in_w) continue;
        const float w = fir[ky * a.kw + kx];
        const float* src = a.in + (((size_t)m * a.in_h + iy) * a.in_w + ix) * a.minor + c;
        if (VW == 4) {
          const float4 x = *reinterpret_cast<const float4*>(src);
          acc[0] = fmaf(w, x.x, acc[0]); acc[1 % VW] = fmaf(w, x.y, acc[1 % VW]);
          acc[2 % VW] = fmaf(w, x.z, acc[2 % VW]); acc[3 % VW] = fmaf(w, x.w, acc[3 % VW]);
        } else {
          acc[0] = fmaf(w, src[0], acc[0]);
        }
      }
    }
    uf_store<VW>(a, (((size_t)m * a.out_h + oy) * a.out_w + ox) * a.minor + c, acc);
  }
}

// up = down = 1 fast path (Blur and its backward -- every upfirdn2d call of the discriminator): each thread produces a
// vertical strip of R outputs for one (x, 4 channels), so an input row it loads feeds up to kh outputs from
// registers: (R + kh - 1) * kw loads for R outputs instead of kh * kw each (7 vs 16 per output at R = 4).
template <int R>
__global__ __launch_bounds__(256) void upfirdn2d_strip_kernel(UpfirdnArgs a) {
  __shared__ float fir[MAX_FIR * MAX_FIR];  // flipped
  for (int e = threadIdx.x; e < a.kh * a.kw; e += blockDim.x) {
    const int ky = e / a.kw, kx = e - ky * a.kw;
    fir[e] = a.kernel[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)];
  }
  __syncthreads();
  const int mv = a.minor >> 2;
  const int strips = (a.out_h + R - 1) / R;
  const long long total = (long long)a.major * strips * a.out_w * mv;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % mv) * 4;
    long long t = e / mv;
    const int ox = (int)(t % a.out_w);
    t /= a.out_w;
    const int oy0 = (int)(t % strips) * R;
    const int m = (int)(t / strips);
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int iy0 = oy0 - a.pad_y0;                  // input row feeding (output row oy0, tap ky = 0)
    for (int dy = 0; dy < R + a.kh - 1; ++dy) {
      const int iy = iy0 + dy;
      if ((unsigned)iy >= (unsigned)a.in_h) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ix = ox + kx - a.pad_x0;
        if ((unsigned)ix >= (unsigned)a.in_w) continue;
        const float4 v = *reinterpret_cast<const float4*>(a.in + (((size_t)m * a.in_h + iy) * a.in_w + ix) * a.minor + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ky = dy - r;                     // tap row of this input row for output row oy0 + r
          if (ky >= 0 && ky < a.kh) {
            const float w = fir[ky * a.kw + kx];
            acc[r].x = fmaf(w, v.x, acc[r].x); acc[r].y = fmaf(w, v.y, acc[r].y);
            acc[r].z = fmaf(w, v.z, acc[r].z); acc[r].w = fmaf(w, v.w, acc[r].w);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (oy0 + r < a.out_h) uf_store4(a, (((size_t)m * a.out_h + oy0 + r) * a.out_w + ox) * a.minor + c, acc[r]);
  }
}

// ---- 4x4 FIR specialisations (every upfirdn2d call of the StyleGAN2 discriminator / generator uses [1,3,3,1]^2) ----
// Weights live in scalar registers (uniform loads of the flipped kernel), all tap loops are compile-time, each thread
// owns a small output tile so that an input value it loads feeds several outputs from registers.
struct Fir4 { float w[16]; };
__device__ __forceinline__ Fir4 load_fir4(const float* k) {      // flipped: correlation with k[3-ky][3-kx]
  Fir4 f;
#pragma unroll
  for (int e = 0; e < 16; ++e) f.w[e] = k[15 - e];
  return f;
}
__device__ __forceinline__ void fma4(float4& acc, float w, const float4& v) {
  acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}

// Block order of the 4x4-FIR kernels: thread tiles run channel-fastest, then x, then y, and vertically adjacent tiles share 3 of
// their 6 - 7 input rows.  The dispatcher puts block b on XCD b % 8 (own L2 each): in launch order the row above belongs to
// another XCD and the shared rows cross the fabric once per XCD (rocprofv3 FETCH_SIZE of the blur: 1.6x the tensor, of the
// 2x upsampling FIR: 3x).  With each XCD on a contiguous run of blocks instead (common.h xcd_remap; results unchanged), round 6,
// same box, 48 x 512^2 x 32: blur + decimation 386 -> 360 / 368 us, blur 641 / 653 -> 650 / 668, blur backward 908 / 914 -> 909 /
// 920, upsampling FIR 1197 / 1210 -> 1229 / 1238; StyleGAN2_512 step 47.73 -> 47.73 ms: the re-fetched rows come out of the
// 256 MB Infinity Cache and are not what bounds these kernels.  Kept for the decimating kernel only (tools/dev/ab_fir.sh).
template <bool XCD>
__device__ __forceinline__ int uf_block() {
  return XCD ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
}

// up = down = 1: tile of 4 (y) x 2 (x) outputs per thread and 4 channels: 7 x 5 loads feed 8 outputs (4.4 per output
// instead of 16).  Blur in front of the strided convs and its backward.
__global__ __launch_bounds__(256) void upfirdn4_u1d1_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 3) >> 2;
  const long long total = (long long)a.major * sy * sx * mv;
  const long long e = (long long)uf_block<false>() * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % mv) * 4;
  long long t = e / mv;
  const int ox0 = (int)(t % sx) * 2;
  t /= sx;
  const int oy0 = (int)(t % sy) * 4;
  const int m = (int)(t / sy);
  const int ix0 = ox0 - a.pad_x0, iy0 = oy0 - a.pad_y0;
  float4 acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r][0] = make_float4(0.f, 0.f, 0.f, 0.f); acc[r][1] = acc[r][0]; }
  const float* base = a.in + (size_t)m * a.in_h * a.in_w * a.minor + c;
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
    const int iy = iy0 + dy;
    const bool vy = (unsigned)iy < (unsigned)a.in_h;
    float4 v[5];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ix = ix0 + dx;
      v[dx] = (vy && (unsigned)ix < (unsigned)a.in_w)
                  ? uf_ld4(base + ((size_t)iy * a.in_w + ix) * a.minor)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ky = dy - r;
      if (ky >= 0 && ky < 4) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          fma4(acc[r][0], f.w[ky * 4 + kx], v[kx]);
          fma4(acc[r][1], f.w[ky * 4 + kx], v[kx + 1]);
        }
      }
    }
  }
  if (a.mc_bias) {       // uniform: the generator's upsampling StyledConv tail (blur -> demod + noise + bias + lrelu)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (oy0 + r < a.out_h && ox0 + q < a.out_w)
          mc_store4(a, m, ((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q, c, acc[r][q]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (oy0 + r < a.out_h && ox0 + q < a.out_w)
        uf_store4(a, (((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q) * a.minor + c, acc[r][q]);
}

// ---- branch-free forms of the three 4x4-FIR kernels (round 6) ----
// The kernels above guard every tap load and every store with its own bounds test: 75 loads in 210 exec-mask regions with 49
// vmcnt(0) waits (168 VGPRs) in the blur.  Here one image = grid.y, so the image base is wave-uniform and all accesses are raw
// buffer instructions on per-image resources: a tap outside the image is an out-of-range offset (the hardware returns zeros),
// a pixel outside the output an out-of-range store (dropped), an absent addend / act_ref / out / out2 a resource of zero
// records.  One basic block from the first load to the last store.  Same box, 48 x 512^2 x 32 (profiles/r06_ab_fir_branch_free.txt):
// blur 678 -> 594 us, its backward with act' 917 -> 813, decimating blur 386 -> 345, upsampling FIR 1197 -> 1099 (5.4 - 6.2 TB/s of
// algorithmic bytes); StyleGAN2_512 step 47.13 -> 46.61 ms, StyleGAN2-32 12.46 -> 12.36.  Absent operands are scalar branches:
// as zero-record resources their dropped loads and stores still cost address cycles (blur 650 -> 745 us).  The pointer forms above
// remain for tensors beyond 2^31 bytes per image / 65535 images.
constexpr unsigned UF_OOB = 0x80000000u;
// offset if ok, else out of range -- as a select on a value computed on every lane (left to itself the compiler sinks the address
// arithmetic into an exec-masked region per tap: dozens of tiny basic blocks)
__device__ __forceinline__ unsigned uf_sel(unsigned off, bool ok) {
  asm volatile("" : "+v"(off));
  return ok ? off : UF_OOB;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uf_rsrc(const float* base, size_t img_elems, int m, bool on) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(on ? base + img_elems * (size_t)m : base), 0,
                                           on ? (int)(unsigned)(img_elems * 4) : 0, 0x00020000);
}
__device__ __forceinline__ float4 uf_bld4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
template <int AUX>
__device__ __forceinline__ void uf_bst4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, (int)voff, 0, AUX);
}
// the fused epilogue of uf_store on NT output pixels (byte offsets voff[], UF_OOB = no such pixel) of one thread
template <int NT, int AUX>
__device__ __forceinline__ void uf_epilogue(const UpfirdnArgs& a, int m, size_t out_elems, const unsigned* voff, float4* acc) {
  const __amdgpu_buffer_rsrc_t rA = uf_rsrc(a.addend, out_elems, m, a.addend != nullptr);
  const __amdgpu_buffer_rsrc_t rR = uf_rsrc(a.act_ref, out_elems, m, a.out2 != nullptr);
  const __amdgpu_buffer_rsrc_t rO = uf_rsrc(a.out, out_elems, m, a.out != nullptr);
  const __amdgpu_buffer_rsrc_t rO2 = uf_rsrc(a.out2, out_elems, m, a.out2 != nullptr);
  const float neg = a.slope * a.gain;
  if (a.addend) {        // (uniform: scalar branches; an absent operand costs no memory instructions)
    float4 t[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) t[i] = uf_bld4(rA, voff[i]);
#pragma unroll
    for (int i = 0; i < NT; ++i) { acc[i].x += t[i].x; acc[i].y += t[i].y; acc[i].z += t[i].z; acc[i].w += t[i].w; }
  }
  if (a.out) {
#pragma unroll
    for (int i = 0; i < NT; ++i) uf_bst4<AUX>(rO, voff[i], acc[i]);
  }
  if (a.out2) {
    float4 rf[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) rf[i] = uf_bld4(rR, voff[i]);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float4 w;
      w.x = acc[i].x * (rf[i].x > 0.f ? a.gain : neg); w.y = acc[i].y * (rf[i].y > 0.f ? a.gain : neg);
      w.z = acc[i].z * (rf[i].z > 0.f ? a.gain : neg); w.w = acc[i].w * (rf[i].w > 0.f ? a.gain : neg);
      uf_bst4<AUX>(rO2, voff[i], w);
    }
  }
}

// (152 VGPRs = 3 waves per SIMD with all 35 taps in flight; forced to 4 / 5 waves the loads spill: 600 -> 1040 / 1830 us)
// up = down = 1 (blur and its backward): 4 x 2 outputs per thread and 4 channels, grid = (tiles of an image / 256, images)
#ifndef UF_TX
#define UF_TX 2      // output columns per thread of the blur (dev knob: 4 = 7 x 7 loads for 16 outputs, 3.1 instead of 4.4 loads per
                     // output but 221 registers = 2 waves per SIMD: 597 -> 639 us, backward 812 -> 844: the tap loads are not the limiter)
#endif
template <int TX>
__global__ __launch_bounds__(256) void upfirdn4_u1d1_buf_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + TX - 1) / TX, sy = (a.out_h + 3) >> 2;
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int m = (int)blockIdx.y;
  const bool live = e < sy * sx * mv;
  const int c = (e % mv) * 4;
  const int t = e / mv;
  const int ox0 = (t % sx) * TX, oy0 = (t / sx) * 4;
  const int ix0 = ox0 - a.pad_x0, iy0 = oy0 - a.pad_y0;
  const size_t in_elems = (size_t)a.in_h * a.in_w * a.minor, out_elems = (size_t)a.out_h * a.out_w * a.minor;
  const __amdgpu_buffer_rsrc_t rI = uf_rsrc(a.in, in_elems, m, true);
  float4 acc[4 * TX];
#pragma unroll
  for (int i = 0; i < 4 * TX; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int off0 = ((iy0 * a.in_w + ix0) * a.minor + c) * 4;
  const int rowb = a.in_w * a.minor * 4, colb = a.minor * 4;
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
    const bool vy = live && (unsigned)(iy0 + dy) < (unsigned)a.in_h;
    float4 v[TX + 3];
#pragma unroll
    for (int dx = 0; dx < TX + 3; ++dx)
      v[dx] = uf_bld4(rI, uf_sel((unsigned)(off0 + dy * rowb + dx * colb), vy && (unsigned)(ix0 + dx) < (unsigned)a.in_w));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ky = dy - r;
      if (ky >= 0 && ky < 4) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
          for (int q = 0; q < TX; ++q) fma4(acc[TX * r + q], f.w[ky * 4 + kx], v[kx + q]);
      }
    }
  }
  unsigned voff[4 * TX];
  const int oo0 = ((oy0 * a.out_w + ox0) * a.minor + c) * 4, orow = a.out_w * a.minor * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < TX; ++q)
      voff[TX * r + q] = uf_sel((unsigned)(oo0 + r * orow + q * colb), live && oy0 + r < a.out_h && ox0 + q < a.out_w);
  if (a.mc_bias) {       // uniform: the generator's upsampling StyledConv tail (blur -> demod + noise + bias + lrelu)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < TX; ++q)
        if (live && oy0 + r < a.out_h && ox0 + q < a.out_w)
          mc_store4(a, m, ((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q, c, acc[TX * r + q]);
    return;
  }
  if (a.nt_store) uf_epilogue<4 * TX, 2>(a, m, out_elems, voff, acc); else uf_epilogue<4 * TX, 0>(a, m, out_elems, voff, acc);
}

// up = 1, down = 2: out[oy][ox] = sum k[ky][kx] in[2 oy + ky - pad][2 ox + kx - pad]; tile of 2 x 2 outputs per thread:
// 6 x 6 loads feed 4 outputs (9 per output instead of 16).  Blur + decimation of the residual skip path.
__global__ __launch_bounds__(256) void upfirdn4_u1d2_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 1) >> 1;
  const long long total = (long long)a.major * sy * sx * mv;
  const long long e = (long long)uf_block<true>() * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % mv) * 4;
  long long t = e / mv;
  const int ox0 = (int)(t % sx) * 2;
  t /= sx;
  const int oy0 = (int)(t % sy) * 2;
  const int m = (int)(t / sy);
  const int ix0 = 2 * ox0 - a.pad_x0, iy0 = 2 * oy0 - a.pad_y0;
  float4 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) { acc[r][0] = make_float4(0.f, 0.f, 0.f, 0.f); acc[r][1] = acc[r][0]; }
  const float* base = a.in + (size_t)m * a.in_h * a.in_w * a.minor + c;
#pragma unroll
  for (int dy = 0; dy < 6; ++dy) {
    const int iy = iy0 + dy;
    const bool vy = (unsigned)iy < (unsigned)a.in_h;
    float4 v[6];
#pragma unroll
    for (int dx = 0; dx < 6; ++dx) {
      const int ix = ix0 + dx;
      v[dx] = (vy && (unsigned)ix < (unsigned)a.in_w)
                  ? uf_ld4(base + ((size_t)iy * a.in_w + ix) * a.minor)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ky = dy - 2 * r;
      if (ky >= 0 && ky < 4) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          fma4(acc[r][0], f.w[ky * 4 + kx], v[kx]);
          fma4(acc[r][1], f.w[ky * 4 + kx], v[kx + 2]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (oy0 + r < a.out_h && ox0 + q < a.out_w)
        uf_store4(a, (((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q) * a.minor + c, acc[r][q]);
}

// up = 2, down = 1 (zero-insertion upsampling: the backward of the decimating blur, and the generator's Upsample):
// out[oy][ox] = sum_{ky,kx} k[ky][kx] z[oy + ky - pad_y0][ox + kx - pad_x0],  z[2i][2j] = in[i][j], zero elsewhere.
// Each output sees a 2 x 2 subset of the taps; a thread produces the 2 x 2 output quad at (2 qy .. +1, 2 qx .. +1) from
// the <= 3 x 3 input pixels it can touch (9 loads for 4 outputs instead of 16 tap tests each).
// Which tap an (input pixel, output) pair uses depends only on the PARITY of the padding: with py_lo = 2 qy - pad_y0 and
// iy_lo = ceil(py_lo / 2), the tap row of input row iy_lo + dy for output row 2 qy + r is  ky = PY + 2 dy - r,  PY =
// pad_y0 & 1 -- the same for every thread of the launch.  PY / PX are template parameters, so every weight index is a
// compile-time constant and the 16 weights stay in scalar registers.  (Rounds 2 - 4 computed ky / kx per thread: a
// dynamically indexed private array, which the compiler promotes to LDS -- 16 KB per block, 36 ds_read_b32 per thread with
// an LDS bank-conflict fraction of 0.937 in every PMC summary.)
template <int PY, int PX>
__device__ __forceinline__ void upfirdn4_u2d1_body(const UpfirdnArgs& a, const Fir4& f, const float* base, int iy_lo,
                                                   int ix_lo, float4 (&acc)[2][2]) {
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = iy_lo + dy;
    const bool vy = (unsigned)iy < (unsigned)a.in_h;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ix_lo + dx;
      // (pixels that reach no output of the quad -- both tap indices out of range -- are not loaded at all)
      const bool used_y = (PY + 2 * dy >= 0 && PY + 2 * dy <= 3) || (PY + 2 * dy - 1 >= 0 && PY + 2 * dy - 1 <= 3);
      const bool used_x = (PX + 2 * dx >= 0 && PX + 2 * dx <= 3) || (PX + 2 * dx - 1 >= 0 && PX + 2 * dx - 1 <= 3);
      if (!(used_y && used_x)) continue;
      if (!(vy && (unsigned)ix < (unsigned)a.in_w)) continue;
      const float4 v = uf_ld4(base + ((size_t)iy * a.in_w + ix) * a.minor);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ky = PY + 2 * dy - r;
        if (ky < 0 || ky > 3) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int kx = PX + 2 * dx - q;
          if (kx < 0 || kx > 3) continue;
          fma4(acc[r][q], f.w[ky * 4 + kx], v);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void upfirdn4_u2d1_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 1) >> 1;
  const long long total = (long long)a.major * sy * sx * mv;
  const long long e = (long long)uf_block<false>() * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % mv) * 4;
  long long t = e / mv;
  const int ox0 = (int)(t % sx) * 2;
  t /= sx;
  const int oy0 = (int)(t % sy) * 2;
  const int m = (int)(t / sy);
  // zero-inserted coordinates touched by the quad: py in [oy0 - pad_y0, oy0 + 1 - pad_y0 + 3] (5 values) -> input rows
  // ceil(lo / 2) .. floor(hi / 2): at most 3
  const int py_lo = oy0 - a.pad_y0, px_lo = ox0 - a.pad_x0;
  const int iy_lo = (py_lo + 1) >> 1, ix_lo = (px_lo + 1) >> 1;         // arithmetic shift == floor for negatives
  float4 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) { acc[r][0] = make_float4(0.f, 0.f, 0.f, 0.f); acc[r][1] = acc[r][0]; }
  const float* base = a.in + (size_t)m * a.in_h * a.in_w * a.minor + c;
  // oy0 / ox0 are even: the parities are those of the paddings, uniform over the launch (a scalar branch)
  const int par = ((a.pad_y0 & 1) << 1) | (a.pad_x0 & 1);
  if (par == 0) upfirdn4_u2d1_body<0, 0>(a, f, base, iy_lo, ix_lo, acc);
  else if (par == 1) upfirdn4_u2d1_body<0, 1>(a, f, base, iy_lo, ix_lo, acc);
  else if (par == 2) upfirdn4_u2d1_body<1, 0>(a, f, base, iy_lo, ix_lo, acc);
  else upfirdn4_u2d1_body<1, 1>(a, f, base, iy_lo, ix_lo, acc);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (oy0 + r < a.out_h && ox0 + q < a.out_w)
        uf_store4(a, (((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q) * a.minor + c, acc[r][q]);
}

// branch-free forms of the decimating and the upsampling FIR (see upfirdn4_u1d1_buf_kernel): grid = (quads of an image / 256, images)
__global__ __launch_bounds__(256) void upfirdn4_u1d2_buf_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 1) >> 1;
  const int e = xcd_remap((int)blockIdx.x, (int)gridDim.x) * 256 + (int)threadIdx.x;
  const int m = (int)blockIdx.y;
  const bool live = e < sy * sx * mv;
  const int c = (e % mv) * 4;
  const int t = e / mv;
  const int ox0 = (t % sx) * 2, oy0 = (t / sx) * 2;
  const int ix0 = 2 * ox0 - a.pad_x0, iy0 = 2 * oy0 - a.pad_y0;
  const size_t in_elems = (size_t)a.in_h * a.in_w * a.minor, out_elems = (size_t)a.out_h * a.out_w * a.minor;
  const __amdgpu_buffer_rsrc_t rI = uf_rsrc(a.in, in_elems, m, true);
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int off0 = ((iy0 * a.in_w + ix0) * a.minor + c) * 4;
  const int rowb = a.in_w * a.minor * 4, colb = a.minor * 4;
#pragma unroll
  for (int dy = 0; dy < 6; ++dy) {
    const bool vy = live && (unsigned)(iy0 + dy) < (unsigned)a.in_h;
    float4 v[6];
#pragma unroll
    for (int dx = 0; dx < 6; ++dx)
      v[dx] = uf_bld4(rI, uf_sel((unsigned)(off0 + dy * rowb + dx * colb), vy && (unsigned)(ix0 + dx) < (unsigned)a.in_w));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ky = dy - 2 * r;
      if (ky >= 0 && ky < 4) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          fma4(acc[2 * r], f.w[ky * 4 + kx], v[kx]);
          fma4(acc[2 * r + 1], f.w[ky * 4 + kx], v[kx + 2]);
        }
      }
    }
  }
  unsigned voff[4];
  const int oo0 = ((oy0 * a.out_w + ox0) * a.minor + c) * 4, orow = a.out_w * a.minor * 4;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      voff[2 * r + q] = uf_sel((unsigned)(oo0 + r * orow + q * colb), live && oy0 + r < a.out_h && ox0 + q < a.out_w);
  if (a.nt_store) uf_epilogue<4, 2>(a, m, out_elems, voff, acc); else uf_epilogue<4, 0>(a, m, out_elems, voff, acc);
}

template <int PY, int PX>
__device__ __forceinline__ void upfirdn4_u2d1_buf_body(const UpfirdnArgs& a, const Fir4& f, __amdgpu_buffer_rsrc_t rI, bool live, int c,
                                                       int iy_lo, int ix_lo, float4* acc) {
  const int off0 = ((iy_lo * a.in_w + ix_lo) * a.minor + c) * 4;
  const int rowb = a.in_w * a.minor * 4, colb = a.minor * 4;
  float4 v[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      // (pixels that reach no output of the quad -- both tap indices out of range -- are not loaded at all)
      const bool used_y = (PY + 2 * dy >= 0 && PY + 2 * dy <= 3) || (PY + 2 * dy - 1 >= 0 && PY + 2 * dy - 1 <= 3);
      const bool used_x = (PX + 2 * dx >= 0 && PX + 2 * dx <= 3) || (PX + 2 * dx - 1 >= 0 && PX + 2 * dx - 1 <= 3);
      if (!(used_y && used_x)) continue;
      v[dy][dx] = uf_bld4(rI, uf_sel((unsigned)(off0 + dy * rowb + dx * colb),
                                     live && (unsigned)(iy_lo + dy) < (unsigned)a.in_h && (unsigned)(ix_lo + dx) < (unsigned)a.in_w));
    }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ky = PY + 2 * dy - r;
        if (ky < 0 || ky > 3) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int kx = PX + 2 * dx - q;
          if (kx < 0 || kx > 3) continue;
          fma4(acc[2 * r + q], f.w[ky * 4 + kx], v[dy][dx]);
        }
      }
}

__global__ __launch_bounds__(256) void upfirdn4_u2d1_buf_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int mv = a.minor >> 2;
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 1) >> 1;
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int m = (int)blockIdx.y;
  const bool live = e < sy * sx * mv;
  const int c = (e % mv) * 4;
  const int t = e / mv;
  const int ox0 = (t % sx) * 2, oy0 = (t / sx) * 2;
  const int py_lo = oy0 - a.pad_y0, px_lo = ox0 - a.pad_x0;
  const int iy_lo = (py_lo + 1) >> 1, ix_lo = (px_lo + 1) >> 1;         // arithmetic shift == floor for negatives
  const size_t in_elems = (size_t)a.in_h * a.in_w * a.minor, out_elems = (size_t)a.out_h * a.out_w * a.minor;
  const __amdgpu_buffer_rsrc_t rI = uf_rsrc(a.in, in_elems, m, true);
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int par = ((a.pad_y0 & 1) << 1) | (a.pad_x0 & 1);              // (uniform over the launch: a scalar branch)
  if (par == 0) upfirdn4_u2d1_buf_body<0, 0>(a, f, rI, live, c, iy_lo, ix_lo, acc);
  else if (par == 1) upfirdn4_u2d1_buf_body<0, 1>(a, f, rI, live, c, iy_lo, ix_lo, acc);
  else if (par == 2) upfirdn4_u2d1_buf_body<1, 0>(a, f, rI, live, c, iy_lo, ix_lo, acc);
  else upfirdn4_u2d1_buf_body<1, 1>(a, f, rI, live, c, iy_lo, ix_lo, acc);
  unsigned voff[4];
  const int oo0 = ((oy0 * a.out_w + ox0) * a.minor + c) * 4, orow = a.out_w * a.minor * 4, colo = a.minor * 4;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      voff[2 * r + q] = uf_sel((unsigned)(oo0 + r * orow + q * colo), live && oy0 + r < a.out_h && ox0 + q < a.out_w);
  if (a.nt_store) uf_epilogue<4, 2>(a, m, out_elems, voff, acc); else uf_epilogue<4, 0>(a, m, out_elems, voff, acc);
}

// The same op on single-channel planes (minor == 1: the generator's RGB skip is upsampled as B * 3 planes, generator.py:
// 121-143): one thread per 2 x 2 output quad, consecutive threads along x.  (Was the generic kernel: 16 tap tests with a
// division each per output, 7 launches of 35 us per StyleGAN2_512 step for 50 MB.)
template <int PY, int PX>
__device__ __forceinline__ void upfirdn4_u2d1_plane_body(const UpfirdnArgs& a, const Fir4& f, const float* base, int iy_lo,
                                                         int ix_lo, float (&acc)[2][2]) {
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = iy_lo + dy;
    const bool vy = (unsigned)iy < (unsigned)a.in_h;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ix_lo + dx;
      const bool used_y = (PY + 2 * dy <= 3) || (PY + 2 * dy - 1 >= 0 && PY + 2 * dy - 1 <= 3);
      const bool used_x = (PX + 2 * dx <= 3) || (PX + 2 * dx - 1 >= 0 && PX + 2 * dx - 1 <= 3);
      if (!(used_y && used_x)) continue;
      if (!(vy && (unsigned)ix < (unsigned)a.in_w)) continue;
      const float v = base[(size_t)iy * a.in_w + ix];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ky = PY + 2 * dy - r;
        if (ky < 0 || ky > 3) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int kx = PX + 2 * dx - q;
          if (kx < 0 || kx > 3) continue;
          acc[r][q] = fmaf(f.w[ky * 4 + kx], v, acc[r][q]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void upfirdn4_u2d1_planes_kernel(UpfirdnArgs a) {
  const Fir4 f = load_fir4(a.kernel);
  const int sx = (a.out_w + 1) >> 1, sy = (a.out_h + 1) >> 1;
  const long long total = (long long)a.major * sy * sx;
  const long long e = (long long)uf_block<false>() * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int ox0 = (int)(e % sx) * 2;
  const long long t = e / sx;
  const int oy0 = (int)(t % sy) * 2;
  const int m = (int)(t / sy);
  const int py_lo = oy0 - a.pad_y0, px_lo = ox0 - a.pad_x0;
  const int iy_lo = (py_lo + 1) >> 1, ix_lo = (px_lo + 1) >> 1;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const float* base = a.in + (size_t)m * a.in_h * a.in_w;
  const int par = ((a.pad_y0 & 1) << 1) | (a.pad_x0 & 1);
  if (par == 0) upfirdn4_u2d1_plane_body<0, 0>(a, f, base, iy_lo, ix_lo, acc);
  else if (par == 1) upfirdn4_u2d1_plane_body<0, 1>(a, f, base, iy_lo, ix_lo, acc);
  else if (par == 2) upfirdn4_u2d1_plane_body<1, 0>(a, f, base, iy_lo, ix_lo, acc);
  else upfirdn4_u2d1_plane_body<1, 1>(a, f, base, iy_lo, ix_lo, acc);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (oy0 + r < a.out_h && ox0 + q < a.out_w) {
        float v1[1] = {acc[r][q]};
        uf_store<1>(a, ((size_t)m * a.out_h + oy0 + r) * a.out_w + ox0 + q, v1);
      }
}

// y = act(x + b[(i / step_b) % size_b]) * scale  (grad 0) | x * act'(ref) * scale (grad 1) | 0 (grad 2)
// act 1: linear, act 3: leaky relu with slope alpha  -- the reference's act*10+grad switch
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                      const float* __restrict__ ref, float* __restrict__ y, long long n, int step_b,
                                      int size_b, int act, int grad, float alpha, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    const float r = ref ? ref[i] : 0.f;
    float o;
    if (act == 1) {
      o = (grad == 2) ? 0.f : v * scale;
    } else {
      if (grad == 0) o = (v > 0.f ? v : v * alpha) * scale;
      else if (grad == 1) o = (r > 0.f ? v : v * alpha) * scale;
      else o = 0.f;
    }
    y[i] = o;
  }
}

// y = a * x + b * z (elementwise; residual merge (out + skip) / sqrt(2), discriminator.py:72-74)
__global__ void lincomb_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y,
                               long long n, float a, float bc) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 u = reinterpret_cast<const float4*>(x)[i], v = reinterpret_cast<const float4*>(z)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(a * u.x + bc * v.x, a * u.y + bc * v.y, a * u.z + bc * v.z,
                                                  a * u.w + bc * v.w);
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + bc * z[i];
}

// Weight tables of all modulated convs in one launch (contrad_modconv_tables): block = one 32 (cout) x 32 (cin) tile of one
// layer, all T <= 9 taps; W[co][ci][t] rows are read contiguously (32 * T floats per cout), staged in LDS and written out
// as rows of the packed layout (32 consecutive columns per row piece) in either orientation, plus the tile of wsq.
struct ModconvMap { int start[CONTRAD_MODCONV_MAX_LAYERS + 1]; };
constexpr int MCT_TAPS = 9;

__global__ __launch_bounds__(256) void modconv_tables_kernel(contrad_modconv_batch b, ModconvMap map) {
  __shared__ float tile[32][32 * MCT_TAPS + 1];
  int l = 0;
  while (l + 1 < b.n && (int)blockIdx.x >= map.start[l + 1]) ++l;
  const contrad_modconv_layer& L = b.layers[l];
  const int blk = blockIdx.x - map.start[l];
  const int tiles_ci = (L.Cin + 31) / 32;
  const int co0 = (blk / tiles_ci) * 32, ci0 = (blk % tiles_ci) * 32;
  const int T = L.T;
  const int ncoi = min(32, L.Cout - co0), ncii = min(32, L.Cin - ci0);
  const int roww = ncii * T;                       // contiguous floats of one cout row inside the tile
  for (int e = threadIdx.x; e < 32 * roww; e += 256) {
    const int co = e / roww, r = e - co * roww;
    tile[co][r] = (co < ncoi) ? L.w[((size_t)(co0 + co) * L.Cin + ci0) * T + r] * L.scale : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < T * 1024; e += 256) {
    const int t = e >> 10, a = (e >> 5) & 31, q = e & 31;          // q = fastest index (the output column)
    if (!L.transposed) {                                             // row (t, ci = a), column co = q
      if (a < ncii && q < ncoi) L.wp[((size_t)t * L.Cin + ci0 + a) * L.ldw + co0 + q] = tile[q][a * T + t];
    } else {                                                         // row (t, co = a), column ci = q
      if (a < ncoi && q < ncii) L.wp[((size_t)t * L.Cout + co0 + a) * L.ldw + ci0 + q] = tile[a][q * T + t];
    }
  }
  if (L.wsq) {
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int ci = e >> 5, co = e & 31;
      if (ci < ncii && co < ncoi) {
        float ss = 0.f;
        for (int t = 0; t < T; ++t) { const float v = tile[co][ci * T + t]; ss = fmaf(v, v, ss); }
        L.wsq[(size_t)(ci0 + ci) * L.Cout + co0 + co] = ss;
      }
    }
  }
}

// Demodulation factors of all layers (contrad_modconv_demod): block = (layer, 64 output channels, 16 samples); the 16
// squared style rows sit in LDS transposed ([c][16]: broadcast ds_read_b128 x 4 per c); lane = output channel, the four
// waves of a block split the Cin range and each streams its columns of wsq with 16 independent loads in flight (a
// dependent load per c made the first version latency-bound: 120 us for 10 MB), 16 accumulators per thread; the four
// partial sums are added in wave order through LDS.
constexpr int DM_NB = 16, DM_MAXC = 512, DM_KB = 64;
__global__ __launch_bounds__(256) void modconv_demod_kernel(contrad_demod_batch b, ModconvMap map, float eps) {
  __shared__ float4 s2[DM_MAXC][DM_NB / 4];
  __shared__ float part[3][DM_NB][DM_KB];
  int l = 0;
  while (l + 1 < b.n && (int)blockIdx.x >= map.start[l + 1]) ++l;
  const contrad_demod_layer& L = b.layers[l];
  const int blk = blockIdx.x - map.start[l];
  const int kchunks = (L.K + DM_KB - 1) / DM_KB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = (blk % kchunks) * DM_KB + lane;
  const int n0 = (blk / kchunks) * DM_NB;
  const bool kok = k < L.K;
  float acc[DM_NB];
#pragma unroll
  for (int j = 0; j < DM_NB; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < L.Cin; c0 += DM_MAXC) {
    const int cn = min(DM_MAXC, L.Cin - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < cn * DM_NB; e += 256) {
      const int j = e / cn, c = e - j * cn;                       // consecutive threads: consecutive c of one sample
      const float v = (n0 + j < b.B) ? L.style[(size_t)(n0 + j) * L.Cin + c0 + c] : 0.f;
      reinterpret_cast<float*>(&s2[c][0])[j] = v * v;
    }
    __syncthreads();
    const int per = (cn + 3) / 4, cb = wave * per, ce = min(cn, cb + per);      // this wave's slice of the channels
    const float* wcol = L.wsq + (size_t)c0 * L.K + (kok ? k : 0);
    for (int c = cb; c < ce; c += 16) {
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = (c + u < ce) ? wcol[(size_t)(c + u) * L.K] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (c + u < ce) {
#pragma unroll
          for (int q = 0; q < DM_NB / 4; ++q) {
            const float4 v = s2[c + u][q];
            acc[4 * q] = fmaf(v.x, w[u], acc[4 * q]); acc[4 * q + 1] = fmaf(v.y, w[u], acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(v.z, w[u], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v.w, w[u], acc[4 * q + 3]);
          }
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < DM_NB; ++j) part[wave - 1][j][lane] = acc[j];
  }
  __syncthreads();
  if (wave == 0 && kok) {
#pragma unroll
    for (int j = 0; j < DM_NB; ++j) {
      const float t = ((acc[j] + part[0][j][lane]) + part[1][j][lane]) + part[2][j][lane];
      if (n0 + j < b.B) L.out[(size_t)(n0 + j) * L.K + k] = rsqrtf(t + eps);
    }
  }
}

// PixelNorm: one wave per row
__global__ void pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int K) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float ss = 0.f;
  for (int c = lane; c < K; c += 64) { const float v = x[(size_t)row * K + c]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)K + 1e-8f);
  for (int c = lane; c < K; c += 64) y[(size_t)row * K + c] = x[(size_t)row * K + c] * r;
}

__global__ void nhwc_scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y,
                                  int N, long long HW, int C) {
  const int c4n = C >> 2;
  const long long total = (long long)N * HW * c4n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    const int n = (int)(e / (HW * c4n));
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    const float4 m = *reinterpret_cast<const float4*>(s + (size_t)n * C + c4 * 4);
    reinterpret_cast<float4*>(y)[e] = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
  }
}

__global__ void modconv_epilogue_kernel(const float* __restrict__ x, const float* __restrict__ demod,
                                        const float* __restrict__ noise, const float* __restrict__ noise_w,
                                        const float* __restrict__ bias, const float* __restrict__ post,
                                        float* __restrict__ y, int N, long long HW, int K) {
  const int k4n = K >> 2;
  const long long total = (long long)N * HW * k4n;
  const float nw = (noise && noise_w) ? noise_w[0] : 0.f;
  const float g = 1.4142135623730951f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int k4 = (int)(e % k4n);
    const long long pix = e / k4n;           // n*HW + hw
    const int n = (int)(pix / HW);
    float4 v = reinterpret_cast<const float4*>(x)[e];
    if (demod) {
      const float4 d = *reinterpret_cast<const float4*>(demod + (size_t)n * K + k4 * 4);
      v.x *= d.x; v.y *= d.y; v.z *= d.z; v.w *= d.w;
    }
    const float nz = noise ? nw * noise[pix] : 0.f;
    const float4 b = *reinterpret_cast<const float4*>(bias + k4 * 4);
    v.x += nz + b.x; v.y += nz + b.y; v.z += nz + b.z; v.w += nz + b.w;
    v.x = (v.x > 0.f ? v.x : 0.2f * v.x) * g; v.y = (v.y > 0.f ? v.y : 0.2f * v.y) * g;
    v.z = (v.z > 0.f ? v.z : 0.2f * v.z) * g; v.w = (v.w > 0.f ? v.w : 0.2f * v.w) * g;
    if (post) {            // the consumer's weight modulation (its nhwc_scale pass) folded into this store
      const float4 q = *reinterpret_cast<const float4*>(post + (size_t)n * K + k4 * 4);
      v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w;
    }
    reinterpret_cast<float4*>(y)[e] = v;
  }
}

// out[n][c] = sum_hw a[n,hw,c] * b[n,hw,c]  (bcs = 1)  or  sum_hw a[n,hw,c] * b[n,hw]  (bcs = 0, b broadcast over c):
// the style / demodulation / noise-strength gradients of the modulated convolution (generator step).  Deterministic
// two-stage reduction: partial[n][s][c] over S row segments, then a fixed-order sum over s.
__global__ __launch_bounds__(256) void nhwc_dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               float* __restrict__ partial, long long HW, int C,
                                                               int bcs, int S, int lanes_c) {
  __shared__ float4 red[256];
  const int s = blockIdx.x, n = blockIdx.y;
  const int c4n = C >> 2;
  const int lc = threadIdx.x % lanes_c, lr = threadIdx.x / lanes_c, R = 256 / lanes_c;
  const long long chunk = (HW + S - 1) / S;
  const long long r0 = (long long)s * chunk, r1 = (r0 + chunk < HW) ? r0 + chunk : HW;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lc < c4n) {
    const float* an = a + (size_t)n * HW * C;
    for (long long r = r0 + lr; r < r1; r += R) {
      const float4 u = *reinterpret_cast<const float4*>(an + (size_t)r * C + lc * 4);
      if (bcs) {
        const float4 v = *reinterpret_cast<const float4*>(b + ((size_t)n * HW + r) * C + lc * 4);
        acc.x = fmaf(u.x, v.x, acc.x); acc.y = fmaf(u.y, v.y, acc.y);
        acc.z = fmaf(u.z, v.z, acc.z); acc.w = fmaf(u.w, v.w, acc.w);
      } else {
        const float v = b[(size_t)n * HW + r];
        acc.x = fmaf(u.x, v, acc.x); acc.y = fmaf(u.y, v, acc.y);
        acc.z = fmaf(u.z, v, acc.z); acc.w = fmaf(u.w, v, acc.w);
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (lr == 0 && lc < c4n) {
    float4 t = red[lc];
    for (int q = 1; q < R; ++q) {
      const float4 u = red[q * lanes_c + lc];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(partial + ((size_t)n * S + s) * C + lc * 4) = t;
  }
}

__global__ void nhwc_dot_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int N, int C, int S) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * C) return;
  const int n = e / C, c = e - n * C;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += partial[((size_t)n * S + s) * C + c];
  out[e] = t;
}

}  // namespace

extern "C" int contrad_pixelnorm(const float* x, float* y, int M, int K, contrad_stream_t stream) {
  CONTRAD_ARG(x && y && M > 0 && K > 0);
  hipLaunchKernelGGL(pixelnorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, y, M, K);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_nhwc_scale(const float* x, const float* s, float* y, int N, long long HW, int C,
                                  contrad_stream_t stream) {
  CONTRAD_ARG(x && s && y && N > 0 && HW > 0 && C > 0 && (C & 3) == 0);
  long long grid = ((long long)N * HW * (C / 4) + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(nhwc_scale_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, s, y, N, HW, C);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}


static int nhwc_dot_segments(int N, long long HW) {
  long long S = (HW + 63) / 64;
  const long long cap = (2048 + N - 1) / N;
  if (S > cap) S = cap;
  return (int)(S < 1 ? 1 : S);
}

extern "C" long long contrad_nhwc_dot_workspace_bytes(int N, long long HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return -22;
  return (long long)N * nhwc_dot_segments(N, HW) * C * (long long)sizeof(float);
}

extern "C" int contrad_nhwc_dot(const float* a, const float* b, float* out, int N, long long HW, int C,
                                int b_per_channel, float* workspace, long long workspace_bytes,
                                contrad_stream_t stream) {
  CONTRAD_ARG(a && b && out && N > 0 && HW > 0 && C > 0 && (C & 3) == 0 && C <= 1024);
  CONTRAD_ARG(workspace && workspace_bytes >= contrad_nhwc_dot_workspace_bytes(N, HW, C));
  const int S = nhwc_dot_segments(N, HW);
  int lanes_c = 1;
  while (lanes_c < C / 4) lanes_c <<= 1;
  hipLaunchKernelGGL(nhwc_dot_partial_kernel, dim3(S, N), dim3(256), 0, (hipStream_t)stream, a, b, workspace, HW, C,
                     b_per_channel ? 1 : 0, S, lanes_c);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(nhwc_dot_final_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, workspace, out,
                     N, C, S);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_modconv_tables(const contrad_modconv_batch* b, contrad_stream_t stream) {
  CONTRAD_ARG(b && b->n > 0 && b->n <= CONTRAD_MODCONV_MAX_LAYERS);
  ModconvMap map{};
  for (int l = 0; l < b->n; ++l) {
    const contrad_modconv_layer& L = b->layers[l];
    CONTRAD_ARG(L.w && L.wp && L.Cout > 0 && L.Cin > 0 && L.T > 0 && L.T <= MCT_TAPS);
    CONTRAD_ARG(L.ldw >= (L.transposed ? L.Cin : L.Cout));
    map.start[l + 1] = map.start[l] + cdiv(L.Cout, 32) * cdiv(L.Cin, 32);
  }
  hipLaunchKernelGGL(modconv_tables_kernel, dim3(map.start[b->n]), dim3(256), 0, (hipStream_t)stream, *b, map);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_modconv_demod(const contrad_demod_batch* b, float eps, contrad_stream_t stream) {
  CONTRAD_ARG(b && b->n > 0 && b->n <= CONTRAD_MODCONV_MAX_LAYERS && b->B > 0);
  ModconvMap map{};
  for (int l = 0; l < b->n; ++l) {
    const contrad_demod_layer& L = b->layers[l];
    CONTRAD_ARG(L.style && L.wsq && L.out && L.Cin > 0 && L.K > 0);
    map.start[l + 1] = map.start[l] + cdiv(L.K, DM_KB) * cdiv(b->B, DM_NB);
  }
  hipLaunchKernelGGL(modconv_demod_kernel, dim3(map.start[b->n]), dim3(256), 0, (hipStream_t)stream, *b, map, eps);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_modconv_epilogue(const float* x, const float* demod, const float* noise,
                                        const float* noise_w, const float* bias, const float* post_scale, float* y,
                                        int N, long long HW, int K, contrad_stream_t stream) {
  CONTRAD_ARG(x && bias && y && N > 0 && HW > 0 && K > 0 && (K & 3) == 0);
  long long grid = ((long long)N * HW * (K / 4) + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(modconv_epilogue_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, demod, noise,
                     noise_w, bias, post_scale, y, N, HW, K);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

struct ModconvEpi { const float *demod, *noise, *noise_w, *bias, *post; };

static int upfirdn2d_launch(const float* input, const float* kernel, float* out, int major, int in_h, int in_w,
                            int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                            int pad_x1, int pad_y0, int pad_y1, const float* addend, const float* act_ref, float slope,
                            float gain, float* out2, contrad_stream_t stream, const ModconvEpi* mc = nullptr) {
  CONTRAD_ARG(input && kernel && (out || out2) && major > 0 && in_h > 0 && in_w > 0 && minor > 0);
  CONTRAD_ARG(kh > 0 && kw > 0 && kh <= MAX_FIR && kw <= MAX_FIR);
  CONTRAD_ARG(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0);
  CONTRAD_ARG(!out2 || act_ref);
  UpfirdnArgs a{};
  a.in = input; a.kernel = kernel; a.out = out;
  a.major = major; a.in_h = in_h; a.in_w = in_w; a.minor = minor;
  a.kh = kh; a.kw = kw; a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
  a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
  a.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;   // op/upfirdn2d_kernel.cu:227-228
  a.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  a.addend = addend; a.act_ref = act_ref; a.out2 = out2; a.slope = slope; a.gain = gain;
  if (mc) { a.mc_demod = mc->demod; a.mc_noise = mc->noise; a.mc_noise_w = mc->noise_w; a.mc_bias = mc->bias; a.mc_post = mc->post; }
  CONTRAD_ARG(a.out_h > 0 && a.out_w > 0);
  // non-temporal stores once the output cannot stay in the 256 MB Infinity Cache for its consumer (the same rule as the conv
  // engine's epilogues, igemm.hip NT_STORE_BYTES): StyleGAN2_512, five same-box alternations: 62.79 -> 62.04 ms per step
  a.nt_store = ((long long)major * a.out_h * a.out_w * minor * (long long)sizeof(float) >= (256ll << 20)) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (minor & 3) == 0;
  const bool fir4 = vec && kh == 4 && kw == 4 && up_x == up_y && down_x == down_y;
  // the branch-free forms: one image per grid.y, 32-bit byte offsets inside an image
  const bool buf_ok = major <= 65535 && (long long)in_h * in_w * minor * 4 < (1ll << 31) && (long long)a.out_h * a.out_w * minor * 4 < (1ll << 31);
  if (fir4 && up_x == 1 && down_x == 1) {
    if (buf_ok) {
      const long long per = (long long)((a.out_h + 3) / 4) * ((a.out_w + UF_TX - 1) / UF_TX) * (minor / 4);
      hipLaunchKernelGGL(upfirdn4_u1d1_buf_kernel<UF_TX>, dim3((unsigned)cdivll(per, 256), (unsigned)major), dim3(256), 0, s, a);
      CONTRAD_CHECK_LAUNCH();
      return 0;
    }
    const long long tot = (long long)major * ((a.out_h + 3) / 4) * ((a.out_w + 1) / 2) * (minor / 4);
    hipLaunchKernelGGL(upfirdn4_u1d1_kernel, dim3((unsigned)cdivll(tot, 256)), dim3(256), 0, s, a);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  CONTRAD_ARG(!mc);          // the modulated-conv epilogue exists on the 4x4-FIR blur path only
  if (fir4 && ((up_x == 1 && down_x == 2) || (up_x == 2 && down_x == 1))) {
    const long long per = (long long)((a.out_h + 1) / 2) * ((a.out_w + 1) / 2) * (minor / 4);
    const long long tot = (long long)major * per;
    if (buf_ok) {
      const dim3 g((unsigned)cdivll(per, 256), (unsigned)major);
      if (down_x == 2) hipLaunchKernelGGL(upfirdn4_u1d2_buf_kernel, g, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(upfirdn4_u2d1_buf_kernel, g, dim3(256), 0, s, a);
    } else if (down_x == 2) hipLaunchKernelGGL(upfirdn4_u1d2_kernel, dim3((unsigned)cdivll(tot, 256)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(upfirdn4_u2d1_kernel, dim3((unsigned)cdivll(tot, 256)), dim3(256), 0, s, a);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  if (minor == 1 && kh == 4 && kw == 4 && up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1) {
    const long long tot = (long long)major * ((a.out_h + 1) / 2) * ((a.out_w + 1) / 2);
    hipLaunchKernelGGL(upfirdn4_u2d1_planes_kernel, dim3((unsigned)cdivll(tot, 256)), dim3(256), 0, s, a);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  if (vec && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && a.out_h >= 4) {
    const long long tot = (long long)major * ((a.out_h + 3) / 4) * a.out_w * (minor / 4);
    long long g = (tot + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(upfirdn2d_strip_kernel<4>, dim3((int)g), dim3(256), 0, s, a);
    CONTRAD_CHECK_LAUNCH();
    return 0;
  }
  const long long total = (long long)major * a.out_h * a.out_w * (vec ? minor / 4 : minor);
  long long grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  if (vec) hipLaunchKernelGGL(upfirdn2d_kernel<4>, dim3((int)grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(upfirdn2d_kernel<1>, dim3((int)grid), dim3(256), 0, s, a);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_upfirdn2d(const float* input, const float* kernel, float* out, int major, int in_h,
                                 int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                 int pad_x0, int pad_x1, int pad_y0, int pad_y1, contrad_stream_t stream) {
  return upfirdn2d_launch(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0,
                          pad_x1, pad_y0, pad_y1, nullptr, nullptr, 1.f, 1.f, nullptr, stream);
}

extern "C" int contrad_upfirdn2d_fused(const float* input, const float* kernel, float* out, int major, int in_h,
                                       int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x,
                                       int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* addend,
                                       const float* act_ref, float slope, float gain, float* out2,
                                       contrad_stream_t stream) {
  return upfirdn2d_launch(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0,
                          pad_x1, pad_y0, pad_y1, addend, act_ref, slope, gain, out2, stream);
}

extern "C" int contrad_upfirdn2d_modconv(const float* input, const float* kernel, float* y, int N, int in_h, int in_w,
                                         int K, int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* demod,
                                         const float* noise, const float* noise_w, const float* bias,
                                         const float* post_scale, contrad_stream_t stream) {
  CONTRAD_ARG(y && bias && K > 0 && (K & 3) == 0);
  const ModconvEpi mc{demod, noise, noise_w, bias, post_scale};
  return upfirdn2d_launch(input, kernel, y, N, in_h, in_w, K, 4, 4, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, nullptr,
                          nullptr, 1.f, 1.f, nullptr, stream, &mc);
}

extern "C" int contrad_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, long long n,
                                      int step_b, int size_b, int act, int grad, float alpha, float scale,
                                      contrad_stream_t stream) {
  CONTRAD_ARG(x && y && n > 0 && (act == 1 || act == 3) && grad >= 0 && grad <= 2);
  CONTRAD_ARG(!bias || (step_b > 0 && size_b > 0));
  CONTRAD_ARG(grad != 1 || act == 1 || ref);
  long long grid = (n + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, bias, ref, y, n,
                     step_b, size_b, act, grad, alpha, scale);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_lincomb(const float* x, const float* z, float* y, long long n, float a, float b,
                               contrad_stream_t stream) {
  CONTRAD_ARG(x && z && y && n > 0);
  long long grid = (n / 4 + 255) / 256 + 1;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(lincomb_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, z, y, n, a, b);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

// ----------------------------------------------------------------------------------------------------------------------
// Minibatch-stddev channel (_minibatch_stddev_layer, models/gan/stylegan2/discriminator.py:22-33) with its first and second
// backward, on NHWC.  x [B][P][C] (P = H*W pixels, dense), y [B][P][Cp] with Cp >= C + 1: y[.., :C] = x, y[.., C] =
// s[b mod M], y[.., C+1:] = 0 (the channel count padded to a multiple of 16 for the conv engine).  group G = min(B, 4),
// M = B / G;  s[m] = mean_{p,c} sqrt( var_g( x[g*M + m, p, c] ) + 1e-8 ), var biased (as the reference: unbiased=False).
// One block per m (1024 threads): G * P * C = 32 k ... 131 k values per block, read once; sums in a fixed order.
// The R1 penalty differentiates D's input gradient, so the backward is itself differentiable (autograd_ops.MinibatchStddevFn):
//   first backward   gx[g,p,c]  = gy[g,p,c] + gs * (x_g - mu) / (G * PC * sigma),  gs = sum_{g,p} gy[g,p,C]  (per m), PC = P*C
//   second backward  (cotangent h of gx)
//                    ggy[g,p,c] = h[g,p,c];  ggy[g,p,C] = t = sum_{g,p,c} h_g (x_g - mu) / (G * PC * sigma);  pad channels 0
//                    gx2[k,p,c] = gs / (G * PC) * ( (h_k - mean_g h) / sigma - (x_k - mu) * sum_g h_g (x_g - mu) / (G sigma^3) )
// ----------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int MBS_THREADS = 1024;
constexpr int MBS_MAXG = 4;

__device__ __forceinline__ float mbs_block_sum(float v, float* red) {   // fixed-order block reduction (1024 threads)
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MBS_THREADS / 64; ++i) s += red[i];
  return s;
}

// mode 0: forward; 1: first backward; 2: second backward
template <int MODE>
__global__ __launch_bounds__(MBS_THREADS) void mbstd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                            const float* __restrict__ h, float* __restrict__ out,
                                                            float* __restrict__ out2, int G, int M, int P, int C, int Cp) {
  __shared__ float red[MBS_THREADS / 64];
  const int m = blockIdx.x;
  const long long PC = (long long)P * C;
  const float invG = 1.f / (float)G;
  float gs = 0.f;
  if (MODE >= 1) {                       // gs = sum over the group's samples and pixels of the stddev channel of gy
    float v = 0.f;
    for (int i = threadIdx.x; i < G * P; i += MBS_THREADS) {
      const int g = i / P, p = i - g * P;
      v += gy[((long long)(g * M + m) * P + p) * Cp + C];
    }
    gs = mbs_block_sum(v, red);
  }
  const float k1 = gs / ((float)G * (float)PC);
  float acc = 0.f;
  for (long long e = threadIdx.x; e < PC; e += MBS_THREADS) {
    const int p = (int)(e / C), c = (int)(e - (long long)p * C);
    float xv[MBS_MAXG], hv[MBS_MAXG];
    float mu = 0.f;
#pragma unroll
    for (int g = 0; g < MBS_MAXG; ++g)
      if (g < G) { xv[g] = x[(long long)(g * M + m) * PC + e]; mu += xv[g]; }
    mu *= invG;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < MBS_MAXG; ++g)
      if (g < G) { const float dd = xv[g] - mu; var += dd * dd; }
    const float sigma = sqrtf(var * invG + 1e-8f);
    if (MODE == 0) {
      acc += sigma;
#pragma unroll
      for (int g = 0; g < MBS_MAXG; ++g)
        if (g < G) out[((long long)(g * M + m) * P + p) * Cp + c] = xv[g];
    } else if (MODE == 1) {
      const float r = k1 / sigma;
#pragma unroll
      for (int g = 0; g < MBS_MAXG; ++g)
        if (g < G) out[(long long)(g * M + m) * PC + e] = gy[((long long)(g * M + m) * P + p) * Cp + c] + r * (xv[g] - mu);
    } else {
      float hbar = 0.f, hx = 0.f;
#pragma unroll
      for (int g = 0; g < MBS_MAXG; ++g)
        if (g < G) { hv[g] = h[(long long)(g * M + m) * PC + e]; hbar += hv[g]; hx += hv[g] * (xv[g] - mu); }
      hbar *= invG;
      const float is = 1.f / sigma;
      acc += hx * is;
      const float q = hx * invG * is * is * is;
#pragma unroll
      for (int g = 0; g < MBS_MAXG; ++g)
        if (g < G) {
          out[(long long)(g * M + m) * PC + e] = k1 * ((hv[g] - hbar) * is - (xv[g] - mu) * q);     // gx2
          out2[((long long)(g * M + m) * P + p) * Cp + c] = hv[g];                                   // ggy[.., :C] = h
        }
    }
  }
  if (MODE == 1) return;
  const float tot = mbs_block_sum(acc, red);
  const float chan = (MODE == 0) ? tot / (float)PC : tot / ((float)G * (float)PC);
  float* y = (MODE == 0) ? out : out2;
  const int extra = Cp - C;              // stddev channel + zero padding
  for (int i = threadIdx.x; i < G * P * extra; i += MBS_THREADS) {
    const int j = i % extra, gp = i / extra;
    const int g = gp / P, p = gp - g * P;
    y[((long long)(g * M + m) * P + p) * Cp + C + j] = (j == 0) ? chan : 0.f;
  }
}

// partial[b] = sum of x^2 over a fixed slice; r1_final: out = scale * sum_b partial[b]  (fixed order)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float v = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) v += x[i] * x[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float v = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) v += partial[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}
// y = x * (c * s[0])   (s: a scalar in device memory -- the incoming gradient of a scalar loss term)
__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float c,
                                                        float* __restrict__ y, long long n) {
  const float f = c * s[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * f;
}

}  // namespace

extern "C" int contrad_minibatch_stddev(int mode, const float* x, const float* gy, const float* h, float* out, float* out2,
                                        int B, int P, int C, int Cp, contrad_stream_t stream) {
  CONTRAD_ARG(x && out && B > 0 && P > 0 && C > 0 && Cp >= C + 1 && mode >= 0 && mode <= 2);
  CONTRAD_ARG(mode == 0 || gy);
  CONTRAD_ARG(mode != 2 || (h && out2));
  const int G = B < MBS_MAXG ? B : MBS_MAXG;
  CONTRAD_ARG(B % G == 0);                 // (the reference's reshape(group, -1, ...) needs it too)
  const int M = B / G;
  if (mode == 0)
    hipLaunchKernelGGL(mbstd_kernel<0>, dim3(M), dim3(MBS_THREADS), 0, (hipStream_t)stream, x, gy, h, out, out2, G, M, P, C, Cp);
  else if (mode == 1)
    hipLaunchKernelGGL(mbstd_kernel<1>, dim3(M), dim3(MBS_THREADS), 0, (hipStream_t)stream, x, gy, h, out, out2, G, M, P, C, Cp);
  else
    hipLaunchKernelGGL(mbstd_kernel<2>, dim3(M), dim3(MBS_THREADS), 0, (hipStream_t)stream, x, gy, h, out, out2, G, M, P, C, Cp);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" long long contrad_sumsq_workspace_bytes(long long n) {
  if (n <= 0) return -22;
  long long nb = (n + 256 * 16 - 1) / (256 * 16);
  if (nb > 1024) nb = 1024;
  return nb * (long long)sizeof(float);
}

extern "C" int contrad_sumsq(const float* x, long long n, float scale, float* out, float* workspace,
                             long long workspace_bytes, contrad_stream_t stream) {
  CONTRAD_ARG(x && out && workspace && n > 0 && workspace_bytes >= contrad_sumsq_workspace_bytes(n));
  const int nb = (int)(contrad_sumsq_workspace_bytes(n) / (long long)sizeof(float));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, workspace);
  CONTRAD_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nb, scale, out);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}

extern "C" int contrad_scale_dev(const float* x, const float* s, float c, float* y, long long n, contrad_stream_t stream) {
  CONTRAD_ARG(x && s && y && n > 0);
  long long grid = (n + 256 * 8 - 1) / (256 * 8);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_dev_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, s, c, y, n);
  CONTRAD_CHECK_LAUNCH();
  return 0;
}
