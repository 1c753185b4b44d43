/* libcontrad_hip.so -- C ABI of the MI355X-native (gfx950) ContraD discriminator-step hot path.
 *
 * Drop-in boundary B2 of SURVEY.md section 8(b): the reference's only native interface is two
 * pybind11 torch extensions JIT-built at import (models/gan/stylegan2/op/upfirdn2d.cpp:12-23,
 * op/fused_bias_act.cpp:11-21); everything else on the path reaches cuDNN/ATen through PyTorch
 * (F.conv2d, F.linear, grid_sample, spectral_norm, log_softmax, optim.Adam).  This library replaces
 * all of them with hand-written HIP kernels behind one convention:
 *
 *   - plain C: raw DEVICE pointers (fp32 unless noted) + sizes; no torch types, no hidden allocation,
 *     no host synchronisation, re-entrant; every launcher enqueues on the caller's `stream`
 *     (a hipStream_t passed as void*);
 *   - outputs and workspaces are caller-allocated;
 *   - return value 0 = ok, >0 = hipError_t of the launch, <0 = -EINVAL style argument error
 *     (the Python host raises RuntimeError, mirroring TORCH_CHECK in the reference's .cpp shims).
 *
 * Internal activation layout is NHWC ("pixel-major": a pixel's channels are contiguous, `ld*` floats
 * between pixels); the Python module boundary stays NCHW like the reference.  Weights are consumed in
 * a packed GEMM layout Wp[(kh*KW+kw)*Cin + c][Cout] produced by contrad_weight_prep_* below.
 */
#ifndef CONTRAD_HIP_H
#define CONTRAD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* contrad_stream_t; /* hipStream_t */

int contrad_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution / linear engine: implicit-GEMM on v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate.
 * Replaces F.conv2d fwd + dgrad + wgrad (models/gan/sndcgan.py:91-109, stylegan2/layers.py:115-121),
 * nn.ConvTranspose2d forward (= dgrad; sndcgan.py:26-38) and F.linear (conv with H=W=KH=KW=1;
 * models/gan/base.py:14-35,92-101).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int N, H, W, C;   /* input  tensor: batch, height, width, channels (Cin)            */
  int ldx;          /* floats between consecutive input pixels  (>= C, multiple of 4)  */
  int Ho, Wo, K;    /* output tensor: height, width, channels (Cout)                   */
  int ldy;          /* floats between consecutive output pixels (>= K)                 */
  int KH, KW;       /* filter size                                                     */
  int stride, pad;  /* same in both dims; stride in {1,2}                              */
  int ldw;          /* floats between rows of the packed weight [KH*KW*C][ldw] (>= K)  */
} contrad_conv_desc;

/* y[n,ho,wo,k] = gain * lrelu_slope( sum_{kh,kw,c} x[n,ho*s-p+kh,wo*s-p+kw,c] * wp[(kh,kw,c),k] + bias[k] )
 * bias may be NULL; slope = 1 disables the activation. */
int contrad_conv2d_fwd(const contrad_conv_desc* d, const float* x, const float* wp, const float* bias,
                       float* y, float slope, float gain, contrad_stream_t stream);

/* dx[n,h,w,c] = act'(act_ref[n,h,w,c]) * sum_{kh,kw,k} gy[n,(h+p-kh)/s,(w+p-kw)/s,k] * wp[(kh,kw,c),k]
 * act_ref (may be NULL) is the OUTPUT of the leaky-relu that produced this conv's input (same layout
 * as dx, leading dimension ldx): act' = gain * (act_ref > 0 ? 1 : slope).
 * Stride-2 is decomposed into s*s output-parity classes so only contributing taps are multiplied. */
int contrad_conv2d_dgrad(const contrad_conv_desc* d, const float* gy, const float* wp, float* dx,
                         const float* act_ref, float slope, float gain, contrad_stream_t stream);

/* dwp[(kh,kw,c),k] = sum_{n,ho,wo} x[n,ho*s-p+kh,wo*s-p+kw,c] * gy[n,ho,wo,k]      (split over the
 * n*ho*wo axis into deterministic partial slabs in `workspace`, then reduced in fixed order). */
long long contrad_conv2d_wgrad_workspace_bytes(const contrad_conv_desc* d);
int contrad_conv2d_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp,
                         float* workspace, long long workspace_bytes, contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Contrastive losses: fused pairwise-cosine + log-sum-exp, S = Z Z^T / temp never written to HBM.
 * Replaces nt_xent (training/criterion.py:24-45; mode 0, R = 2N) and supcon_fake
 * (training/gan/contrad.py:8-32; mode 1, R = 3N, anchors = rows 2N..3N) incl. their autograd backward,
 * and F.normalize (contrad.py:43,48).
 * ---------------------------------------------------------------------------------------------- */
/* z[R,D] row-normalised. Writes lse[R], rowloss[R] (scratch) and loss[0] = mean anchor loss. */
int contrad_contrast_fwd(const float* z, int R, int D, int N, int mode, float inv_temp, float* lse,
                         float* rowloss, float* loss, contrad_stream_t stream);
/* dz[R,D] = (grad_scale ? grad_scale[0] : 1) * d loss / d z   (lse from contrad_contrast_fwd). */
int contrad_contrast_bwd(const float* z, const float* lse, int R, int D, int N, int mode, float inv_temp,
                         const float* grad_scale, float* dz, contrad_stream_t stream);
/* z = u / max(||u||_2, eps) per row; inv_norm[R] kept for the backward. */
int contrad_l2norm_fwd(const float* u, int ldu, float* z, float* inv_norm, int R, int D, float eps,
                       contrad_stream_t stream);
/* du (+)= (dz - z <z,dz>) * inv_norm */
int contrad_l2norm_bwd(const float* dz, const float* z, const float* inv_norm, float* du, int ldu, int R,
                       int D, int accumulate, contrad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTRAD_HIP_H */
