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

/* Alignment: when the channel counts / leading dimensions are multiples of 4 the kernels use 16-byte accesses, and
 * x, wp, y, gy, dx, act_ref and the workspaces must then be 16-byte aligned (-EINVAL otherwise).
 *
 * y[n,ho,wo,k] = gain * lrelu_slope( sum_{kh,kw,c} x[n,ho*s-p+kh,wo*s-p+kw,c] * wp[(kh,kw,c),k] + bias[k] )
 * bias may be NULL; slope = 1 disables the activation. */
int contrad_conv2d_fwd(const contrad_conv_desc* d, const float* x, const float* wp, const float* bias,
                       float* y, float slope, float gain, float* workspace, long long workspace_bytes,
                       contrad_stream_t stream);
/* Same with a residual addend: y = gain * lrelu_slope(conv + bias) + addend, `addend` (may be NULL) a tensor of y's
 * shape and leading dimension (may alias nothing).  The ResBlock merge (out + skip) of the StyleGAN2 discriminator
 * (models/gan/stylegan2/discriminator.py:72-74) is the epilogue of its 1x1 skip convolution this way. */
int contrad_conv2d_fwd_add(const contrad_conv_desc* d, const float* x, const float* wp, const float* bias,
                           const float* addend, float* y, float slope, float gain, float* workspace,
                           long long workspace_bytes, contrad_stream_t stream);
/* Scratch for the split-K partial slabs of small-M / deep-K shapes (the merged head GEMM); 0 for most shapes, in
 * which case workspace may be NULL. */
long long contrad_conv2d_fwd_workspace_bytes(const contrad_conv_desc* d);

/* dx[n,h,w,c] = act'(act_ref[n,h,w,c]) * sum_{kh,kw,k} gy[n,(h+p-kh)/s,(w+p-kw)/s,k] * wp[(kh,kw,c),k]
 * act_ref (may be NULL) is the OUTPUT of the leaky-relu that produced this conv's input (same layout
 * as dx, leading dimension ldx): act' = gain * (act_ref > 0 ? 1 : slope).
 * Stride-2 is decomposed into s*s output-parity classes so only contributing taps are multiplied. */
int contrad_conv2d_dgrad(const contrad_conv_desc* d, const float* gy, const float* wp, float* dx,
                         const float* act_ref, float slope, float gain, contrad_stream_t stream);
/* Same, with a scratch buffer: small-M / deep-K stride-1 layers (the 4x4 and 8x8 levels at small per-rank batches) then
 * split the contraction into deterministic partial slabs (dx's layout) that a second kernel sums in fixed order before
 * the fused act' epilogue.  workspace may be NULL (never splits); size from contrad_conv2d_dgrad_workspace_bytes. */
long long contrad_conv2d_dgrad_workspace_bytes(const contrad_conv_desc* d);
int contrad_conv2d_dgrad_ws(const contrad_conv_desc* d, const float* gy, const float* wp, float* dx,
                            const float* act_ref, float slope, float gain, float* workspace,
                            long long workspace_bytes, contrad_stream_t stream);

/* Winograd F(2x2, 3x3) in exact fp32 (csrc/wino.h) for 3x3 stride-1 pad-1 layers on power-of-two maps >= 4x4 whose input
 * channels are a multiple of 16 and output channels a multiple of 64: 2.25x fewer multiply-adds than the direct kernels,
 * fp32 round-off class results (rel-L2 3 - 6e-7 against fp64; F.conv2d of the reference reaches cuDNN's Winograd-class
 * kernels the same way: models/gan/sndcgan.py:91-109, models/gan/stylegan2/layers.py:115-121).  contrad_conv2d_fwd_add /
 * contrad_conv2d_dgrad_ws choose it by themselves for launches of about one item (64 tiles x 64 output channels) per CU or
 * more -- their *_workspace_bytes then cover the transformed filter (16 * C * K floats, rewritten by every call) --;
 * contrad_conv2d_wino runs it on ANY shape contrad_conv2d_wino_ok accepts (parity tests, integrators with their own plan).
 * mode 0: in = x, out = y = gain * lrelu(conv + bias) [+ ref], bias / ref may be NULL;
 * mode 1: in = gy, out = dx [* act'(ref)], bias must be NULL (semantics of contrad_conv2d_fwd_add / _dgrad_ws).
 * The weight gradient of the same layers (mode 2: F(3x3, 2x2), input AND output channels multiples of 64) runs on
 * wino_wgrad_kernel: contrad_conv2d_wgrad picks it when every CU gets a long enough share of the tiles,
 * contrad_conv2d_wino_wgrad forces it (arguments and results of contrad_conv2d_wgrad, workspace = split slabs).
 * The same entry points serve the strided layers: 4x4 stride 2 pad 1 (csrc/wino22.h, all three modes) and, mode 0 only, 3x3
 * stride 2 pad 0 on a (2G + 1) x (2G + 1) map (csrc/wino23.h: StyleGAN2's blurred down-sampling convolution, models/gan/
 * stylegan2/layers.py:174-198; F(2x2, 2x2) on the four input phases with the structurally zero planes skipped: 25 / 36 of the
 * dense layer's multiply-adds; contrad_conv2d_path = 10). */
int contrad_conv2d_wino_ok(const contrad_conv_desc* d, int mode);
long long contrad_conv2d_wino_workspace_bytes(const contrad_conv_desc* d, int mode);
int contrad_conv2d_wino_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp, float* dbias,
                              float* workspace, long long workspace_bytes, contrad_stream_t stream);
int contrad_conv2d_wino(const contrad_conv_desc* d, int mode, const float* in, const float* wp, const float* bias,
                        const float* ref, float* out, float slope, float gain, float* workspace,
                        long long workspace_bytes, contrad_stream_t stream);

/* Winograd F(4x4, 3x3) in fp32 (csrc/wino44.h) for the same 3x3 stride-1 pad-1 layers on power-of-two maps >= 16x16 whose
 * input channels and output channels are multiples of 32 (output channels in whole 64-wide blocks: wino44_kernel; an odd number
 * of 32-wide blocks -- StyleGAN2_512's 32 -> 32 channel layers -- csrc/wino44n.h): 2.25 multiply-adds per output instead of 4
 * (F(2x2, 3x3)) or 9 (direct); standard interpolation points (0, +-1, +-2, inf), round-off rel-L2 1 - 5e-6 against fp64 (the
 * contract of this path is 1e-3; reference as above: F.conv2d of models/gan/sndcgan.py:91-109, stylegan2/layers.py:115-121).
 * contrad_conv2d_fwd_add / contrad_conv2d_dgrad_ws choose it ahead of F(2x2, 3x3) for launches of a full round of its items
 * (512 output pixels x 64 or 32 output channels per CU) -- their *_workspace_bytes then cover 36 * C * K floats --;
 * contrad_conv2d_wino44 forces it on any shape contrad_conv2d_wino44_ok accepts (arguments and semantics of
 * contrad_conv2d_wino, modes 0 and 1). */
int contrad_conv2d_wino44_ok(const contrad_conv_desc* d, int mode);
long long contrad_conv2d_wino44_workspace_bytes(const contrad_conv_desc* d);
int contrad_conv2d_wino44(const contrad_conv_desc* d, int mode, const float* in, const float* wp, const float* bias,
                          const float* ref, float* out, float slope, float gain, float* workspace,
                          long long workspace_bytes, contrad_stream_t stream);

/* dwp[(kh,kw,c),k] = sum_{n,ho,wo} x[n,ho*s-p+kh,wo*s-p+kw,c] * gy[n,ho,wo,k]      (split over the
 * n*ho*wo axis into deterministic partial slabs in `workspace`, then reduced in fixed order).
 * dbias (may be NULL; needs C, K, ldx, ldy multiples of 4): dbias[k] = sum_{n,ho,wo} gy[n,ho,wo,k], the bias
 * gradient, accumulated for free from the gy tiles the kernel streams anyway. */
long long contrad_conv2d_wgrad_workspace_bytes(const contrad_conv_desc* d);
/* Introspection for profiling: block tile (bm x bn x 32) the launcher picks for `mode` (0 fwd, 1 dgrad,
 * 2 wgrad) on this geometry, i.e. which igemm_kernel<mode, bm, bn> instance runs. */
int contrad_conv2d_tile(const contrad_conv_desc* d, int mode, int* bm, int* bn);
/* Which kernel family serves this shape: 0 = general scalar-gather (igemm_kernel<..., false>), 1 = general float4
 * (igemm_kernel<..., true>), 2 = lean loop (igemm_lean_kernel), 3 = lean loop on tiles that skip padding taps (small maps:
 * pixel-major tiles -- the rows of a tile are images at one output pixel -- or border classes -- (image, pixel) rows inside
 * a rectangle of pixels that share their non-padding taps), 4 = the accumulator-stationary weight-gradient kernel of
 * the 32 -> 32 channel 3x3 layers (wgrad_c32_kernel, mode 2 only), 5 = the single-output 1x1 layer (fwd_k1_kernel, mode 0
 * only: the 512 -> 1 logit of the heads), 6 = the weight-stationary kernel of the 32 -> 32 channel 3x3 stride-1 layers
 * (conv_c32_kernel<mode>, modes 0 and 1), 7 = Winograd F(2x2, 3x3) (wino_kernel<mode>, modes 0 and 1; with a workspace) / F(3x3, 2x2) (wino_wgrad_kernel, mode 2),
 * 8 = F(2x2, 2x2) on the phases of the 4x4 stride-2 layers (wino22_kernel<mode> / wino22_wgrad_kernel), 9 = Winograd F(4x4, 3x3)
 * (wino44_kernel<mode>, modes 0 and 1; with a workspace), 10 = F(2x2, 2x2) on the phases of a 3x3 stride-2 pad-0 layer with the zero
 * planes skipped (wino23_kernel, mode 0; with a workspace), 11 = F(4x4, 3x3) with 32-wide output-channel blocks (wino44n_kernel<mode>:
 * output channels not a multiple of 64, and the 4x4 maps; with a workspace);  negative = bad descriptor.  Profiling aid. */
int contrad_conv2d_path(const contrad_conv_desc* d, int mode);
/* Share of the layer's nominal multiply-adds (2*N*Ho*Wo*K*C*KH*KW, the count every roofline here is quoted on, padding
 * taps included as in the reference's dense layer) that the kernel actually issues: 1 except on pixel-major tiles (path
 * 3), which skip the tap-positions that read padding (0.69 for a 3x3 pad-1 layer on a 4x4 map), and on the Winograd path
 * (7): 4/9, (8): 9/16, (9) and (11): 1/4, (10): 25/36 -- the transform-domain multiply-adds.  (A weight-gradient tile
 * that also sums the bias gradient visits everything: not reflected.)  Profiling aid. */
double contrad_conv2d_executed_fraction(const contrad_conv_desc* d, int mode);
/* Workgroups (256 threads each) of the main igemm launch this geometry gets for `mode` (with_workspace != 0: the plan the
 * *_ws entry points use, i.e. split-K allowed).  Profiling aid: lets a rocprofv3 kernel trace, which names only the
 * template instance, be joined to the layer shape a dispatch served (tools/rocpd_rows.py); negative = bad descriptor. */
long long contrad_conv2d_grid_blocks(const contrad_conv_desc* d, int mode, int with_workspace);
/* Pixel-major launches (path 3, mode 0 / 1) with at most 256 M-tiles per table: the order in which the launch walks its
 * M-tiles, out[i] = image block * pixels + pixel of the i-th M-tile (for the strided data gradient: of every parity class,
 * the pixel mirrored along the class's odd axes), chosen so that the tiles a CU is dealt -- an XCD's blocks go round-robin
 * over its 32 CUs -- carry equal work.  Returns the number of entries written, 0 when the launch uses no such table
 * (image-major tiles, border classes, more than 256 tiles), negative = bad argument.  Host logic only; lets the plan be
 * tested without a GPU (tests/test_abi_cpu.py). */
int contrad_conv2d_tile_order(const contrad_conv_desc* d, int mode, unsigned char* out, int capacity);
int contrad_conv2d_wgrad(const contrad_conv_desc* d, const float* x, const float* gy, float* dwp,
                         float* dbias, float* workspace, long long workspace_bytes, contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Contrastive losses: fused pairwise-cosine + log-sum-exp, S = Z Z^T / temp never written to HBM.
 * Replaces nt_xent (training/criterion.py:24-45; mode 0, R = 2N) and supcon_fake
 * (training/gan/contrad.py:8-32; mode 1, R = 3N, anchors = rows 2N..3N) incl. their autograd backward,
 * and F.normalize (contrad.py:43,48).
 * ---------------------------------------------------------------------------------------------- */
/* Scratch for the column-split partials of either call (the R x R work is spread over ~256 blocks; the
 * per-split (max, sum, target) triples / dZ slabs are merged in a fixed order -> deterministic). */
long long contrad_contrast_workspace_bytes(int R, int D);
/* z[R,D] row-normalised. Writes lse[R], rowloss[R] (scratch) and loss[0] = mean anchor loss. */
int contrad_contrast_fwd(const float* z, int R, int D, int N, int mode, float inv_temp, float* lse,
                         float* rowloss, float* loss, float* workspace, long long workspace_bytes,
                         contrad_stream_t stream);
/* dz[R,D] = (grad_scale ? grad_scale[0] : 1) * d loss / d z   (lse from contrad_contrast_fwd). */
int contrad_contrast_bwd(const float* z, const float* lse, int R, int D, int N, int mode, float inv_temp,
                         const float* grad_scale, float* dz, float* workspace, long long workspace_bytes,
                         contrad_stream_t stream);
/* z = u / max(||u||_2, eps) per row; inv_norm[R] kept for the backward. */
int contrad_l2norm_fwd(const float* u, int ldu, float* z, float* inv_norm, int R, int D, float eps,
                       contrad_stream_t stream);
/* du (+)= (dz - z <z,dz>) * inv_norm */
int contrad_l2norm_bwd(const float* dz, const float* z, const float* inv_norm, float* du, int ldu, int R,
                       int D, int accumulate, contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight preparation, batched over the layers of a network: spectral normalisation (one power
 * iteration per forward in training mode, u/v updated in place -- torch.nn.utils.spectral_norm as
 * applied at models/gan/sndcgan.py:111-118) or a fixed runtime scale (EqualConv2d/EqualLinear,
 * models/gan/stylegan2/layers.py:104,117,141), fused with the OIHW -> packed-GEMM-layout transpose.
 * ---------------------------------------------------------------------------------------------- */
#define CONTRAD_SN_MAX_LAYERS 24
#define CONTRAD_SN_MAX_PARTIALS 256
typedef struct {
  const float* w;    /* [K][C*T] original weight (OIHW flattened; column = c*T + tap)            */
  float* u;          /* [K]   left singular vector estimate (in/out)      (NULL if fixed_scale)  */
  float* v;          /* [C*T] right singular vector estimate (in/out)     (NULL if fixed_scale)  */
  float* u_snap;     /* [K]   copy of the u used by THIS call (fwd: written; bwd: read); may be NULL  */
  float* v_snap;     /* [C*T] copy of the v used by THIS call (fwd: written; bwd: read); may be NULL  */
  float* wp;         /* [T*C][ldw] packed effective weight (out fwd, in bwd)                      */
  const float* gwp;  /* [T*C][ldw] gradient wrt wp            (backward only)                     */
  float* gw;         /* [K][C*T]   gradient wrt w (out)       (backward only)                     */
  int K, C, T, ldw;  /* T = KH*KW                                                                 */
  float fixed_scale; /* > 0: wp = w * fixed_scale, no spectral norm                               */
} contrad_sn_layer;
typedef struct {
  int n;
  contrad_sn_layer layers[CONTRAD_SN_MAX_LAYERS];
  long long scratch_off[CONTRAD_SN_MAX_LAYERS]; /* float offset of layer l's scratch, each at least
                                                   contrad_sn_scratch_floats(K, C, T) long            */
} contrad_sn_batch;

long long contrad_sn_scratch_floats(int K, int C, int T);
/* sigma_out[n]: per-layer sigma (1/fixed_scale for fixed layers), needed by the backward. */
int contrad_sn_weight_prep(const contrad_sn_batch* b, int training, float eps, float* scratch,
                           float* sigma_out, contrad_stream_t stream);
/* gw = (gwp - <gwp, wp> u v^T) / sigma, with the sigma, wp and u/v (u_snap/v_snap when given, else u/v)
 * of the matching forward. */
int contrad_sn_weight_grad(const contrad_sn_batch* b, float* scratch, const float* sigma,
                           contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RGB ends of the networks (one GEMM dimension is 3 -> direct VALU kernels, HBM-bound).
 * ---------------------------------------------------------------------------------------------- */
/* First conv of D on the NCHW image with the x*2-1 input rescale fused (D_SNDCGAN.penultimate +
 * main.0, models/gan/sndcgan.py:91,123; StyleGAN2 FromRGB, stylegan2/discriminator.py:17-19,226):
 * y[n,h,w,:] = gain*lrelu_slope(conv_kxk(img*in_scale+in_shift) + bias), NHWC out; k in {1,3}, Cin = 3. */
int contrad_rgb_conv_fwd(const float* img, const float* wp, const float* bias, float* y, int N, int Cin,
                         int H, int W, int K, int k, int ldy, int ldw, float in_scale, float in_shift,
                         float slope, float gain, contrad_stream_t stream);
long long contrad_rgb_conv_wgrad_workspace_bytes(int N, int Cin, int H, int W, int K, int k);
/* dwp[(tap*Cin+ci)][k] and dbias[k] (may be NULL) from gy = gradient wrt the pre-activation output. */
int contrad_rgb_conv_wgrad(const float* img, const float* gy, float* dwp, float* dbias, int N, int Cin,
                           int H, int W, int K, int k, int ldy, int ldw, float in_scale, float in_shift,
                           float* workspace, long long workspace_bytes, contrad_stream_t stream);
/* Stride-1 transposed conv onto C <= 4 channels, NHWC in -> NCHW out,
 * out = f(acc + bias + residual) * out_scale + out_shift, f = identity (act 0) or tanh (act 1); `mod` (may be
 * NULL) is a per-sample [N][K] modulation of the input channels, `residual` (may be NULL) an NCHW tensor:
 * G_SNDCGAN's last ConvTranspose2d+Tanh+0.5x+0.5 (models/gan/sndcgan.py:37-38,47), d loss / d image of D's first
 * conv, and StyleGAN2's ToRGB = 1x1 modulated conv + bias + upsampled skip (stylegan2/generator.py:122-143). */
int contrad_rgb_conv_dgrad(const float* gy, const float* wp, const float* bias, const float* mod,
                           const float* residual, float* out, int N, int C, int H, int W, int K, int k,
                           int ldy, int ldw, int act, float out_scale, float out_shift,
                           contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * HBM-bound helpers: column statistics, BatchNorm (generator forward), GAN logit losses, Adam.
 * ---------------------------------------------------------------------------------------------- */
/* out[0][c] = sum_r x[r][c] (and out[1][c] = sum_r x[r][c]^2 if with_sq) over M rows; bias gradients
 * (autograd of the "+ bias" in nn.Conv2d / nn.Linear) and BatchNorm batch statistics. */
long long contrad_colstats_workspace_bytes(long long M, int K, int with_sq);
int contrad_colstats(const float* x, long long M, int K, int ld, int with_sq, float* out, int accumulate,
                     float* workspace, long long workspace_bytes, contrad_stream_t stream);
/* nn.BatchNorm2d (train mode) + ReLU of G_SNDCGAN (models/gan/sndcgan.py:25-35,43): stats = {sum, sumsq}
 * over `count` rows (all ranks for SyncBatchNorm); perm_hw > 1 additionally maps column c*perm_hw+hw to
 * NHWC (hw, c) (the view(-1, 512, 4, 4) after norm_init, sndcgan.py:44). */
int contrad_bn_relu_apply(const float* x, float* y, long long M, int K, int ldx, int ldy, const float* stats,
                          float count, const float* gamma, const float* beta, float eps, int perm_hw,
                          contrad_stream_t stream);
/* Backward of BatchNorm(train)+ReLU: out2k = {S1[c] = sum dyM, S2[c] = sum dyM*xhat} (dyM = dy*[bn(x) > 0]);
 * for SyncBatchNorm all-reduce out2k and use the global count; then dx = gamma*rstd*(dyM - S1/n - xhat*S2/n).
 * d gamma = S2, d beta = S1.  `x` is the pre-normalisation activation, `stats` the forward {sum, sumsq}. */
int contrad_bn_relu_bwd_stats(const float* dy, const float* x, long long M, int K, int ld, const float* stats,
                              float count, const float* gamma, const float* beta, float eps, float* out2k,
                              float* workspace, long long workspace_bytes, contrad_stream_t stream);
int contrad_bn_relu_bwd_apply(const float* dy, const float* x, float* dx, long long M, int K, int ld,
                              const float* stats, float count, const float* gamma, const float* beta, float eps,
                              const float* bstats, contrad_stream_t stream);
/* running_mean / running_var update of nn.BatchNorm2d in train mode (unbiased variance); num_batches_tracked (int64
 * scalar on the device, may be NULL) is incremented by the same launch. */
int contrad_bn_running_update(const float* stats, float count, int K, const float* conv_bias, float momentum,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              contrad_stream_t stream);
/* contrad.loss_D_fn's GAN term (training/gan/contrad.py:51-64) on logits[3N] (stride ld): kind 0 nonsat,
 * 1 wgan, 2 hinge, 3 lsgan.  out3 = {loss, mean d_real, mean d_gen}; grad[3N] = d loss / d logits. */
int contrad_gan_d_loss(const float* logits, int ld, int N, int kind, float* out3, float* grad,
                       contrad_stream_t stream);
/* contrad.loss_G_fn (contrad.py:73-82) on logits[N]: kind 0 nonsat, 3 lsgan, other: -mean. */
int contrad_gan_g_loss(const float* logits, int ld, int N, int kind, float* out1, float* grad,
                       contrad_stream_t stream);

/* torch.optim.Adam step (train_gan.py:273-274; no weight decay / amsgrad) fused over up to 64 tensors per
 * launch; `step` is the 1-based step count of these tensors, grad_scale multiplies every gradient first
 * (1/world_size after a sum all-reduce). */
#define CONTRAD_ADAM_MAX_TENSORS 64
#define CONTRAD_ADAM_CHUNK 16384
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long numel;
} contrad_adam_tensor;
typedef struct {
  int n;
  contrad_adam_tensor t[CONTRAD_ADAM_MAX_TENSORS];
  int block_start[CONTRAD_ADAM_MAX_TENSORS + 1]; /* filled by the launcher */
} contrad_adam_batch;
int contrad_adam_step(const contrad_adam_batch* b, int step, float lr, float beta1, float beta2, float eps,
                      float grad_scale, contrad_stream_t stream);
/* The same update with the step-dependent scalars read from DEVICE memory: hyper_dev = {lr / (1 - beta1^t),
 * 1 / sqrt(1 - beta2^t), grad_scale}.  For a D-step captured once into a hipGraph and replayed every iteration (LR
 * warm-up and the bias corrections change per step; launch arguments are frozen at capture). */
int contrad_adam_step_dev(const contrad_adam_batch* b, const float* hyper_dev, float beta1, float beta2, float eps,
                          contrad_stream_t stream);
/* y = a*y + b*x (G EMA `accumulate`, utils.py:130-143) */
int contrad_axpby(float* y, const float* x, long long n, float a, float b, contrad_stream_t stream);
/* dst[0..n) (device) = host_pinned[0..n): a kernel reads device-mapped pinned host memory (hipHostMalloc) directly --
 * the stream-ordered hand-over of the per-step host-side random draws (latents, augmentation parameters). */
int contrad_pull_host(const float* host_pinned, float* dst, long long n, contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SimCLR augmentation, fused (augment/__init__.py:106-122 `simclr()` / `simclr_hq()`):
 * RandomResizeCropLayer + HorizontalFlipLayer (augment/spatial.py:84-148) as ONE bilinear/reflection
 * gather, RandomApply(ColorJitterLayer) (augment/color_jitter.py:16-104, augment/utils.py:6-63),
 * RandomApply(RandomColorGrayLayer) (augment/__init__.py:82-103).  NCHW in, NCHW out, x != y.
 * contrast_first < 0: the colour-op order is read per sample from params[.][15] (hipGraph replay).
 * params[B][CONTRAD_AUG_NPARAM] = {theta00, theta11, theta02, theta12, flip_sign, jitter_mask,
 * f_contrast, f_h, f_s, f_v, gray_mask, blur_mask, cutout_mask, cutout_h_center, cutout_w_center, contrast_first},
 * sampled on the host in the reference's draw order.
 * ---------------------------------------------------------------------------------------------- */
#define CONTRAD_AUG_NPARAM 16
long long contrad_simclr_workspace_bytes(int B, int H, int W);
int contrad_simclr_augment(const float* x, float* y, const float* params, int B, int H, int W,
                           int contrast_first, int has_contrast, float* workspace,
                           long long workspace_bytes, contrad_stream_t stream);
/* Backward of contrad_simclr_augment for the generator step (gradient flows through the augmentation into G,
 * training/gan/contrad.py:73-82, train_stylegan2_contraD.py:138-146): grad_in = d loss / d x given grad_out =
 * d loss / d y.  Bilinear gather transpose, contrast backward, straight-through HSV (augment/color_jitter.py:97-104),
 * gray backward.  Small images (7*H*W + H*H + W*W floats of LDS <= 64 KiB, i.e. CIFAR) run in one block per image and
 * need no workspace; larger images (AFHQ 512x512) take four launches and the workspace sized by the query. */
long long contrad_simclr_augment_bwd_workspace_bytes(int B, int H, int W);
int contrad_simclr_augment_bwd(const float* x, const float* params, const float* grad_out, float* grad_in,
                               int B, int H, int W, int contrast_first, int has_contrast, float* workspace,
                               long long workspace_bytes, contrad_stream_t stream);
/* RandomApply(GaussianBlur) (augment/__init__.py:53-78): separable (2*radius+1)-tap blur with reflect
 * padding on samples whose blur_mask != 0, copy-through otherwise; tmp is a scratch image batch. */
int contrad_gaussian_blur_masked(const float* x, float* tmp, float* y, const float* params,
                                 const float* kernel1d, int B, int H, int W, int radius,
                                 contrad_stream_t stream);
/* RandomApply(CutOut(length)) (augment/spatial.py:152-181; the last stage of `simclr_hq_cutout`, augment/__init__.py:
 * 124-133): in place, y[n,:,i,j] = 0 where |i - h_center| <= (length-1)/2 and |j - w_center| <= (length-1)/2, on the
 * samples whose cutout_mask != 0.  The op is its own backward (a 0/1 mask on the gradient). */
int contrad_cutout_masked(float* y, const float* params, int B, int H, int W, int length, contrad_stream_t stream);
/* Adjoint of contrad_gaussian_blur_masked (the generator step through simclr_hq): grad_in = blur^T(grad_out) on the
 * masked samples (reflect-padding transpose: border contributions fold back), copy-through otherwise. */
int contrad_gaussian_blur_masked_bwd(const float* grad_out, float* tmp, float* grad_in, const float* params,
                                     const float* kernel1d, int B, int H, int W, int radius,
                                     contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The reference's two native ops (models/gan/stylegan2/op/), same tensor contracts.
 * ---------------------------------------------------------------------------------------------- */
/* upfirdn2d_op.upfirdn2d (op/upfirdn2d.cpp:12-23): input [major,in_h,in_w,minor] -> out [major,out_h,out_w,minor],
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (likewise w); zero-insertion upsampling, padding (may be
 * negative), correlation with the flipped FIR kernel [kh,kw], decimation.  The NHWC build passes major = B,
 * minor = C.  Blur / Upsample / Downsample of stylegan2/layers.py:34-92 and their backward / double backward
 * (op/upfirdn2d.py:19-142) are all instances of this one call. */
int contrad_upfirdn2d(const float* input, const float* kernel, float* out, int major, int in_h, int in_w,
                      int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                      int pad_x1, int pad_y0, int pad_y1, contrad_stream_t stream);
/* The same op with a fused epilogue, used by the discriminator's first-order backward so that no separate elementwise
 * pass exists (the reference runs FusedLeakyReLUFunctionBackward and the residual-gradient adds as their own kernels,
 * op/fused_act.py:20-55, discriminator.py:72-74):  v = upfirdn2d(input) [+ addend];  out = v (if out != NULL);
 * out2 = v * (act_ref > 0 ? gain : slope * gain) (if out2 != NULL).  addend / act_ref / out2 have the output's shape. */
int contrad_upfirdn2d_fused(const float* input, const float* kernel, float* out, int major, int in_h, int in_w,
                            int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                            int pad_x1, int pad_y0, int pad_y1, const float* addend, const float* act_ref, float slope,
                            float gain, float* out2, contrad_stream_t stream);
/* fused.fused_bias_act (op/fused_bias_act.cpp:11-21): act 1 linear / 3 leaky-relu(alpha); grad 0: y = act(x +
 * bias[(i/step_b) % size_b]) * scale; grad 1: y = x * act'(ref) * scale; grad 2: y = 0.  bias / ref may be NULL
 * where unused.  (FusedLeakyReLU and its first / second backward, op/fused_act.py:20-71.) */
int contrad_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, long long n,
                           int step_b, int size_b, int act, int grad, float alpha, float scale,
                           contrad_stream_t stream);
/* y = a*x + b*z  (ResBlock merge (out + skip)/sqrt(2), stylegan2/discriminator.py:72-74) */
int contrad_lincomb(const float* x, const float* z, float* y, long long n, float a, float b,
                    contrad_stream_t stream);

/* Minibatch-stddev channel (_minibatch_stddev_layer, models/gan/stylegan2/discriminator.py:22-33) on NHWC, with the two
 * backward passes the R1 penalty needs (train_stylegan2.py:106-113 differentiates D's input gradient).  x [B][P][C] dense
 * (P = H*W); the output / gy layout is [B][P][Cp], Cp >= C + 1: channels [0, C) = x, channel C = the group statistic of sample
 * b mod (B / min(B, 4)), channels above = 0 (padding for the conv engine).
 *   mode 0  out [B][P][Cp] = forward(x)
 *   mode 1  out [B][P][C]  = d/dx of <forward(x), gy>                         (gy [B][P][Cp])
 *   mode 2  out [B][P][C], out2 [B][P][Cp] = d/dx, d/dgy of <mode-1 result, h>   (h [B][P][C]) */
int contrad_minibatch_stddev(int mode, const float* x, const float* gy, const float* h, float* out, float* out2,
                             int B, int P, int C, int Cp, contrad_stream_t stream);
/* out[0] = scale * sum_i x[i]^2 in a fixed summation order (r1_loss, train_stylegan2.py:112: grad.pow(2).sum / N) */
long long contrad_sumsq_workspace_bytes(long long n);
int contrad_sumsq(const float* x, long long n, float scale, float* out, float* workspace, long long workspace_bytes,
                  contrad_stream_t stream);
/* y = x * (c * s[0]), s a scalar in device memory (the backward of the line above) */
int contrad_scale_dev(const float* x, const float* s, float c, float* y, long long n, contrad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * StyleGAN2 generator forward helpers (models/gan/stylegan2/generator.py).
 * ---------------------------------------------------------------------------------------------- */
/* Forward-only weight tables of every ModulatedConv2d of the generator in ONE launch (generator.py:52-82): the packed
 * shared weight wp = scale * W in the GEMM layout the conv engine reads, and the demodulation table
 * wsq[c][k] = sum_taps (scale * W[k][c][tap])^2  (demod[n][k] = rsqrt(sum_c style[n][c]^2 wsq[c][k] + 1e-8)).
 * W is ModulatedConv2d.weight[0] = [Cout][Cin][T] as stored (no transposed copy):
 *   transposed == 0: wp[(tap * Cin + c)][k]  (k = cout; a plain conv: contrad_conv2d_fwd),
 *   transposed != 0: wp[(tap * Cout + k)][c] (the upsampling layers' transposed conv = contrad_conv2d_dgrad, and ToRGB =
 *                    contrad_rgb_conv_dgrad).  ldw = leading dimension of wp; columns past the last one are not written.
 * wsq may be NULL (ToRGB does not demodulate). */
#define CONTRAD_MODCONV_MAX_LAYERS 32
typedef struct {
  const float* w;
  float* wp;
  float* wsq;        /* [Cin][Cout] or NULL */
  int Cout, Cin, T, ldw;
  int transposed;
  float scale;
} contrad_modconv_layer;
typedef struct {
  int n;
  contrad_modconv_layer layers[CONTRAD_MODCONV_MAX_LAYERS];
} contrad_modconv_batch;
int contrad_modconv_tables(const contrad_modconv_batch* b, contrad_stream_t stream);
/* Demodulation factors of ALL demodulated layers of one generator forward in one launch (generator.py:62-64):
 * out_l[n][k] = rsqrt( sum_c style_l[n][c]^2 * wsq_l[c][k] + eps ), style_l [B][Cin] (rows contiguous), wsq_l [Cin][K]
 * from contrad_modconv_tables, out_l [B][K].  (Was per layer: a squaring pass, a split-K GEMM + its reduce, add, rsqrt.) */
typedef struct {
  const float* style;
  const float* wsq;
  float* out;
  int Cin, K;
} contrad_demod_layer;
typedef struct {
  int n, B;
  contrad_demod_layer layers[CONTRAD_MODCONV_MAX_LAYERS];
} contrad_demod_batch;
int contrad_modconv_demod(const contrad_demod_batch* b, float eps, contrad_stream_t stream);
/* PixelNorm (stylegan2/layers.py:14-19): y = x * rsqrt(mean_c(x^2) + 1e-8) over rows of [M][K]. */
int contrad_pixelnorm(const float* x, float* y, int M, int K, contrad_stream_t stream);
/* y[n,h,w,c] = x[n,h,w,c] * s[n,c]: weight modulation moved onto the input channels of the shared-weight conv
 * (ModulatedConv2d, generator.py:52-60: (scale * W * style) applied to x == conv(x * style, scale * W)). */
int contrad_nhwc_scale(const float* x, const float* s, float* y, int N, long long HW, int C,
                       contrad_stream_t stream);
/* y = sqrt2 * lrelu_0.2( x * demod[n,k] + noise_w[0] * noise[n,h,w] + bias[k] ), in place allowed:
 * demodulation (generator.py:62-64) + NoiseInjection (:85-94) + FusedLeakyReLU (:113-118) in one pass.
 * demod / noise may be NULL.  post_scale [N,K] (may be NULL): y is additionally multiplied by post_scale[n,k] -- the
 * style vector of the layer that consumes y, i.e. that layer's contrad_nhwc_scale pass folded into this store. */
int contrad_modconv_epilogue(const float* x, const float* demod, const float* noise, const float* noise_w,
                             const float* bias, const float* post_scale, float* y, int N, long long HW, int K,
                             contrad_stream_t stream);
/* The upsampling StyledConv's tail in ONE pass (generator.py:80-83,97-118: Blur after the transposed conv, then
 * demodulation + NoiseInjection + FusedLeakyReLU):  y = modconv_epilogue( upfirdn2d(input; 4x4 FIR, up = down = 1) ),
 * input [N,in_h,in_w,K] -> y [N,out_h,out_w,K]; demod [N,K] / noise [N,out_h,out_w] / post_scale [N,K] may be NULL.
 * post_scale is the NEXT layer's style vector: its weight modulation (contrad_nhwc_scale) rides on this store. */
int contrad_upfirdn2d_modconv(const float* input, const float* kernel, float* y, int N, int in_h, int in_w, int K,
                              int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* demod, const float* noise,
                              const float* noise_w, const float* bias, const float* post_scale,
                              contrad_stream_t stream);
/* out[n,c] = sum_hw a[n,hw,c] * b[n,hw,c] (b_per_channel != 0) or sum_hw a[n,hw,c] * b[n,hw] (b broadcast over the
 * channels): the gradients of the style vector, the demodulation factor and the noise strength in the backward of
 * ModulatedConv2d / NoiseInjection (generator.py:52-94; the reference gets them from autograd over its grouped
 * conv).  Deterministic two-stage reduction; workspace from contrad_nhwc_dot_workspace_bytes. */
long long contrad_nhwc_dot_workspace_bytes(int N, long long HW, int C);
int contrad_nhwc_dot(const float* a, const float* b, float* out, int N, long long HW, int C, int b_per_channel,
                     float* workspace, long long workspace_bytes, contrad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTRAD_HIP_H */
