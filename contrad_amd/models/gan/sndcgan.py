"""SNDCGAN generator / discriminator on the HIP kernel library -- counterpart of models/gan/sndcgan.py.

``D_SNDCGAN`` (reference sndcgan.py:69-148 + base.py:79-150) runs as ONE autograd node: the forward chains
the batched spectral-norm weight prep, the fused RGB first conv, six implicit-GEMM convs and the merged head
GEMMs; the backward chains dgrad (with the producer's LeakyReLU derivative fused in its epilogue), wgrad, bias
column sums and the batched spectral-norm weight gradient.  Activations are NHWC internally; inputs/outputs
keep the reference's NCHW / (B, features) contract, state-dict names match the reference.

``G_SNDCGAN`` (sndcgan.py:13-66) implements the no-grad forward used inside the discriminator step
(train_gan.py:155-156): linear -> BatchNorm(batch stats, SyncBN across ranks) -> ReLU -> 3 x [transposed conv
(= conv dgrad kernel) -> BatchNorm -> ReLU] -> transposed conv + tanh -> 0.5x+0.5.
"""
import torch
import torch.nn as nn
import torch.distributed as dist

from ... import ops
from ...hostio import upload
from ... import autograd_ops as A
from .base import BaseDiscriminator, SNParams, TinyHead, _Act, make_projection

# (cin, cout, k, stride, pad) -- reference sndcgan.py:91-109
_D_CONVS = [(3, 64, 3, 1, 1), (64, 128, 4, 2, 1), (128, 128, 3, 1, 1), (128, 256, 4, 2, 1),
            (256, 256, 3, 1, 1), (256, 512, 4, 2, 1), (512, 512, 3, 1, 1)]
_SLOPE = 0.1


def _flat_views(total_sizes, device):
    """One allocation, list of 1-D views."""
    offs, tot = [], 0
    for n in total_sizes:
        offs.append(tot)
        tot += ops.round_up(n, 4)
    buf = torch.empty(max(tot, 4), device=device, dtype=torch.float32)
    return buf, [buf[o:o + n] for o, n in zip(offs, total_sizes)]


class _DPlan(object):
    """Per-forward packed-weight / snapshot storage of the discriminator (lives in the autograd ctx)."""

    def __init__(self, D, device):
        self.hb = D.s_hb
        self.wb = D.s_wb
        dh, dp = D.d_hidden, D.d_project
        self.dh, self.dp = dh, dp
        convs = [D.main[2 * i] for i in range(7)]
        heads1 = [D.linear.l1, D.projection[0], D.projection2[0]]
        heads2 = [D.linear.l2, D.projection[2], D.projection2[2]]
        self.layers = convs + heads1 + heads2
        T_head = self.hb * self.wb
        self.specs = []
        for m in convs:
            self.specs.append(ops.SnSpec(m.weight_orig, m.weight_u, m.weight_v))
        for m in heads1:   # Linear over the NCHW-flattened 512 x hb x wb features == hb x wb "conv" on NHWC
            self.specs.append(ops.SnSpec(m.weight_orig, m.weight_u, m.weight_v, view_kct=(dh, 512, T_head)))
        for m in heads2:
            self.specs.append(ops.SnSpec(m.weight_orig, m.weight_u, m.weight_v))
        # packed layouts
        sizes = [s.T * s.C * s.K for s in self.specs[:7]]
        sizes.append(T_head * 512 * 3 * dh)               # merged first head layers [feat][3*dh]
        sizes += [dh * 4, dh * dp, dh * dp]               # second head layers ([dh][4] for the 1-wide logit)
        self.pack_sizes = sizes
        self.feat = T_head * 512

    def alloc_packed(self, device):
        buf, v = _flat_views(self.pack_sizes, device)
        dh, dp = self.dh, self.dp
        wps, ldws = [], []
        for i, s in enumerate(self.specs[:7]):
            wps.append(v[i].view(s.T * s.C, s.K)); ldws.append(s.K)
        merged = v[7].view(self.feat, 3 * dh)
        for j in range(3):
            wps.append(merged[:, j * dh:(j + 1) * dh]); ldws.append(3 * dh)
        wps.append(v[8].view(dh, 4)); ldws.append(4)
        wps.append(v[9].view(dh, dp)); ldws.append(dp)
        wps.append(v[10].view(dh, dp)); ldws.append(dp)
        return buf, wps, ldws, merged


class _DFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, D, images, sg_linear, want_features, trunk_grad, *params):
        dev = images.device
        plan = _DPlan(D, dev)
        specs = plan.specs
        n = len(specs)
        training = D.training
        B = images.shape[0]
        dh, dp = plan.dh, plan.dp

        wbuf, wps, ldws, merged = plan.alloc_packed(dev)
        offs, scr_n = ops.sn_scratch_floats(specs)
        snap_sizes = []
        for s in specs:
            snap_sizes += [s.K, s.C * s.T]
        snapbuf, snaps = _flat_views(snap_sizes, dev)
        u_snaps, v_snaps = snaps[0::2], snaps[1::2]
        scratch = torch.empty(scr_n, device=dev, dtype=torch.float32)
        sigma = torch.empty(n, device=dev, dtype=torch.float32)
        if trunk_grad:
            ops.sn_weight_prep(specs, wps, ldws, training, scratch, offs, sigma, u_snaps, v_snaps)
        else:
            # finetuning (base.py:111-119): the features come from the network in eval mode -- no power iteration in the
            # seven trunk layers -- while the heads stay in the caller's mode
            ops.sn_weight_prep(specs[:7], wps[:7], ldws[:7], False, scratch, offs[:7], sigma, u_snaps[:7], v_snaps[:7])
            ops.sn_weight_prep(specs[7:], wps[7:], ldws[7:], training, scratch, offs[7:], sigma[7:], u_snaps[7:],
                               v_snaps[7:])

        biases = [m.bias for m in plan.layers]
        acts = []
        x = ops.rgb_conv_fwd(images, wps[0], biases[0], 64, 3, 2.0, -1.0, _SLOPE, 1.0)
        acts.append(x)
        for i in range(1, 7):
            ci, co, k, s, p = _D_CONVS[i]
            x = ops.conv2d_fwd(x, wps[i], biases[i], co, k, k, s, p, _SLOPE, 1.0)
            acts.append(x)
        bias_cat = torch.cat([biases[7], biases[8], biases[9]])
        # first head layers: Linear over the flattened features == 1x1 conv on the (B,1,1,hb*wb*512) NHWC view (the
        # packed rows (tap*512 + c) are exactly the NHWC-flat feature index)
        hidden = ops.conv2d_fwd(x.view(B, 1, 1, plan.feat), merged, bias_cat, 3 * dh, 1, 1, 1, 0, _SLOPE, 1.0)
        logits = ops.conv2d_fwd(hidden[..., 0:dh], wps[10], biases[10], 1, 1, 1, 1, 0).view(B, 1)
        proj = ops.conv2d_fwd(hidden[..., dh:2 * dh], wps[11], biases[11], dp, 1, 1, 1, 0).view(B, dp)
        proj2 = ops.conv2d_fwd(hidden[..., 2 * dh:3 * dh], wps[12], biases[12], dp, 1, 1, 1, 0).view(B, dp)

        if getattr(D, '_record_activations', False):     # test hook: expose the linear regions actually used
            D._last_activations = (acts, hidden)
        ctx.plan, ctx.D = plan, D
        ctx.images, ctx.acts, ctx.hidden = images, acts, hidden
        ctx.wps, ctx.ldws, ctx.merged, ctx.wbuf = wps, ldws, merged, wbuf
        ctx.sigma, ctx.u_snaps, ctx.v_snaps, ctx.snapbuf = sigma, u_snaps, v_snaps, snapbuf
        ctx.sg_linear, ctx.trunk_grad = sg_linear, trunk_grad
        ctx.param_shapes = [tuple(p.shape) for p in params]
        if want_features:
            feats = x.permute(0, 3, 1, 2).reshape(B, -1)      # NCHW-flattened, as the reference returns it
        else:
            feats = logits.new_empty(0)
            ctx.mark_non_differentiable(feats)
        return logits, proj, proj2, feats

    @staticmethod
    def backward(ctx, g_logits, g_proj, g_proj2, g_feats):
        plan, D = ctx.plan, ctx.D
        specs, wps, ldws = plan.specs, ctx.wps, ctx.ldws
        acts, hidden, images = ctx.acts, ctx.hidden, ctx.images
        dev = images.device
        B = images.shape[0]
        dh, dp = plan.dh, plan.dp
        n = len(specs)
        need_params = any(ctx.needs_input_grad[5:])
        need_images = ctx.needs_input_grad[1]
        a6 = acts[6]
        comm = getattr(D, '_grad_comm', None)      # engine.OverlappedGradReducer or None

        def exchange(t):
            if comm is not None:
                comm.reduce_async(t)

        def _c(g, shape):
            if g is None:
                return torch.zeros(shape, device=dev, dtype=torch.float32)
            return g.contiguous()

        g_logits = _c(g_logits, (B, 1)).view(B, 1, 1, 1)
        g_proj = _c(g_proj, (B, dp)).view(B, 1, 1, dp)
        g_proj2 = _c(g_proj2, (B, dp)).view(B, 1, 1, dp)

        # packed weight-gradient storage mirrors the packed weights
        if need_params:
            gbuf, gv = _flat_views(plan.pack_sizes, dev)
            gwps = []
            for i, s in enumerate(specs[:7]):
                gwps.append(gv[i].view(s.T * s.C, s.K))
            gmerged = gv[7].view(plan.feat, 3 * dh)
            for j in range(3):
                gwps.append(gmerged[:, j * dh:(j + 1) * dh])
            gwps += [gv[8].view(dh, 4), gv[9].view(dh, dp), gv[10].view(dh, dp)]
            bsizes = [m.bias.numel() for m in plan.layers]
            bbuf, gbias = _flat_views(bsizes, dev)

        # ---- heads, second layers ----
        g_hidden = torch.empty((B, 1, 1, 3 * dh), device=dev, dtype=torch.float32)
        outs = [(g_logits, 10, 0), (g_proj, 11, 1), (g_proj2, 12, 2)]
        for g, li, j in outs:
            hs = hidden[..., j * dh:(j + 1) * dh]
            ops.conv2d_dgrad(g, wps[li], (B, 1, 1, dh), 1, 1, 1, 0, act_ref=hs, slope=_SLOPE, gain=1.0,
                             out=g_hidden[..., j * dh:(j + 1) * dh])
            if need_params:
                if g.shape[3] % 4 == 0:
                    ops.conv2d_wgrad(hs, g, 1, 1, 1, 0, out=gwps[li], dbias=gbias[li])
                else:                                     # the 1-wide logit layer takes the scalar path
                    ops.conv2d_wgrad(hs, g, 1, 1, 1, 0, out=gwps[li])
                    ops.colstats(ops.as_rows(g), out=gbias[li].view(1, -1))
                exchange(gwps[li])
        # ---- heads, first (merged) layer ----
        if need_params:
            gb_hidden = torch.empty(3 * dh, device=dev, dtype=torch.float32)
            ops.conv2d_wgrad(a6.view(B, 1, 1, plan.feat), g_hidden, 1, 1, 1, 0, out=gmerged, dbias=gb_hidden)
            exchange(gmerged)                  # 12.6 M of the 18.6 M parameters: hidden behind the trunk backward
            for j in range(3):
                gbias[7 + j].copy_(gb_hidden[j * dh:(j + 1) * dh])
        g = None
        if ctx.trunk_grad:
            lo = dh if ctx.sg_linear else 0
            fused = g_feats is None or g_feats.numel() == 0
            g = ops.conv2d_dgrad(g_hidden[..., lo:], ctx.merged[:, lo:], (B, 1, 1, plan.feat), 1, 1, 1, 0,
                                 act_ref=a6.view(B, 1, 1, plan.feat) if fused else None, slope=_SLOPE,
                                 gain=1.0).view(a6.shape)
            if not fused:   # gradient arriving through aux['penultimate'] (not on the hot path)
                g = g + g_feats.reshape(B, 512, plan.hb, plan.wb).permute(0, 2, 3, 1)
                g = (g * torch.where(a6 > 0, 1.0, _SLOPE)).contiguous()
            # ---- trunk ----
            for i in range(6, 0, -1):
                ci, co, k, s, p = _D_CONVS[i]
                if need_params:     # weight gradient + bias gradient (column sums of g) in one kernel
                    ops.conv2d_wgrad(acts[i - 1], g, k, k, s, p, out=gwps[i], dbias=gbias[i])
                    exchange(gwps[i])
                g = ops.conv2d_dgrad(g, wps[i], tuple(acts[i - 1].shape), k, k, s, p, act_ref=acts[i - 1],
                                     slope=_SLOPE, gain=1.0)
            if need_params:
                ops.rgb_conv_wgrad(images, g, 3, 2.0, -1.0, gwps[0], gbias[0])
                exchange(gwps[0])
        d_images = None
        if need_images and g is not None:
            d_images = ops.rgb_conv_dgrad(g, wps[0], None, 3, 3, act=0, out_scale=2.0, out_shift=0.0)

        grads = [None] * (2 * n)
        if need_params:
            if g is None:   # trunk frozen (finetuning): its packed gradients are exactly zero
                for i in range(7):
                    gwps[i].zero_(); gbias[i].zero_()
                    exchange(gwps[i])
            if comm is not None:
                exchange(bbuf)
                comm.wait()     # compute stream waits for the collectives; the host does not block
            wsizes = [s.K * s.C * s.T for s in specs]
            gwbuf, gws = _flat_views(wsizes, dev)
            offs, scr_n = ops.sn_scratch_floats(specs)
            scratch = torch.empty(scr_n, device=dev, dtype=torch.float32)
            ops.sn_weight_grad(specs, wps, ldws, gwps, gws, scratch, offs, ctx.sigma, ctx.u_snaps, ctx.v_snaps)
            for i in range(n):
                grads[2 * i] = gws[i].view(ctx.param_shapes[2 * i])
                grads[2 * i + 1] = gbias[i]
        return (None, d_images, None, None, None) + tuple(grads)


class D_SNDCGAN(BaseDiscriminator):
    """Drop-in for the reference's D_SNDCGAN(image_size, mlp_linear=True, d_hidden=512) (sndcgan.py:69-148)."""

    def __init__(self, image_size, ndf=64, n_classes=1, normalize=False, disable_sn=False, mlp_linear=True,
                 d_hidden=512, d_project=128):
        super().__init__()
        if ndf != 64 or n_classes != 1 or normalize or disable_sn or not mlp_linear:
            raise NotImplementedError('only the configuration built by get_architecture("sndcgan") is implemented')
        s_h, s_w, nc = image_size
        if nc != 3 or s_h % 8 or s_w % 8:
            raise NotImplementedError('RGB images with sides divisible by 8')
        self.image_size = image_size
        self.s_hb, self.s_wb = s_h // 8, s_w // 8
        self.n_features = 512 * self.s_hb * self.s_wb
        self.d_penul = self.n_features
        self.n_classes, self.d_hidden, self.d_project = n_classes, d_hidden, d_project

        self.linear = TinyHead(self.n_features, d_hidden, spectral=True)
        self.projection = make_projection(self.n_features, d_hidden, d_project, spectral=True)
        self.projection2 = make_projection(self.n_features, d_hidden, d_project, spectral=True)
        layers = []
        for (ci, co, k, s, p) in _D_CONVS:
            layers += [SNParams((co, ci, k, k)), _Act(_SLOPE)]
        self.main = nn.Sequential(*layers)

    def enable_grad_overlap(self, comm):
        """Exchange gradients inside the backward (engine.OverlappedGradReducer); pass None to disable."""
        self._grad_comm = comm

    def _ordered_params(self):
        out = []
        for m in ([self.main[2 * i] for i in range(7)] +
                  [self.linear.l1, self.projection[0], self.projection2[0],
                   self.linear.l2, self.projection[2], self.projection2[2]]):
            out += [m.weight_orig, m.bias]
        return out

    def _run(self, inputs, sg_linear, finetuning, want_features):
        if not inputs.is_cuda:
            raise RuntimeError('contrad_amd.D_SNDCGAN runs on the MI355X HIP path only (no CPU fallback)')
        if finetuning:
            # reference: features in eval mode under no_grad (base.py:114-119): no trunk gradient, no power iteration
            # in the trunk (handled inside the node); the heads still train
            inputs = inputs.detach()
        images = inputs.contiguous().float()
        logits, proj, proj2, feats = _DFunction.apply(self, images, bool(sg_linear), bool(want_features),
                                                      not finetuning, *self._ordered_params())
        return logits, proj, proj2, (feats if want_features else None)

    def penultimate(self, inputs):
        return self._run(inputs, False, False, True)[3]


# ------------------------------------------------------------------------------------------------------
class BNParams(nn.Module):
    """nn.BatchNorm2d's tensors under the same names (weight, bias, running_mean, running_var,
    num_batches_tracked)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))


def _sync_on(flag):
    from ...engine import dist_on
    return flag and dist_on()


class _BNReLUFn(torch.autograd.Function):
    """BatchNorm (train mode, batch statistics; SyncBN across ranks) + ReLU on an (M, K) row matrix, forward and
    first-order backward on the HIP kernels (nn.BatchNorm2d + ReLU of G_SNDCGAN, sndcgan.py:25-35,43)."""

    @staticmethod
    def forward(ctx, x2d, gamma, beta, bn, conv_bias, sync):
        x2d = x2d.contiguous()
        stats = ops.colstats(x2d, with_sq=True)
        count = float(x2d.shape[0])
        if _sync_on(sync):
            dist.all_reduce(stats)
            count *= dist.get_world_size()
        if bn.training:
            ops.bn_running_update(stats, count, conv_bias, bn.momentum, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked)
        y = torch.empty_like(x2d)
        ops.bn_relu_apply(x2d, y, stats, count, gamma, beta, bn.eps)
        ctx.save_for_backward(x2d, stats, gamma, beta)
        ctx.cfg = (count, bn.eps, sync)
        ctx.cb_shape = tuple(conv_bias.shape) if (conv_bias is not None and conv_bias.requires_grad) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, stats, gamma, beta = ctx.saved_tensors
        count, eps, sync = ctx.cfg
        red = (lambda t: dist.all_reduce(t)) if _sync_on(sync) else None
        dx, dgamma, dbeta = ops.bn_relu_bwd(dy.contiguous(), x2d, stats, count, gamma, beta, eps, red)
        # a bias added in front of a batch-statistics BatchNorm cancels exactly: its true gradient is zero (the
        # reference's autograd returns ~1e-9 round-off noise here); hand Adam an exact zero instead of None
        dcb = torch.zeros(ctx.cb_shape, device=dy.device, dtype=dy.dtype) if ctx.cb_shape is not None else None
        return dx, dgamma, dbeta, None, dcb, None


class _WB(nn.Module):
    def __init__(self, wshape, nbias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape).normal_(0.0, 0.02))
        self.bias = nn.Parameter(torch.zeros(nbias))


class G_SNDCGAN(nn.Module):
    """Drop-in for the reference's G_SNDCGAN (sndcgan.py:13-66): same state-dict, sample_latent on the CPU
    generator; forward under no_grad on the HIP path (the discriminator step's fake batch in train mode; in eval mode
    the running statistics normalise -- sampling from a ``gen.pt`` checkpoint, train_gan.py:181); with gradients
    enabled (generator step, train mode) the same network out of differentiable HIP nodes."""

    _CONVT = [(512, 256, 4, 2, 1), (256, 128, 4, 2, 1), (128, 64, 4, 2, 1), (64, 3, 3, 1, 1)]

    def __init__(self, image_size, ngf=64, nz=128):
        super().__init__()
        if ngf != 64:
            raise NotImplementedError('ngf = 64')
        s_h, s_w, nc = image_size
        self.image_size, self.ngf, self.nz = image_size, ngf, nz
        self.s_hb, self.s_wb = s_h // 8, s_w // 8
        f = ngf * 8 * self.s_hb * self.s_wb
        self.linear = _WB((f, nz), f)
        self.norm_init = BNParams(f)
        layers = []
        for j, (ci, co, k, s, p) in enumerate(self._CONVT):
            layers.append(_WB((ci, co, k, k), co))
            if j < 3:
                layers += [BNParams(co), _Act(0.0)]
            else:
                layers.append(_Act(1.0))          # Tanh placeholder (index 10)
        self.main = nn.Sequential(*layers)
        self.sync_bn = True
        self._packed = None
        self._packed_key = None

    def sample_latent(self, n_samples):
        # U(-1,1) from the CPU generator, as the reference (sndcgan.py:50-52); handed over by contrad_amd/hostio.py
        _device = next(self.parameters()).device
        return upload(torch.empty(n_samples, self.nz).uniform_(-1, 1), _device)

    def invalidate_cache(self):
        """Forget the packed weights: the next forward re-packs (a captured step must RECORD the pack launch, and the
        buffers a capture allocated hold nothing until the first replay)."""
        self._packed, self._packed_key = None, None

    def _weights(self):
        ws = [self.linear.weight] + [self.main[3 * j].weight for j in range(4)]
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if self._packed_key != key:
            dev = ws[0].device
            specs = [ops.SnSpec(w, fixed_scale=1.0) for w in ws]
            sizes = [s.K * s.C * s.T for s in specs]
            buf, v = _flat_views(sizes, dev)
            wps = [v[i].view(s.T * s.C, s.K) for i, s in enumerate(specs)]
            ldws = [s.K for s in specs]
            offs, scr_n = ops.sn_scratch_floats(specs)
            scratch = torch.empty(scr_n, device=dev, dtype=torch.float32)
            sigma = torch.empty(len(specs), device=dev, dtype=torch.float32)
            ops.sn_weight_prep(specs, wps, ldws, False, scratch, offs, sigma)
            self._packed, self._packed_key, self._packed_buf = wps, key, buf
        return self._packed

    def _bn(self, x2d, bn, conv_bias, out2d, perm_hw=1):
        if not self.training:
            # eval mode (sampling from a checkpoint, train_gan.py:181): the running statistics normalise.  Same apply
            # kernel, fed with "one sample" whose sum / sum of squares reproduce mean = running_mean - conv bias (x2d is
            # the transposed conv WITHOUT its bias) and biased variance = running_var
            mean = bn.running_mean if conv_bias is None else bn.running_mean - conv_bias
            stats = torch.stack([mean, bn.running_var + mean * mean]).contiguous()
            ops.bn_relu_apply(x2d, out2d, stats, 1.0, bn.weight, bn.bias, bn.eps, perm_hw)
            return
        stats = ops.colstats(x2d, with_sq=True)
        count = float(x2d.shape[0])
        if _sync_on(self.sync_bn):
            # SyncBatchNorm (train_gan.py:268): one packed all-reduce of {sum, sumsq} per layer over RCCL
            dist.all_reduce(stats)
            count *= dist.get_world_size()
        if self.training:
            ops.bn_running_update(stats, count, conv_bias, bn.momentum, bn.running_mean, bn.running_var,
                                  bn.num_batches_tracked)
        ops.bn_relu_apply(x2d, out2d, stats, count, bn.weight, bn.bias, bn.eps, perm_hw)

    def _forward_with_grad(self, z):
        """Generator step (train_gan.py:173-176): the same network out of differentiable HIP nodes -- transposed
        convs are ConvDgradFn (whose backward is the conv forward / wgrad kernels), BatchNorm+ReLU is _BNReLUFn."""
        N = z.shape[0]
        hb, wb = self.s_hb, self.s_wb
        f = 512 * hb * wb
        entries = [(f, self.nz, 1, 1.0, 0, 0)]
        groups = [(self.nz, f)]
        ws = [self.linear.weight]
        for j, (ci, co, k, s, p) in enumerate(self._CONVT):
            groups.append((k * k * co, ci))
            entries.append((ci, co, k * k, 1.0, j + 1, 0))
            ws.append(self.main[3 * j].weight)
        wp = A.PackWeightsFn.apply(A.PackMeta(entries, groups), *ws)
        z = z.contiguous().float()
        h0 = A.ConvBiasActFn.apply(z.view(N, 1, 1, self.nz), wp[0], self.linear.bias, (f, 1, 1, 1, 0), 1.0, 1.0)
        y0 = _BNReLUFn.apply(h0.view(N, f), self.norm_init.weight, self.norm_init.bias, self.norm_init, None,
                             self.sync_bn)
        x = y0.view(N, 512, hb, wb).permute(0, 2, 3, 1).contiguous()
        for j in range(3):
            ci, co, k, s, p = self._CONVT[j]
            H, W = x.shape[1], x.shape[2]
            y = A.ConvDgradFn.apply(x, wp[1 + j], (N, 2 * H, 2 * W, co), (ci, k, k, s, p))
            bn = self.main[3 * j + 1]
            y2 = _BNReLUFn.apply(y.view(-1, co), bn.weight, bn.bias, bn, self.main[3 * j].bias, self.sync_bn)
            x = y2.view(N, 2 * H, 2 * W, co)
        lin = A.RgbDgradFn.apply(x, wp[4], 3, (3, 1.0))
        return 0.5 * torch.tanh(lin + self.main[9].bias.view(1, 3, 1, 1)) + 0.5

    def forward(self, z):
        if not z.is_cuda:
            raise RuntimeError('contrad_amd.G_SNDCGAN runs on the MI355X HIP path only (no CPU fallback)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if not self.training:
                raise NotImplementedError('eval-mode BatchNorm (running statistics) is not on the training path')
            return self._forward_with_grad(z)
        wps = self._weights()
        N = z.shape[0]
        hb, wb = self.s_hb, self.s_wb
        f = 512 * hb * wb
        z = z.contiguous().float()
        h0 = ops.conv2d_fwd(z.view(N, 1, 1, self.nz), wps[0], self.linear.bias, f, 1, 1, 1, 0).view(N, f)
        x = torch.empty((N, hb, wb, 512), device=z.device, dtype=torch.float32)
        self._bn(h0, self.norm_init, None, x.view(N, f), perm_hw=hb * wb)
        for j in range(3):
            ci, co, k, s, p = self._CONVT[j]
            H, W = x.shape[1], x.shape[2]
            y = ops.conv2d_dgrad(x, wps[1 + j], (N, 2 * H, 2 * W, co), k, k, s, p)
            y2 = ops.as_rows(y)
            self._bn(y2, self.main[3 * j + 1], self.main[3 * j].bias, y2)
            x = y
        return ops.rgb_conv_dgrad(x, wps[4], self.main[9].bias, 3, 3, act=1, out_scale=0.5, out_shift=0.5)
