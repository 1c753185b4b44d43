"""ContraD losses -- counterpart of training/gan/contrad.py.

``loss_D_fn`` keeps the reference's signature and return contract (a ``(simclr + lbd_a * sup, {"penalty",
"d_real", "d_gen"})`` pair; note the GAN loss really is returned under the key "penalty", contrad.py:66-70) but
evaluates the two contrastive terms with ONE fused autograd node: row-normalise both projections, (distributed:
pack the 2N + 3N embedding rows into a single RCCL all-gather instead of the reference's five), fused
cosine/log-sum-exp kernels, local-slice backward.
"""
import torch
import torch.distributed as dist

from ... import ops
from ...third_party.gather_layer import GatherLayer, all_gather_rows
from ..criterion import _Contrast, MODE_NT_XENT, MODE_SUPCON_FAKE


def supcon_fake(out1, out2, others, temperature, distributed=False):
    """Same signature/semantics as training/gan/contrad.py:8-32 (inputs already L2-normalised)."""
    if distributed:
        out1 = torch.cat(GatherLayer.apply(out1), dim=0)
        out2 = torch.cat(GatherLayer.apply(out2), dim=0)
        others = torch.cat(GatherLayer.apply(others), dim=0)
    N = out1.size(0)
    return _Contrast.apply(torch.cat([out1, out2, others], dim=0), N, MODE_SUPCON_FAKE, temperature)


def _regroup(gathered, blocks, n):
    """(W, blocks*n, D) rank-major -> (blocks*W*n, D) block-major, i.e. cat_b(cat_rank(block b))."""
    W, _, D = gathered.shape
    return gathered.view(W, blocks, n, D).transpose(0, 1).reshape(blocks * W * n, D)


class _ContraDContrastive(torch.autograd.Function):
    """(projection (3N,D), projection2 (3N,D)) -> (nt_xent on F.normalize(projection)[:2N],
    supcon_fake on F.normalize(projection2)), contrad.py:42-50."""

    @staticmethod
    def forward(ctx, proj, proj2, N, temperature, distributed):
        D = proj.shape[1]
        z1, inv1 = ops.l2norm_fwd(proj[:2 * N])
        z2, inv2 = ops.l2norm_fwd(proj2)
        from ...engine import dist_on
        world, rank = 1, 0
        if distributed and dist_on():
            world, rank = dist.get_world_size(), dist.get_rank()
            packed = torch.cat([z1, z2], dim=0)                      # (5N, D): one message per rank
            g = all_gather_rows(packed)                              # (W, 5N, D) over RCCL / xGMI
            z1g = _regroup(g[:, :2 * N], 2, N)
            z2g = _regroup(g[:, 2 * N:], 3, N)
        else:
            z1g, z2g = z1, z2
        Ng = N * world
        l1, lse1 = ops.contrast_fwd(z1g, Ng, MODE_NT_XENT, temperature)
        l2, lse2 = ops.contrast_fwd(z2g, Ng, MODE_SUPCON_FAKE, temperature)
        ctx.save_for_backward(z1, inv1, z2, inv2, z1g, z2g, lse1, lse2)
        ctx.cfg = (N, Ng, temperature, world, rank, tuple(proj.shape), D)
        return l1.reshape(()), l2.reshape(())

    @staticmethod
    def backward(ctx, g1, g2):
        z1, inv1, z2, inv2, z1g, z2g, lse1, lse2 = ctx.saved_tensors
        N, Ng, temperature, world, rank, pshape, D = ctx.cfg
        dz1 = ops.contrast_bwd(z1g, lse1, Ng, MODE_NT_XENT, temperature, g1.reshape(1).contiguous().float())
        dz2 = ops.contrast_bwd(z2g, lse2, Ng, MODE_SUPCON_FAKE, temperature, g2.reshape(1).contiguous().float())
        if world > 1:   # GatherLayer.backward semantics: keep this rank's rows only
            dz1 = dz1.view(2, world, N, D)[:, rank].reshape(2 * N, D).contiguous()
            dz2 = dz2.view(3, world, N, D)[:, rank].reshape(3 * N, D).contiguous()
        dproj = torch.zeros(pshape, device=z1.device, dtype=torch.float32)
        ops.l2norm_bwd(dz1, z1, inv1, out=dproj[:2 * N])
        dproj2 = ops.l2norm_bwd(dz2, z2, inv2)
        return dproj, dproj2, None, None, None


class _GanDLoss(torch.autograd.Function):
    """contrad.py:51-64 on logits (3N,1): returns [loss, mean d_real, mean d_gen]."""

    @staticmethod
    def forward(ctx, d_all, N, kind):
        out, grad = ops.gan_d_loss(d_all.contiguous(), N, kind)
        ctx.save_for_backward(grad)
        loss, d_real, d_gen = out.unbind(0)
        ctx.mark_non_differentiable(d_real, d_gen)
        return loss, d_real, d_gen

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        grad, = ctx.saved_tensors
        return grad * g, None, None


class _GanGLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d_gen, kind):
        out, grad = ops.gan_g_loss(d_gen.contiguous(), kind)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return grad * g, None


_D_LOSSES = ('nonsat', 'wgan', 'hinge', 'lsgan')


def loss_D_fn(P, D, options, images, gen_images):
    assert images.size(0) == gen_images.size(0)
    if options['loss'] not in _D_LOSSES:
        raise NotImplementedError()
    gen_images = gen_images.detach()
    N = images.size(0)

    cat_images = torch.cat([images, images, gen_images], dim=0)
    d_all, aux = D(P.augment_fn(cat_images), sg_linear=True, projection=True, projection2=True)
    simclr_loss, sup_loss = _ContraDContrastive.apply(aux['projection'], aux['projection2'], N, P.temp,
                                                      bool(P.distributed))
    d_loss, d_real, d_gen = _GanDLoss.apply(d_all, N, options['loss'])
    return simclr_loss + P.lbd_a * sup_loss, {
        "penalty": d_loss,
        "d_real": d_real,
        "d_gen": d_gen,
    }


def loss_G_fn(P, D, options, images, gen_images):
    d_gen = D(P.augment_fn(gen_images))
    return _GanGLoss.apply(d_gen, options['loss'])
