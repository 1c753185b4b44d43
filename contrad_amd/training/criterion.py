"""Counterpart of training/criterion.py: ``nt_xent`` on the fused HIP contrastive kernel (the similarity matrix
is never written to HBM; forward keeps only the per-row log-sum-exp, backward recomputes the tiles)."""
import torch
import torch.distributed as dist

from .. import ops
from ..third_party.gather_layer import GatherLayer

MODE_NT_XENT = 0
MODE_SUPCON_FAKE = 1


class _RowNormalize(torch.autograd.Function):
    """F.normalize(x) (dim=1, eps=1e-12) forward/backward on the HIP row-norm kernels."""

    @staticmethod
    def forward(ctx, u):
        z, inv = ops.l2norm_fwd(u.contiguous())
        ctx.save_for_backward(z, inv)
        return z

    @staticmethod
    def backward(ctx, dz):
        z, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dz.contiguous(), z, inv)


class _Contrast(torch.autograd.Function):
    """loss(z) for z (R, D) rows already L2-normalised; mode 0: NT-Xent (R = 2N), mode 1: SupCon-fake (R = 3N)."""

    @staticmethod
    def forward(ctx, z, N, mode, temperature):
        z = z.contiguous()
        loss, lse = ops.contrast_fwd(z, N, mode, temperature)
        ctx.save_for_backward(z, lse)
        ctx.cfg = (N, mode, temperature)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        z, lse = ctx.saved_tensors
        N, mode, temperature = ctx.cfg
        dz = ops.contrast_bwd(z, lse, N, mode, temperature, grad_scale=g.reshape(1).contiguous().float())
        return dz, None, None, None


def nt_xent(out1, out2, temperature=0.1, distributed=False, normalize=False):
    """Compute the NT-Xent loss (same signature and semantics as training/criterion.py:24-45)."""
    assert out1.size(0) == out2.size(0)
    if normalize:
        out1 = _RowNormalize.apply(out1)
        out2 = _RowNormalize.apply(out2)
    if distributed:
        out1 = torch.cat(GatherLayer.apply(out1), dim=0)
        out2 = torch.cat(GatherLayer.apply(out2), dim=0)
    N = out1.size(0)
    return _Contrast.apply(torch.cat([out1, out2], dim=0), N, MODE_NT_XENT, temperature)
