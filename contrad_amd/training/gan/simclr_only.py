"""SimCLR-only discriminator training -- counterpart of training/gan/simclr_only.py (scope row N4).

The discriminator learns from the NT-Xent loss on two augmented views of the reals alone (no GAN term: the three
auxiliary entries of the return contract are zeros tied to the loss); the generator still trains against D's logits.
Runs on the same HIP pieces as the ContraD mode: fused augmentation, the discriminator node, the row-normalise and
contrastive kernels of ``training/criterion.py``."""
import torch

from ..criterion import _RowNormalize, nt_xent
from .contrad import _GanGLoss

_AUX_KEYS = ("penalty", "d_real", "d_gen")


def loss_D_fn(P, D, options, images, gen_images):
    """simclr_only.py:9-21.  ``projection(D, x)`` of models/gan/base.py:73-76 is aux['projection'] + d.mean() * 0."""
    two_views = P.augment_fn(torch.cat([images, images], dim=0))
    logits, aux = D(two_views, projection=True)
    z = _RowNormalize.apply(aux['projection'] + logits.mean() * 0)
    n = images.size(0)
    loss = nt_xent(z[:n], z[n:], temperature=P.temp, distributed=P.distributed)
    zero = 0. * loss
    return loss, {k: zero for k in _AUX_KEYS}


def loss_G_fn(P, D, options, images, gen_images):
    """simclr_only.py:24-33: 'nonsat' -> softplus(-d), 'lsgan' -> 0.5 (d - 1)^2, anything else -> -d (means)."""
    kind = options['loss'] if options['loss'] in ('nonsat', 'lsgan') else 'wgan'
    return _GanGLoss.apply(D(P.augment_fn(gen_images)), kind)
