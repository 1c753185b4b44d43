"""SimCLR-only discriminator training -- counterpart of training/gan/simclr_only.py (scope row N4): the
discriminator learns from the NT-Xent loss on two augmented views of the reals alone; the generator still trains
against D's (stop-gradient-free) logits."""
import torch

from ..criterion import _RowNormalize, nt_xent
from .contrad import _GanGLoss


def loss_D_fn(P, D, options, images, gen_images):
    """simclr_only.py:9-21.  (``projection(D, x)`` of models/gan/base.py:73-76 = aux['projection'] + d.mean() * 0.)"""
    real_images = torch.cat([images, images], dim=0)
    d, aux = D(P.augment_fn(real_images), projection=True)
    views = _RowNormalize.apply(aux['projection'] + d.mean() * 0)
    view1, view2 = torch.chunk(views, 2, dim=0)
    simclr_loss = nt_xent(view1, view2, temperature=P.temp, distributed=P.distributed)
    return simclr_loss, {
        "penalty": 0. * simclr_loss,
        "d_real": 0. * simclr_loss,
        "d_gen": 0. * simclr_loss,
    }


def loss_G_fn(P, D, options, images, gen_images):
    """simclr_only.py:24-33: nonsat / lsgan / (anything else) -d.mean()."""
    d_gen = D(P.augment_fn(gen_images))
    kind = options['loss'] if options['loss'] in ('nonsat', 'lsgan') else 'wgan'
    return _GanGLoss.apply(d_gen, kind)
