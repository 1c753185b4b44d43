"""Counterpart of the reference's ``models.gan`` package (models/gan/__init__.py:2-31)."""


def get_architecture(architecture, image_size, P=None):
    """Same contract as the reference: returns (generator, discriminator) for the given architecture name."""
    if architecture == 'sndcgan':
        from .sndcgan import G_SNDCGAN, D_SNDCGAN
        generator = G_SNDCGAN(image_size=image_size)
        discriminator = D_SNDCGAN(image_size=image_size, mlp_linear=True, d_hidden=512)
    elif architecture == 'snresnet18':
        from .sndcgan import G_SNDCGAN
        from .snresnet import D_SNResNet18
        generator = G_SNDCGAN(image_size=image_size)
        discriminator = D_SNResNet18(mlp_linear=True, d_hidden=1024)
    elif architecture in ('stylegan2', 'stylegan2_512'):
        from .stylegan2 import get_stylegan2
        generator, discriminator = get_stylegan2(architecture, image_size)
    else:
        raise NotImplementedError(architecture)
    return generator, discriminator
