"""Counterpart of the reference's ``models.gan`` package (models/gan/__init__.py:2-31)."""


def get_architecture(architecture, image_size, P=None):
    """Same contract as the reference: returns (generator, discriminator) for the given architecture name."""
    if architecture == 'sndcgan':
        from .sndcgan import G_SNDCGAN, D_SNDCGAN
        generator = G_SNDCGAN(image_size=image_size)
        discriminator = D_SNDCGAN(image_size=image_size, mlp_linear=True, d_hidden=512)
    elif architecture in ('stylegan2', 'stylegan2_512'):
        from .stylegan2 import get_stylegan2
        generator, discriminator = get_stylegan2(architecture, image_size)
    else:
        # 'snresnet18' is outside the hot-path scope
        raise NotImplementedError(architecture)
    return generator, discriminator
