"""StyleGAN2 generator / discriminator on the HIP kernel library -- counterpart of models/gan/stylegan2/."""


def get_stylegan2(architecture, image_size):
    """The two StyleGAN2 branches of get_architecture (models/gan/__init__.py:14-27)."""
    from .discriminator import ResidualDiscriminatorP
    from .generator import Generator
    resolution = image_size[0]
    if architecture == 'stylegan2':
        generator = Generator(size=resolution, n_mlp=8, small32=True)
        discriminator = ResidualDiscriminatorP(size=resolution, small32=True, mlp_linear=True, d_hidden=512)
    elif architecture == 'stylegan2_512':
        generator = Generator(size=resolution, n_mlp=8, channel_multiplier=1.0)
        discriminator = ResidualDiscriminatorP(size=resolution, channel_multiplier=1.0, mlp_linear=True, d_hidden=512)
    else:
        raise NotImplementedError(architecture)
    return generator, discriminator
