"""StyleGAN2 generator (models/gan/stylegan2/generator.py:146-291) -- placeholder until the forward lands."""
import torch.nn as nn


class Generator(nn.Module):
    def __init__(self, size, style_dim=512, n_mlp=8, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01,
                 small32=False):
        super().__init__()
        raise NotImplementedError('StyleGAN2 generator forward: scope row G0 (StyleGAN2), not built yet')
