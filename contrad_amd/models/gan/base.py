"""Parameter containers with the reference's state-dict names, and the discriminator head contract
(models/gan/base.py:79-164) shared by the MI355X-native discriminators."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SNParams(nn.Module):
    """Holds what ``spectral_norm(nn.Conv2d/nn.Linear)`` holds -- ``bias``, ``weight_orig`` (parameters),
    ``weight_u``, ``weight_v`` (buffers) -- under the same names and in the same order, so checkpoints written by
    the reference load unchanged.  No forward: the owning network consumes the tensors in one
    batched HIP weight-prep launch."""

    def __init__(self, weight_shape, init_std=0.02):
        super().__init__()
        out = weight_shape[0]
        inn = int(math.prod(weight_shape[1:]))
        self.bias = nn.Parameter(torch.zeros(out))
        self.weight_orig = nn.Parameter(torch.empty(*weight_shape))
        self.register_buffer('weight_u', torch.empty(out))
        self.register_buffer('weight_v', torch.empty(inn))
        self.init_std = init_std
        self.reset_parameters()

    def reset_parameters(self):
        # D_SNDCGAN.reset_parameters (sndcgan.py:130-148): N(0, 0.02) weights, zero bias; u/v as in
        # torch.nn.utils.spectral_norm: normalised N(0,1) vectors.
        with torch.no_grad():
            self.weight_orig.normal_(0.0, self.init_std)
            self.bias.zero_()
            self.weight_u.copy_(F.normalize(torch.randn(self.weight_u.shape), dim=0, eps=1e-12))
            self.weight_v.copy_(F.normalize(torch.randn(self.weight_v.shape), dim=0, eps=1e-12))

    def extra_repr(self):
        return 'weight=%s (spectral norm)' % (tuple(self.weight_orig.shape),)


class PlainParams(nn.Module):
    """``weight`` / ``bias`` holder (heads of the StyleGAN2 discriminator, which carry no spectral norm)."""

    def __init__(self, weight_shape, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.zeros(weight_shape[0])) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Linear default init (kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
        fan_in = int(math.prod(self.weight.shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)


class _Act(nn.Module):
    """Parameter-free placeholder keeping the reference's Sequential indices (convs at 0,2,4,...)."""

    def __init__(self, slope):
        super().__init__()
        self.slope = slope

    def extra_repr(self):
        return 'LeakyReLU(%g), fused into the producing kernel' % self.slope


class TinyHead(nn.Module):
    """TinyDiscriminator's parameters (base.py:14-35): l1 (d_hidden x n_features), l2 (1 x d_hidden)."""

    def __init__(self, n_features, d_hidden, spectral):
        super().__init__()
        mk = SNParams if spectral else PlainParams
        self.l1 = mk((d_hidden, n_features))
        self.l2 = mk((1, d_hidden))


def make_projection(n_features, d_hidden, d_project, spectral):
    """nn.Sequential(Linear, LeakyReLU(0.1), Linear) parameter layout (base.py:92-101): entries '0' and '2'."""
    mk = SNParams if spectral else PlainParams
    return nn.Sequential(mk((d_hidden, n_features)), _Act(0.1), mk((d_project, d_hidden)))


class BaseDiscriminator(nn.Module):
    """Forward-flag semantics of the reference's BaseDiscriminator.forward (base.py:107-150):
    returns ``output`` or ``(output, aux)`` with aux keys 'penultimate' | 'projection' | 'projection2'."""

    d_penul = None

    def _run(self, inputs, sg_linear, finetuning, want_features):
        raise NotImplementedError

    def forward(self, inputs, y=None, penultimate=False, projection=False, projection2=False,
                finetuning=False, sg_linear=False):
        if y is not None:
            raise NotImplementedError('class-conditional heads (n_classes > 1) are not used by get_architecture')
        if getattr(self, '_lazy_projections', False):       # (networks whose heads are separate launches)
            output, project, project2, features = self._run(inputs, sg_linear, finetuning, penultimate,
                                                            want_proj=(projection or projection2))
        else:
            output, project, project2, features = self._run(inputs, sg_linear, finetuning, penultimate)
        aux = {}
        if penultimate:
            aux['penultimate'] = features
        if projection:
            aux['projection'] = project
        if projection2:
            aux['projection2'] = project2
        if aux:
            return output, aux
        return output

    def reset_parameters(self, root=None):
        """Re-initialise ``root`` (default: everything) -- used by --finetune on ``D.linear`` (train_gan.py:265)."""
        root = self if root is None else root
        for m in root.modules():
            if isinstance(m, (SNParams, PlainParams)):
                m.reset_parameters()
