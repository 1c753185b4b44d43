"""Inputs too large to commit are regenerated from CPU-generator seeds on both sides of a golden comparison
(tests/golden/make_golden.py uses the same rule); the fixture carries a float64 checksum."""
import torch
import torch.nn.functional as F


def seeded_images(n, size, seed):
    """Distinct smooth synthetic 'photos' in [0,1]: per-sample random 6x6 colour field, bilinearly upsampled, plus
    10 % pixel noise."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 3, 6, 6, generator=g)
    img = F.interpolate(base, size=(size, size), mode='bilinear', align_corners=False)
    return (0.9 * img + 0.1 * torch.rand(n, 3, size, size, generator=g)).clamp_(0, 1)
