"""GPU parity: fused NT-Xent / SupCon kernels vs the oracle and the reference-generated goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from contrad_amd import ops
from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def run_hip(u1, u2, N, temp):
    dev = torch.device('cuda')
    u1d, u2d = u1.to(dev), u2.to(dev)
    z1, inv1 = ops.l2norm_fwd(u1d[:2 * N])
    z2, inv2 = ops.l2norm_fwd(u2d)
    l1, lse1 = ops.contrast_fwd(z1, N, 0, temp)
    l2, lse2 = ops.contrast_fwd(z2, N, 1, temp)
    dz1 = ops.contrast_bwd(z1, lse1, N, 0, temp)
    dz2 = ops.contrast_bwd(z2, lse2, N, 1, temp)
    g1 = torch.zeros_like(u1d)
    ops.l2norm_bwd(dz1, z1, inv1, out=g1[:2 * N])
    g2 = ops.l2norm_bwd(dz2, z2, inv2)
    return l1.cpu(), l2.cpu(), g1.cpu(), g2.cpu()


@pytest.mark.parametrize('tag', ['small', 'mid', 'hot'])
def test_against_golden(golden, tag):
    g = golden('losses')
    N, temp = int(g[tag + '_N']), float(g[tag + '_temp'])
    u1, u2 = torch.from_numpy(g[tag + '_u1']), torch.from_numpy(g[tag + '_u2'])
    l1, l2, g1, g2 = run_hip(u1, u2, N, temp)
    assert abs(l1.item() - float(g[tag + '_nt_xent'])) < TOL * abs(float(g[tag + '_nt_xent']))
    assert abs(l2.item() - float(g[tag + '_supcon'])) < TOL * abs(float(g[tag + '_supcon']))
    assert rel(g1, torch.from_numpy(g[tag + '_g1'])) < TOL
    assert rel(g2, torch.from_numpy(g[tag + '_g2'])) < TOL


@pytest.mark.parametrize('N,D,temp', [(2, 128, 0.1), (33, 128, 0.1), (64, 128, 0.07), (100, 96, 0.2), (512, 128, 0.1)])
def test_against_oracle(N, D, temp):
    g = torch.Generator().manual_seed(N)
    u1 = torch.randn(3 * N, D, generator=g)
    u2 = torch.randn(3 * N, D, generator=g)
    o1, o2 = u1.clone().requires_grad_(), u2.clone().requires_grad_()
    v, r = F.normalize(o1), F.normalize(o2)
    r1 = O.nt_xent(v[:N], v[N:2 * N], temp)
    r2 = O.supcon_fake(r[:N], r[N:2 * N], r[2 * N:], temp)
    (r1 + r2).backward()
    l1, l2, g1, g2 = run_hip(u1, u2, N, temp)
    assert abs(l1.item() - r1.item()) < TOL * abs(r1.item())
    assert abs(l2.item() - r2.item()) < TOL * abs(r2.item())
    assert rel(g1, o1.grad) < TOL
    assert rel(g2, o2.grad) < TOL


def test_grad_scale_and_permutation_invariance():
    """Size-independent properties at the BASELINE size: upstream-gradient scaling is exact for powers of
    two, and NT-Xent is invariant under swapping the two views."""
    N, D = 512, 128
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(9)
    z = F.normalize(torch.randn(2 * N, D, device=dev, generator=g))
    l, lse = ops.contrast_fwd(z, N, 0, 0.1)
    dz = ops.contrast_bwd(z, lse, N, 0, 0.1)
    dz4 = ops.contrast_bwd(z, lse, N, 0, 0.1, grad_scale=torch.full((1,), 4.0, device=dev))
    assert torch.equal(dz4, dz * 4.0)
    zs = torch.cat([z[N:], z[:N]]).contiguous()
    ls, _ = ops.contrast_fwd(zs, N, 0, 0.1)
    assert abs(ls.item() - l.item()) < 1e-5 * abs(l.item())
