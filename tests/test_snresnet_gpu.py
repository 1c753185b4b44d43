"""Scope row N4 on the device: D_SNResNet18 (models/gan/snresnet.py) under the ContraD discriminator loss and the
simclr_only training mode (training/gan/simclr_only.py), against reference-generated goldens."""
import argparse

import numpy as np
import pytest
import torch

from contrad_amd.models.gan import get_architecture
from contrad_amd.training.gan import setup
from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3
FLIP_TOL = float(__import__('os').environ.get('CONTRAD_FLIP_TOL', '1.5e-2'))     # ReLU slope flips (1 <-> 0): observed worst 7.2e-3; see tests/test_sndcgan_gpu.py
DEV = 'cuda'


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _build(seed):
    G, D = get_architecture('snresnet18', (32, 32, 3))
    shapes = O.snresnet18_param_shapes()
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes and list(D.state_dict()) == list(shapes)
    D.load_state_dict(O.det_fill(shapes, seed=seed, weight_std=0.05))
    return D.to(DEV).train()


def test_snresnet18_forward_against_reference(golden):
    g = golden('snresnet')
    D = _build(int(g['wseed']))
    with torch.no_grad():
        logit, aux = D(torch.from_numpy(g['aug']).to(DEV), sg_linear=True, projection=True, projection2=True,
                       penultimate=True)
    assert rel(logit, g['logit']) < TOL and rel(aux['penultimate'], g['penultimate']) < TOL
    assert rel(aux['projection'], g['projection']) < TOL and rel(aux['projection2'], g['projection2']) < TOL


@pytest.mark.parametrize('mode', ['contrad', 'simclr_only'])
def test_snresnet18_discriminator_losses_against_reference(golden, mode):
    g = golden('snresnet')
    N = int(g['N'])
    D = _build(int(g['wseed']))
    aug = torch.from_numpy(g['aug']).to(DEV)
    P = setup(argparse.Namespace(mode=mode, aug='simclr', temp=0.1, lbd_a=1.0, distributed=False))
    assert P.filename == ('contrad_simclr_L1.0_T0.1' if mode == 'contrad' else 'simclr_only_simclr_T0.1')
    P.augment_fn = lambda t: aug[:t.size(0)]
    d_loss, a = P.train_fn['D'](P, D, {'loss': 'nonsat'}, torch.from_numpy(g['x']).to(DEV),
                                torch.from_numpy(g['fake']).to(DEV))
    (d_loss + a['penalty']).backward()
    if mode == 'contrad':
        assert abs(d_loss.item() - float(g['contrad_loss'])) < TOL * abs(float(g['contrad_loss']))
        assert abs(a['penalty'].item() - float(g['gan_loss'])) < TOL * float(g['gan_loss'])
        for k in ('conv1.weight_u', 'layer4.1.conv2.weight_u', 'linear.l1.weight_u'):      # the in-place power iteration
            assert rel(D.state_dict()[k], g['after/' + k]) < TOL, k
    else:
        assert abs(d_loss.item() - float(g['simclr_only_loss'])) < TOL * abs(float(g['simclr_only_loss']))
        assert a['penalty'].item() == 0.0 and a['d_real'].item() == 0.0
    grads = {k: p.grad for k, p in D.named_parameters()}
    pre = mode + '/gradnorm/'
    for k in g.files:
        if k.startswith(pre):
            name = k[len(pre):]
            ref = float(g[k])
            got = 0.0 if grads[name] is None else grads[name].norm().item()
            if ref < 1e-9:
                assert got < 1e-7, name
            else:
                assert abs(got - ref) < 2e-2 * ref, (name, got, ref)
        elif k.startswith(mode + '/grad/'):
            name = k[len(mode + '/grad/'):]
            if float(g[pre + name]) >= 1e-9:
                assert l2(grads[name], g[k]) < FLIP_TOL, (name, l2(grads[name], g[k]))


def test_simclr_only_generator_loss_and_training_loop_run(tmp_path):
    """simclr_only.loss_G_fn variants + the CLI with --mode=simclr_only on snresnet18."""
    import os
    from contrad_amd.train_gan import main
    from contrad_amd.training.gan import simclr_only
    D = _build(3)
    d = torch.randn(6, 3, 32, 32, device=DEV).sigmoid()
    P = argparse.Namespace(augment_fn=lambda t: t, temp=0.1, distributed=False)
    with torch.no_grad():
        logits = D(d)
    want = {'nonsat': torch.nn.functional.softplus(-logits).mean(), 'lsgan': 0.5 * ((logits - 1.0) ** 2).mean(),
            'hinge': -logits.mean()}
    for kind, w in want.items():
        D2 = _build(3)
        got = simclr_only.loss_G_fn(P, D2, {'loss': kind}, None, d)
        assert abs(got.item() - w.item()) < 1e-4 * max(1.0, abs(w.item())), kind
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin = os.path.join(root, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    logdir = str(tmp_path / 'so')
    main([gin, 'snresnet18', '--mode=simclr_only', '--aug=simclr', '--synthetic', '--max_steps', '2', '--print_every', '1',
          '--evaluate_every', '2', '--logdir', logdir])
    log = open(os.path.join(logdir, 'log.txt')).read()
    assert '[Steps       2]' in log and 'nan' not in log.lower()


def test_snresnet18_finetuning_flag_freezes_the_trunk(golden):
    """forward(..., finetuning=True) (base.py:111-119): features under eval + no_grad (no power iteration, no gradient
    in the trunk); the heads keep training."""
    g = golden('snresnet')
    D = _build(int(g['wseed']))
    u_trunk, u_head = D.conv1.weight_u.clone(), D.linear.l1.weight_u.clone()
    x = torch.from_numpy(g['aug'])[:4].to(DEV)
    out, aux = D(x, finetuning=True, projection=True)
    (out.sum() + aux['projection'].sum()).backward()
    assert torch.equal(D.conv1.weight_u, u_trunk) and not torch.equal(D.linear.l1.weight_u, u_head)
    assert D.conv1.weight_orig.grad is None or D.conv1.weight_orig.grad.abs().max().item() == 0.0
    assert D.linear.l1.weight_orig.grad.abs().max().item() > 0 and D.projection[0].weight_orig.grad.abs().max().item() > 0
