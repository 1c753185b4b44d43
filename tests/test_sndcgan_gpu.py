"""End-to-end GPU parity of the SNDCGAN ContraD discriminator step against the goldens captured from the
imported reference (tests/golden/sndcgan.npz) and against the oracle at a second, larger batch."""
import numpy as np
import pytest
import torch

from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import contrad as hip_contrad
from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3
DEV = 'cuda'


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def grad_close(got, ref, tol):
    """Relative L2 error of a gradient tensor."""
    got, ref = torch.as_tensor(got).double().cpu().reshape(-1), torch.as_tensor(ref).double().cpu().reshape(-1)
    l2 = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    return l2 < tol, l2


# Element-wise gradients vs the REFERENCE-generated goldens: relative L2 per tensor below FLIP_TOL = 1e-3 (the
# north_star tolerance; observed worst 3.5e-4 here, < 2e-4 for the StyleGAN2 fixtures).  What it has to absorb:
# leaky_relu's derivative is discontinuous at 0, and with ~10^6 activations per step a pre-activation of ~1e-8 (fp32
# summation-order noise) lands on the other side of 0 than in the reference's BLAS now and then, changing that unit's
# slope 1 <-> 0.1 (tests/debug/debug_dstep.py pin-points the single flipped unit of this fixture; it shows up as the
# 3.5e-4 of main.4.bias).  That is a property of the maths, not of a kernel: any two fp32 conv implementations
# disagree the same way.  The ReLU networks (SNDCGAN's generator step, SNResNet18) flip between slope 1 and 0 and are
# compared at 1e-2 in their files (observed 5-6e-3).  The strict element-wise check independent of flips is
# test_full_step_on_the_same_linear_region below, where the oracle is evaluated with the activation sign pattern of
# the run under test; forward quantities, gradient NORMS and the u/v/Adam state are 1e-3 everywhere.
# (CONTRAD_FLIP_TOL overrides the value for margin probes.)
FLIP_TOL = float(__import__('os').environ.get('CONTRAD_FLIP_TOL', '1e-3'))


class _P(object):
    def __init__(self, aug):
        self.augment_fn = aug
        self.temp, self.lbd_a, self.distributed = 0.1, 1.0, False


def build(seed_d=1234, seed_g=4321):
    G, D = get_architecture('sndcgan', (32, 32, 3))
    sd = O.det_fill(O.sndcgan_d_param_shapes(), seed=seed_d)
    D.load_state_dict(sd)
    gsd = dict(G.state_dict())
    gsd.update(O.det_fill(O.sndcgan_g_param_shapes(), seed=seed_g))
    G.load_state_dict(gsd)
    return G.to(DEV).train(), D.to(DEV).train()


def test_generator_forward_matches_reference(golden):
    g = golden('sndcgan')
    G, _ = build()
    with torch.no_grad():
        fake = G(torch.from_numpy(g['z']).to(DEV))
    assert rel(fake, g['fake']) < TOL
    sd = G.state_dict()
    for k in ('norm_init.running_mean', 'main.1.running_mean', 'main.1.running_var', 'main.7.running_var'):
        assert rel(sd[k][:256], g['gbuf/' + k]) < TOL, k
    assert int(sd['main.1.num_batches_tracked']) == 1


def test_discriminator_step_matches_reference(golden):
    g = golden('sndcgan')
    N = int(g['N'])
    _, D = build()
    aug = torch.from_numpy(g['aug']).to(DEV)
    x = torch.from_numpy(g['x']).to(DEV)
    fake = torch.from_numpy(g['fake']).to(DEV)

    logit, aux = D(aug, sg_linear=True, projection=True, projection2=True, penultimate=True)
    assert rel(logit, g['logit']) < TOL
    assert rel(aux['projection'], g['projection']) < TOL and rel(aux['projection2'], g['projection2']) < TOL
    assert rel(aux['penultimate'][:, :64], g['penultimate_head']) < TOL
    assert rel(aux['penultimate'].sum(1), g['penultimate_sum']) < TOL

    # fresh model: the forward above already advanced the power iteration once
    _, D = build()
    P = _P(lambda t: aug)
    d_loss, a = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x, fake)
    assert abs(d_loss.item() - float(g['contrad_loss'])) < TOL * abs(float(g['contrad_loss']))
    assert abs(a['penalty'].item() - float(g['gan_loss'])) < TOL * abs(float(g['gan_loss']))
    assert abs(a['d_real'].item() - float(g['d_real'])) < TOL and abs(a['d_gen'].item() - float(g['d_gen'])) < TOL
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt.zero_grad()
    (d_loss + a['penalty']).backward()
    grads = {k: p.grad for k, p in D.named_parameters()}
    worst = 0.0
    for k in g.files:
        if k.startswith('gradnorm/'):
            name = k[len('gradnorm/'):]
            e = abs(grads[name].norm().item() - float(g[k])) / max(float(g[k]), 1e-30)
            worst = max(worst, e)
            assert e < TOL, (name, e)
        elif k.startswith('grad/'):
            ok, info = grad_close(grads[k[5:]], g[k], FLIP_TOL)
            assert ok, (k, info)
        elif k.startswith('gradhead/'):
            name = k[len('gradhead/'):]
            ref = torch.from_numpy(g[k])
            got = grads[name].reshape(-1)[:512].cpu()
            assert (got - ref).abs().max().item() < TOL * float(g['gradnorm/' + name]) , name
    sd = D.state_dict()
    for k in g.files:
        if k.startswith('after/'):
            assert rel(sd[k[6:]], g[k]) < TOL, k
        elif k.startswith('afterhead/'):
            assert rel(sd[k[10:]][:512], g[k]) < TOL, k
    opt.step()
    for k in g.files:
        if k.startswith('adamhead/'):
            name = k[len('adamhead/'):]
            p = dict(D.named_parameters())[name]
            assert rel(p.detach().reshape(-1)[:256], g[k]) < TOL, name
            assert abs(p.detach().double().sum().item() - float(g['adamsum/' + name])) < 1e-3 * max(1.0, abs(float(g['adamsum/' + name])))


def test_full_step_on_the_same_linear_region():
    """G fwd -> augment (host-sampled params, same RNG stream) -> D -> losses -> backward at N = 16, against the
    oracle evaluated on the SAME leaky-relu linear region (activation signs taken from the HIP forward): every
    parameter gradient must agree element-wise within 1e-3 of its max, losses within 1e-3."""
    N = 16
    G, D = build(seed_d=77, seed_g=78)
    D._record_activations = True
    g = torch.Generator().manual_seed(5)
    x = torch.rand(N, 3, 32, 32, generator=g)
    z = torch.rand(N, 128, generator=g) * 2 - 1
    from contrad_amd.augment import SimCLRAugment
    aug = SimCLRAugment(scale=(0.2, 1.0))
    torch.manual_seed(21); np.random.seed(21)
    with torch.no_grad():
        fake = G(z.to(DEV))
    P = _P(aug)
    d_loss, a = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x.to(DEV), fake)
    (d_loss + a['penalty']).backward()
    acts, hidden = D._last_activations
    masks = [(t > 0).permute(0, 3, 1, 2).cpu() for t in acts]
    hm = (hidden.view(3 * N, -1) > 0).cpu()
    hmasks = (hm[:, :512], hm[:, 512:1024], hm[:, 1024:])

    osd = O.det_fill(O.sndcgan_d_param_shapes(), seed=77)
    ogsd = O.det_fill(O.sndcgan_g_param_shapes(), seed=78)
    for k in osd:
        if k.endswith('weight_orig') or k.endswith('bias'):
            osd[k].requires_grad_()
    with torch.no_grad():
        ofake = O.sndcgan_g_forward(ogsd, z)
    assert rel(fake, ofake) < TOL
    torch.manual_seed(21); np.random.seed(21)           # same RNG stream -> same augmentation parameters
    p = O.sample_simclr_params(3 * N, 32, 32, O.SIMCLR_CIFAR)
    oaug = O.simclr_apply(torch.cat([x, x, ofake]), p)
    fwd = lambda t: O.sndcgan_d_forward(osd, t, sg_linear=True, act_masks=masks, hidden_masks=hmasks)[:3]
    closs, gloss, _, _ = O.contrad_loss_d(fwd, oaug, N)
    (closs + gloss).backward()
    # the imposed region is (almost everywhere) the oracle's own: the loss is unchanged
    closs0, gloss0, _, _ = O.contrad_loss_d(
        lambda t: O.sndcgan_d_forward(O.det_fill(O.sndcgan_d_param_shapes(), seed=77), t, sg_linear=True)[:3], oaug, N)
    assert abs(closs.item() - closs0.item()) < 1e-5 and abs(gloss.item() - gloss0.item()) < 1e-5
    assert abs(d_loss.item() - closs.item()) < TOL * abs(closs.item())
    assert abs(a['penalty'].item() - gloss.item()) < TOL * abs(gloss.item())
    for k, prm in D.named_parameters():
        ref = osd[k].grad
        e = (prm.grad.cpu() - ref).abs().max().item() / ref.abs().max().clamp_min(1e-30).item()
        assert e < TOL, (k, e)


def test_eval_mode_and_input_gradient():
    """eval(): no power iteration (u, v untouched); d(loss)/d(images) via the RGB dgrad kernel vs the oracle."""
    _, D = build()
    D.eval()
    sd0 = {k: v.clone() for k, v in D.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    x = torch.rand(6, 3, 32, 32, generator=g)
    xd = x.to(DEV).requires_grad_()
    out = D(xd)
    out.sum().backward()
    for k, v in D.state_dict().items():
        if k.endswith('weight_u') or k.endswith('weight_v'):
            assert torch.equal(v, sd0[k]), k
    osd = O.det_fill(O.sndcgan_d_param_shapes(), seed=1234)
    xr = x.clone().requires_grad_()
    o = O.sndcgan_d_forward(osd, xr, sg_linear=False, training=False)[0]
    o.sum().backward()
    assert rel(out, o.detach()) < TOL
    assert rel(xd.grad, xr.grad) < TOL


def test_finetuning_flag_matches_the_reference_semantics():
    """forward(..., finetuning=True) (base.py:111-119): features from the network in eval mode under no_grad -- trunk u / v
    untouched, zero trunk gradients -- while the heads run in train mode (their power iteration advances) and train;
    values against the oracle evaluated the same way."""
    _, D = build()
    sd0 = {k: v.clone() for k, v in D.state_dict().items()}
    x = torch.rand(6, 3, 32, 32, generator=torch.Generator().manual_seed(8))
    out, aux = D(x.to(DEV), finetuning=True, projection=True)
    (out.sum() + aux['projection'].pow(2).sum()).backward()
    sd1 = D.state_dict()
    for k in sd0:
        if k.endswith(('weight_u', 'weight_v')) and sd0[k].numel() > 1:      # (a 1-vector u is +-1 before and after)
            moved = not torch.equal(sd0[k], sd1[k])
            assert moved == (not k.startswith('main.')), k
    for k, p in D.named_parameters():
        if k.startswith('main.'):
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
    assert D.linear.l1.weight_orig.grad.abs().max().item() > 0
    osd = O.det_fill(O.sndcgan_d_param_shapes(), seed=1234)
    with torch.no_grad():
        feats = O.sndcgan_d_features(osd, x, training=False)
    o, proj, _ = O.d_heads(osd, feats, sg_linear=False, training=True)
    assert rel(out, o) < TOL and rel(aux['projection'], proj) < TOL


def test_overlapped_gradient_exchange_single_rank_rccl():
    """The in-backward gradient exchange on a 1-rank RCCL group: every async all-reduce / wait is exercised and the
    gradients must be bitwise those of the plain backward (SUM over one rank)."""
    import torch.distributed as dist
    from contrad_amd.engine import OverlappedGradReducer
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1,
                                device_id=torch.device('cuda', 0))
    try:
        g = torch.Generator().manual_seed(8)
        x = torch.rand(12, 3, 32, 32, generator=g).to(DEV)

        def grads(overlap):
            _, D = build()
            if overlap:
                comm = OverlappedGradReducer()
                comm.force = True
                D.enable_grad_overlap(comm)
            o, a = D(x, sg_linear=True, projection=True, projection2=True)
            (o.sum() + a['projection'].pow(2).sum() + a['projection2'].sum()).backward()
            return {k: p.grad.clone() for k, p in D.named_parameters()}

        ga, gb = grads(False), grads(True)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
    finally:
        if own:
            dist.destroy_process_group()


def test_full_size_step_is_bitwise_deterministic():
    """BASELINE size (global batch 512 -> 1536 images through D): every reduction on the path (split-K slabs, bias /
    BatchNorm column sums, spectral-norm partials, contrastive column splits) has a fixed order, so two runs from the
    same state and the same samples give bit-identical losses, gradients, power-iteration vectors and Adam updates --
    and the step has the properties the loss promises (finite, GAN loss of an untrained D = 2 log 2)."""
    import argparse
    import copy
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import d_step, set_grad
    from contrad_amd.training.gan import setup
    import os
    from contrad_amd import config
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b512.gin')])
    N = 512
    G0, D0 = build()
    aug = get_augment(mode='simclr').to(DEV)
    torch.manual_seed(3); np.random.seed(3)
    images = torch.rand(N, 3, 32, 32, device=DEV)
    z = torch.empty(N, 128).uniform_(-1, 1)
    params = aug.sample(3 * N, 32, 32)
    outs = []
    for _ in range(2):
        G, D = copy.deepcopy(G0), copy.deepcopy(D0)
        G.sample_latent = lambda n: z.to(DEV)
        a = copy.deepcopy(aug)
        a.sample = lambda B, h, w: params
        P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=False))
        P.augment_fn = a
        opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
        set_grad(G, False); set_grad(D, True)
        d_loss, aux = d_step(P, G, D, opt, {'loss': 'nonsat', 'batch_size': N}, images, None)
        torch.cuda.synchronize()
        outs.append((d_loss.detach().clone(), aux['penalty'].detach().clone(),
                     [p.grad.detach().clone() for p in D.parameters()],
                     [p.detach().clone() for p in D.parameters()],
                     [b.detach().clone() for b in D.buffers()]))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in (2, 3, 4):
        assert len(a[k]) == len(b[k]) and all(torch.equal(x, y) for x, y in zip(a[k], b[k]))
    assert torch.isfinite(a[0]).item() and all(torch.isfinite(g).all().item() for g in a[2])
    assert abs(a[1].item() - 2 * np.log(2)) < 0.05        # sigmoid(0) on both sides: softplus(0) * 2


def test_generator_eval_mode_forward_matches_reference(golden):
    """G_SNDCGAN.eval() (running statistics, sndcgan.py:41-52 / train_gan.py:181: what a round-tripped gen.pt is sampled
    with) against the imported reference; eval mode leaves every buffer untouched."""
    g = golden('sndcgan_eval')
    G, _ = build()
    gsd = O.det_fill(O.sndcgan_g_param_shapes(), seed=int(g['wseed']))
    full = dict(G.state_dict())
    full.update({k: v.clone() for k, v in gsd.items()})
    G.load_state_dict(full)
    G = G.to(DEV).eval()
    before = {k: v.clone() for k, v in G.state_dict().items()}
    with torch.no_grad():
        img = G(torch.from_numpy(g['z']).to(DEV))
    assert rel(img, g['img']) < TOL
    for k, v in G.state_dict().items():
        assert torch.equal(v, before[k]), k
    G.train()
    with torch.no_grad():
        img_t = G(torch.from_numpy(g['z']).to(DEV))
    assert rel(img_t, g['img']) > 1e-2                       # (batch statistics give a different image)
