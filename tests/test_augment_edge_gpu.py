"""HIP-side parity holes named by the round-1 review: the reference's per-stage fixtures and HSV edge cases (black
pixel -> Cmax = 0, gray -> atan2(0, 0), pure red / blue) go THROUGH THE KERNEL (both the one-block small-image path and
the two-pass large-image path), and the index part of the augmentation (flip, identity crop) is bit-exact
(BASELINE.json: "bit-exact for index ops"; SURVEY.md 8a rows A1/A2: the flip is an exact column permutation)."""
import numpy as np
import pytest
import torch

from contrad_amd import ops
from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _params(B, theta=None, flip=None, jitter=None, fc=None, fh=None, fs=None, fv=None, gray=None, blur=None):
    P = torch.zeros(B, ops.AUG_NPARAM)
    P[:, 0] = 1.0; P[:, 1] = 1.0; P[:, 4] = 1.0; P[:, 6] = 1.0; P[:, 8] = 1.0; P[:, 9] = 1.0
    if theta is not None:
        P[:, 0], P[:, 1], P[:, 2], P[:, 3] = theta[:, 0, 0], theta[:, 1, 1], theta[:, 0, 2], theta[:, 1, 2]
    for col, v in ((4, flip), (5, jitter), (6, fc), (7, fh), (8, fs), (9, fv), (10, gray), (11, blur)):
        if v is not None:
            P[:, col] = torch.as_tensor(v, dtype=torch.float32)
    return P


def _run(x, P, contrast_first=True, has_contrast=True):
    return ops.simclr_augment(x.contiguous().to(DEV), P.to(DEV), contrast_first, has_contrast).cpu()


@pytest.mark.parametrize('size', [32, 64, 256, 512])      # 32/64: LDS-resident path; 256/512: two-pass path
def test_flip_and_identity_crop_are_bit_exact(size):
    g = torch.Generator().manual_seed(size)
    B = 6
    x = torch.rand(B, 3, size, size, generator=g)
    sign = torch.tensor([1., -1., -1., 1., -1., 1.])
    out = _run(x, _params(B, flip=sign))
    for i in range(B):
        want = x[i] if sign[i] > 0 else x[i].flip(-1)
        assert torch.equal(out[i], want), (size, i)
    # the oracle's two grid_samples: bit for bit at the CIFAR size (SURVEY.md 8a row A2's probe), round-off elsewhere
    # (ATen builds its base grid with linspace; the kernel's closed form stays exact at every power-of-two size)
    ref = O.hflip(O.resized_crop(x, torch.eye(2, 3).repeat(B, 1, 1)), sign)
    assert torch.equal(out, ref) if size == 32 else (out - ref).abs().max().item() < 1e-4 * size / 32


@pytest.mark.parametrize('tag', ['c10a', 'c10b'])
def test_crop_and_flip_stage_fixtures_through_the_kernel(golden, tag):
    g = golden('augment')
    x = torch.from_numpy(g[tag + '_x'])
    B = x.shape[0]
    theta = torch.from_numpy(g[tag + '_p_theta'])
    crop = _run(x, _params(B, theta=theta))
    assert (crop - torch.from_numpy(g[tag + '_stage_crop'])).abs().max().item() < 1e-5
    flip = _run(x, _params(B, theta=theta, flip=g[tag + '_p_flip_sign']))
    assert (flip - torch.from_numpy(g[tag + '_stage_flip'])).abs().max().item() < 1e-5
    # flip(crop(x)) is the exact column permutation of crop(x): bit-exact between the two kernel runs
    sign = torch.from_numpy(g[tag + '_p_flip_sign'])
    for i in range(B):
        assert torch.equal(flip[i], crop[i] if sign[i] > 0 else crop[i].flip(-1))


@pytest.mark.parametrize('rep', [1, 8])               # rep 8: 256x256 images -> the two-pass (statistics + apply) kernels
@pytest.mark.parametrize('contrast_first', [True, False])
def test_hsv_edge_case_fixture_through_the_kernel(golden, rep, contrast_first):
    """RandomHSVFunction.forward on black / gray / pure-red / pure-blue pixels (Cmax = 0, atan2(0,0), the 255/360 hue
    quirk) -- reference output ``hsv_adjusted``."""
    g = golden('augment')
    x = torch.from_numpy(g['hsv_x'])
    want = torch.from_numpy(g['hsv_adjusted'])
    if rep > 1:
        x = x.repeat_interleave(rep, 2).repeat_interleave(rep, 3)
        want = want.repeat_interleave(rep, 2).repeat_interleave(rep, 3)
    B = x.shape[0]
    P = _params(B, jitter=torch.ones(B), fh=g['hsv_fh'], fs=g['hsv_fs'], fv=g['hsv_fv'])
    out = _run(x, P, contrast_first=contrast_first, has_contrast=False)
    assert torch.isfinite(out).all()
    assert (out - want).abs().max().item() < 1e-5
    # the four special pixels of sample 0, explicitly
    for col in range(4):
        assert (out[0, :, 0, col * rep] - want[0, :, 0, col * rep]).abs().max().item() < 1e-6, col
    # un-selected samples (RandomApply mask 0) pass through untouched -- bit-exact
    P0 = _params(B, jitter=torch.zeros(B), fh=g['hsv_fh'], fs=g['hsv_fs'], fv=g['hsv_fv'])
    assert torch.equal(_run(x, P0, contrast_first=contrast_first, has_contrast=False), x)


@pytest.mark.parametrize('rep', [1, 8])
def test_contrast_and_gray_fixtures_through_the_kernel(golden, rep):
    g = golden('augment')
    x = torch.from_numpy(g['con_x'])
    con = torch.from_numpy(g['con_out'])
    gray = torch.from_numpy(g['gray_out'])
    if rep > 1:
        x, con, gray = [t.repeat_interleave(rep, 2).repeat_interleave(rep, 3) for t in (x, con, gray)]
    B = x.shape[0]
    # ColorJitterLayer with the HSV factors at identity: contrast (+clamp) followed by the rgb->hsv->rgb round trip
    out = _run(x, _params(B, jitter=torch.ones(B), fc=g['con_f']), contrast_first=True, has_contrast=True)
    one, zero = torch.ones(B), torch.zeros(B)
    want = O.adjust_hsv(con, zero, one, one)                      # pinned to the reference by hsv_roundtrip / hsv_adjusted
    assert (out - want).abs().max().item() < 1e-5
    # (the atan2 "circular" hue is not the hexagonal one: the round trip moves values by up to ~2e-2, as in the reference)
    assert (want - con).abs().max().item() > 1e-3 and (out - con).abs().max().item() < 5e-2
    # RandomColorGrayLayer
    out = _run(x, _params(B, gray=torch.ones(B)))
    assert (out - gray).abs().max().item() < 1e-6
    mixed = _run(x, _params(B, gray=torch.tensor([1., 0., 1., 0., 0., 1.])))
    for i, m in enumerate([1, 0, 1, 0, 0, 1]):
        assert torch.equal(mixed[i], out[i] if m else x[i])


def test_hip_gaussian_blur_properties_at_512():
    """A6 is 'parity unpinned' (kornia absent): property checks ON THE HIP BLUR at the real size (512^2 -> ksize 51):
    constant images are invariant (kernel sums to 1, reflect padding), the separable evaluation equals the 2-D
    correlation of the oracle, masked-out samples pass through bit-exact."""
    from contrad_amd.augment import SimCLRAugment
    B, H, sigma = 3, 512, 1.7
    radius, k1 = SimCLRAugment.blur_kernel(H, sigma)
    assert radius == 25 and abs(k1.sum().item() - 1.0) < 1e-6
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, H, H, generator=g)
    x[1] = 0.3125                                                 # a constant image
    P = _params(B, blur=torch.tensor([1., 1., 0.]))
    out = ops.gaussian_blur_masked(x.to(DEV), P.to(DEV), k1.to(DEV), radius).cpu()
    assert (out[1] - 0.3125).abs().max().item() < 1e-6
    assert torch.equal(out[2], x[2])
    want = O.gaussian_blur(x[:1], sigma)                          # 51 x 51 depthwise correlation, reflect padding
    assert (out[:1] - want).abs().max().item() < 1e-5
    # sigma at the low end of the range: the kernel degenerates to (almost) a delta
    radius, k1 = SimCLRAugment.blur_kernel(H, 0.1)
    out = ops.gaussian_blur_masked(x.to(DEV), P.to(DEV), k1.to(DEV), radius).cpu()
    assert (out[0] - x[0]).abs().max().item() < 1e-6


def test_simclr_hq_at_512_matches_oracle():
    """Config 5's augmentation at its real size: crop scale (0.08, 1), jitter 0.8/0.8/0.8/0.2, ksize-51 blur, the
    two-pass statistics path; explicit parameters drawn by the oracle in the reference's RNG order."""
    from contrad_amd.augment import SimCLRAugment
    B, H = 4, 512
    torch.manual_seed(21); np.random.seed(21)
    from sg2_inputs import seeded_images
    x = seeded_images(B, H, 77)
    p = O.sample_simclr_params(B, H, H, O.SIMCLR_HQ_AFHQ)
    p['jitter_mask'][:] = torch.tensor([1., 1., 0., 1.])
    p['gray_mask'][:] = torch.tensor([0., 1., 0., 0.])
    p['blur_mask'][:] = torch.tensor([1., 0., 1., 1.])
    ref = O.simclr_apply(x, p)
    P = _params(B, theta=p['theta'], flip=p['flip_sign'], jitter=p['jitter_mask'], fc=p['f_contrast'], fh=p['f_h'],
                fs=p['f_s'], fv=p['f_v'], gray=p['gray_mask'], blur=p['blur_mask'])
    aug = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                        sigma_range=(0.1, 2.0))
    for cf in (True, False):
        p['contrast_first'] = cf
        ref = O.simclr_apply(x, p)
        out = aug.apply(x.to(DEV), P, cf, p['sigma']).cpu()
        assert (out - ref).abs().max().item() < 1e-4, cf
    # host sampler == oracle sampler at this size (RNG draw order incl. the blur mask and sigma)
    torch.manual_seed(5); np.random.seed(5)
    P2, cf2, s2 = aug.sample(B, H, H)
    torch.manual_seed(5); np.random.seed(5)
    q = O.sample_simclr_params(B, H, H, O.SIMCLR_HQ_AFHQ)
    assert cf2 == q['contrast_first'] and abs(s2 - q['sigma']) < 1e-12
    assert torch.equal(P2[:, 11], q['blur_mask']) and torch.equal(P2[:, 0], q['theta'][:, 0, 0])


def test_simclr_hq_cutout_against_reference_golden(golden, margin):
    """Scope row N4: `simclr_hq_cutout` (augment/__init__.py:124-133) = simclr_hq + RandomApply(CutOut(15), 0.5); golden
    from the reference pipeline, host sampler reproduces the draw order incl. CutOut's two randint draws."""
    from contrad_amd import config
    from contrad_amd.augment import get_augment
    import os
    g = golden('augment')
    x = torch.from_numpy(g['cut_x'])
    B = x.shape[0]
    P = _params(B, theta=torch.from_numpy(g['cut_p_theta']), flip=g['cut_p_flip_sign'], jitter=g['cut_p_jitter_mask'],
                fc=g['cut_p_f_contrast'], fh=g['cut_p_f_h'], fs=g['cut_p_f_s'], fv=g['cut_p_f_v'],
                gray=g['cut_p_gray_mask'], blur=g['cut_p_blur_mask'])
    P[:, 12] = torch.from_numpy(g['cut_p_cut_mask'])
    P[:, 13] = torch.from_numpy(g['cut_p_cut_h']).float()
    P[:, 14] = torch.from_numpy(g['cut_p_cut_w']).float()
    assert 0 < P[:, 12].sum().item() < B                       # the fixture exercises both branches of the mask
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'stylegan2', 'afhq_dog_style64.gin')])
    aug = get_augment(mode='simclr_hq_cutout')
    assert aug.cutout_length == 15 and aug.p_cutout == 0.5
    out = aug.apply(x.to(DEV), P, bool(g['cut_p_contrast_first']), float(g['cut_p_sigma'])).cpu()
    assert (out - torch.from_numpy(g['cut_out'])).abs().max().item() < 1e-4
    seed = int(g['cut_seed'])
    torch.manual_seed(seed); np.random.seed(seed)
    P2, cf2, s2 = aug.sample(B, 64, 64)
    assert cf2 == bool(g['cut_p_contrast_first']) and abs(s2 - float(g['cut_p_sigma'])) < 1e-12
    assert torch.equal(P2[:, :15], P[:, :15])
    # backward through the whole pipeline (cutout mask -> blur transpose -> colour / gather transpose) vs the oracle
    p = {k[len('cut_p_'):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('cut_p_')}
    p['contrast_first'], p['sigma'], p['cut_length'] = bool(p['contrast_first']), float(p['sigma']), 15
    p['cut_h'], p['cut_w'] = p['cut_h'].long(), p['cut_w'].long()
    w = torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    xr = x.clone().requires_grad_()
    (O.simclr_apply(xr, p) * w).sum().backward()
    xd = x.to(DEV).requires_grad_()
    (aug.apply(xd, P, p['contrast_first'], p['sigma']) * w.to(DEV)).sum().backward()
    e = ((xd.grad.cpu() - xr.grad).norm() / xr.grad.norm()).item()
    margin('augment backward simclr_hq_cutout 64^2 (l2)', e, 1e-3)     # observed 2.8e-6
