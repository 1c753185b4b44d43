"""Scope row N3: checkpoint interchange with the reference.  The structure of the reference's gen.pt / dis.pt /
optim.pt (key order, shapes, dtypes, torch.optim.Adam's state layout) is pinned by tests/golden/checkpoint_manifest.json
(written by make_golden.py from the imported reference modules); files of exactly that structure are rebuilt here with
seeded values and go through --resume and --finetune; optimizer state interchanges with torch.optim.Adam both ways."""
import json
import os

import pytest
import torch

from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _manifest():
    return json.load(open(os.path.join(ROOT, 'tests', 'golden', 'checkpoint_manifest.json')))


def _seeded_state(entries, seed):
    sd = {}
    for i, (k, shape, dtype) in enumerate(entries):
        g = torch.Generator().manual_seed(seed + i)
        if dtype == 'torch.int64':
            sd[k] = torch.tensor(7, dtype=torch.int64)
        elif k.endswith('running_var'):
            sd[k] = torch.rand(*shape, generator=g) + 0.5
        elif k.endswith(('weight_u', 'weight_v')):
            sd[k] = torch.nn.functional.normalize(torch.randn(*shape, generator=g), dim=0)
        else:
            sd[k] = torch.randn(*shape, generator=g) * 0.02
    return sd


def _reference_format_checkpoint(dirname, arch='sndcgan', epoch=3):
    """Files as the reference's rank 0 writes them (train_gan.py:211-225)."""
    m = _manifest()[arch]
    os.makedirs(dirname, exist_ok=True)
    gsd, dsd = _seeded_state(m['gen'], 100), _seeded_state(m['dis'], 200)
    torch.save(gsd, os.path.join(dirname, 'gen.pt'))
    torch.save(dsd, os.path.join(dirname, 'dis.pt'))
    optim = {'epoch': epoch}
    for tag, sd, key in (('gen', gsd, 'optim_G'), ('dis', dsd, 'optim_D')):
        om = m['optim_' + tag]
        state = {}
        for i, name in enumerate(om['param_names']):
            g = torch.Generator().manual_seed(300 + i)
            state[i] = {'step': torch.tensor(float(epoch)),
                        'exp_avg': torch.randn(sd[name].shape, generator=g) * 1e-3,
                        'exp_avg_sq': torch.rand(sd[name].shape, generator=g) * 1e-6}
        groups = [dict(g_) for g_ in om['param_groups']]
        for g_ in groups:
            g_['betas'] = tuple(g_['betas'])
        optim[key] = {'state': state, 'param_groups': groups}
    torch.save(optim, os.path.join(dirname, 'optim.pt'))
    return gsd, dsd, optim


def test_state_dict_structure_matches_the_reference_manifest():
    from contrad_amd.models.gan import get_architecture
    m = _manifest()
    for arch in ('sndcgan', 'stylegan2'):
        G, D = get_architecture(arch, (32, 32, 3))
        for tag, mod in (('gen', G), ('dis', D)):
            got = [[k, list(v.shape), str(v.dtype)] for k, v in mod.state_dict().items()]
            assert got == m[arch][tag], (arch, tag)
            assert [k for k, _ in mod.named_parameters()] == m[arch]['optim_' + tag]['param_names']


def test_resume_from_a_reference_format_checkpoint(tmp_path):
    from contrad_amd.train_gan import main
    ck = str(tmp_path / 'refck')
    gsd, dsd, optim = _reference_format_checkpoint(ck, epoch=3)
    gin = os.path.join(ROOT, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    main([gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--synthetic', '--max_steps', '5', '--print_every', '1',
          '--evaluate_every', '5', '--resume', ck])
    log = open(os.path.join(ck, 'log.txt')).read()
    assert '[Steps       4]' in log and '[Steps       5]' in log and '[Steps       3]' not in log
    assert 'nan' not in log.lower()
    after = torch.load(os.path.join(ck, 'dis.pt'), map_location='cpu')
    assert list(after) == list(dsd)                                  # same keys, same order: loadable by the reference
    # two Adam steps at lr 2e-4 from the loaded moments: every weight moved, none by more than ~2 * lr
    d = (after['main.0.weight_orig'] - dsd['main.0.weight_orig']).abs()
    assert 0 < d.max().item() < 1e-3
    ck2 = torch.load(os.path.join(ck, 'optim.pt'), map_location='cpu')
    assert ck2['epoch'] == 5 and int(ck2['optim_D']['state'][0]['step']) == 5


def test_finetune_loads_the_trunk_and_reinitialises_the_linear_head(tmp_path):
    """--finetune (train_gan.py:255-266): load_state_dict(strict=False) + D.reset_parameters(D.linear)."""
    from contrad_amd.train_gan import main
    ck = str(tmp_path / 'refck')
    gsd, dsd, _ = _reference_format_checkpoint(ck)
    gin = os.path.join(ROOT, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    logdir = str(tmp_path / 'ft')
    main([gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--synthetic', '--max_steps', '1', '--evaluate_every', '1',
          '--finetune', ck, '--logdir', logdir])
    after = torch.load(os.path.join(logdir, 'dis.pt'), map_location='cpu')
    for k in ('main.0.weight_orig', 'main.12.weight_orig', 'projection.0.weight_orig', 'projection2.2.weight_orig'):
        assert (after[k] - dsd[k]).abs().max().item() < 5e-4, k       # loaded, then one Adam step
    for k in ('linear.l1.weight_orig', 'linear.l2.weight_orig'):
        assert (after[k] - dsd[k]).abs().max().item() > 1e-2, k       # re-drawn
    assert after['linear.l1.bias'].abs().max().item() < 5e-4            # zero-initialised bias (+ one step)


def test_optimizer_state_interchanges_with_torch_adam():
    """optim.pt written by the reference (torch.optim.Adam) continues on FusedAdam and vice versa."""
    from contrad_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 3, 3), (64,), (512, 8192), (1,)]
    p0 = [torch.randn(*s, generator=g) * 0.05 for s in shapes]
    grads = [[torch.randn(*s, generator=g) * 0.01 for s in shapes] for _ in range(3)]
    # reference optimizer: two steps, save
    ref = [torch.nn.Parameter(p.clone()) for p in p0]
    opt_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    for t in range(2):
        for p, gr in zip(ref, grads[t]):
            p.grad = gr.clone()
        opt_ref.step()
    saved = opt_ref.state_dict()
    # continue on the HIP optimizer from the saved state
    mine = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    opt = FusedAdam(mine, lr=2e-4, betas=(0.5, 0.999))
    opt.load_state_dict(saved)
    for p, gr in zip(mine, grads[2]):
        p.grad = gr.clone().to(DEV)
    opt.step()
    for p, gr in zip(ref, grads[2]):
        p.grad = gr.clone()
    opt_ref.step()
    for a, b in zip(mine, ref):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 1e-6
    # and back: FusedAdam's state continues on torch.optim.Adam
    back = [torch.nn.Parameter(p.detach().cpu().clone()) for p in mine]
    opt_back = torch.optim.Adam(back, lr=2e-4, betas=(0.5, 0.999))
    sd = opt.state_dict()
    sd['state'] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd['state'].items()}
    opt_back.load_state_dict(sd)
    for plist, o in ((back, opt_back), (ref, opt_ref)):
        for p, gr in zip(plist, grads[0]):
            p.grad = gr.clone()
        o.step()
    for a, b in zip(back, ref):
        assert (a.detach() - b.detach()).abs().max().item() < 1e-6
