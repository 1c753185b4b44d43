"""The D-step as one captured hipGraph (engine.GraphedDStep) against the eager launch sequence: same host RNG stream,
same kernels, same order -> bitwise identical weights, optimizer state and losses."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _setup(N):
    from contrad_amd import config
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import set_grad
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b64.gin')])
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture('sndcgan', (32, 32, 3))
    G, D = G.to(DEV).train(), D.to(DEV).train()
    P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=False))
    P.augment_fn = get_augment(mode='simclr').to(DEV)
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    set_grad(G, False); set_grad(D, True)
    x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    return P, G, D, opt, x


@pytest.mark.parametrize('N', [16, 64])
def test_graph_replay_equals_eager_steps(N):
    from contrad_amd.engine import GraphedDStep, d_step
    K, W = 4, 2
    P, G, D, opt, x = _setup(N)
    torch.manual_seed(7); np.random.seed(7)
    losses_e = []
    for _ in range(W + K):
        dl, aux = d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
        losses_e.append((dl.item(), aux['penalty'].item()))
    want = [p.detach().clone() for p in D.parameters()]
    want_u = D.main[0].weight_u.clone()
    want_bn = G.norm_init.running_mean.clone()

    P, G, D, opt, x = _setup(N)
    torch.manual_seed(7); np.random.seed(7)
    g = GraphedDStep(P, G, D, opt, {'loss': 'nonsat'}, x, warmup=W)
    losses_g = []
    for _ in range(K):
        dl, aux = g()
        losses_g.append((dl.item(), aux['penalty'].item()))
    assert losses_g == losses_e[W:]
    for a, p in zip(want, D.parameters()):
        assert torch.equal(a, p.detach())
    assert torch.equal(want_u, D.main[0].weight_u) and torch.equal(want_bn, G.norm_init.running_mean)
    st = opt.state[next(iter(D.parameters()))]
    assert int(st['step']) == W + K


@pytest.mark.parametrize('contrad_script,every', [(False, 1), (True, 2)])
def test_stylegan2_graph_replay_equals_eager_steps(contrad_script, every):
    """GraphedSG2DStep: the StyleGAN2 D-steps (single 3N call + R1 every step / separate calls + lazy R1) replayed from a
    captured graph consume exactly the eager path's random numbers -> bitwise identical weights and losses."""
    from contrad_amd.augment import SimCLRAugment
    from contrad_amd.engine import GraphedSG2DStep, d_step_stylegan2, d_step_stylegan2_contrad, set_grad
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup
    N, K, W = 8, 4, 2
    eager = d_step_stylegan2_contrad if contrad_script else d_step_stylegan2

    def build():
        torch.manual_seed(0); np.random.seed(0)
        G, D = get_architecture('stylegan2', (32, 32, 3))
        G, D = G.to(DEV).train(), D.to(DEV).train()
        P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=False, lbd_r1=0.1,
                                     d_reg_every=every))
        P.augment_fn = SimCLRAugment(scale=(0.2, 1.0))
        opt = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
        set_grad(G, False)
        x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
        return P, G, D, opt, x

    def seed():
        torch.manual_seed(7); np.random.seed(7); torch.cuda.manual_seed(7)

    P, G, D, opt, x = build()
    seed()
    le = []
    for s in range(1, W + 1):
        eager(P, G, D, opt, {'loss': 'nonsat'}, x, s if every == 1 else 1)
    for s in range(1, K + 1):
        dl, aux = eager(P, G, D, opt, {'loss': 'nonsat'}, x, s)
        le.append((dl.item(), aux['penalty'].item(), aux['r1'].item() if 'r1' in aux else None))
    want = [p.detach().clone() for p in D.parameters()]

    P, G, D, opt, x = build()
    seed()
    g = GraphedSG2DStep(P, G, D, opt, {'loss': 'nonsat'}, x, contrad_script=contrad_script, warmup=W)
    lg = []
    for s in range(1, K + 1):
        dl, aux = g(s)
        lg.append((dl.item(), aux['penalty'].item(), aux['r1'].item() if 'r1' in aux else None))
    assert lg == le
    assert any(v[2] is not None for v in lg)
    for a, p in zip(want, D.parameters()):
        assert torch.equal(a, p.detach())


def _state(logdir, names):
    return {n: torch.load(os.path.join(logdir, n), map_location='cpu') for n in names}


def _assert_same(a, b):
    if isinstance(a, dict):
        assert list(a) == list(b)
        for k in a:
            _assert_same(a[k], b[k])
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _assert_same(x, y)
    elif torch.is_tensor(a):
        assert torch.equal(a, b)
    else:
        assert a == b


def test_train_gan_with_graph_writes_the_eager_runs_checkpoints(tmp_path):
    """train_gan.py --graph: D-steps replayed from the captured graph, G-steps (which move G's weights, BN statistics
    and D's power-iteration vectors between them) eager -> bitwise the checkpoints of the eager loop."""
    from contrad_amd.train_gan import main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin = os.path.join(root, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    runs = []
    for tag, extra in (('eager', []), ('graph', ['--graph'])):
        logdir = str(tmp_path / tag)
        main([gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--use_warmup', '--synthetic', '--max_steps', '6',
              '--print_every', '3', '--evaluate_every', '6', '--seed', '5', '--logdir', logdir] + extra)
        runs.append(_state(logdir, ('gen.pt', 'dis.pt', 'optim.pt')))
    _assert_same(runs[0], runs[1])


def test_train_stylegan2_contrad_with_graph_writes_the_eager_runs_checkpoints(tmp_path):
    """train_stylegan2_contraD.py --graph with lazy R1 (every 2nd step eager inside the graphed critic)."""
    from contrad_amd.train_stylegan2_contraD import main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin = os.path.join(root, 'configs', 'gan', 'stylegan2', 'c10_style64.gin')
    runs = []
    for tag, extra in (('eager', []), ('graph', ['--graph'])):
        logdir = str(tmp_path / tag)
        main([gin, 'stylegan2', '--mode=contrad', '--aug=simclr', '--lbd_r1', '0.1', '--d_reg_every', '2',
              '--synthetic', '--batch_size', '8', '--halflife_k', '1', '--ema_start_k', '0', '--print_every', '3',
              '--max_steps', '6', '--evaluate_every', '6', '--seed', '5', '--logdir', logdir] + extra)
        runs.append(_state(logdir, ('gen.pt', 'dis.pt', 'gen_ema.pt', 'optim.pt')))
    _assert_same(runs[0], runs[1])


def _fresh_process(script, args):
    """Run a launcher in a NEW interpreter: cold caches (device constants, workspaces, kernel modules not loaded) --
    the conditions under which a capture in the very first iteration used to fail."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, script)] + args, cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('family', ['sndcgan', 'stylegan2'])
def test_resume_with_graph_in_a_fresh_process(tmp_path, family):
    """--resume + --graph: the optimizer state exists from the first iteration, the process is cold.  The capture has to
    wait for one eager iteration of THIS process (GraphedCritic._may_capture); the resumed run with --graph writes
    bitwise the checkpoints of the resumed eager run."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if family == 'sndcgan':
        script, names = 'train_gan.py', ('gen.pt', 'dis.pt', 'optim.pt')
        base = [os.path.join(root, 'configs', 'gan', 'cifar10', 'c10_b64.gin'), 'sndcgan', '--mode=contrad',
                '--aug=simclr', '--synthetic', '--print_every', '2', '--seed', '5']
    else:
        script, names = 'train_stylegan2_contraD.py', ('gen.pt', 'dis.pt', 'gen_ema.pt', 'optim.pt')
        base = [os.path.join(root, 'configs', 'gan', 'stylegan2', 'c10_style64.gin'), 'stylegan2', '--mode=contrad',
                '--aug=simclr', '--lbd_r1', '0.1', '--d_reg_every', '2', '--synthetic', '--batch_size', '8',
                '--halflife_k', '1', '--ema_start_k', '0', '--print_every', '2', '--seed', '5']
    first = str(tmp_path / 'first')
    _fresh_process(script, base + ['--max_steps', '3', '--evaluate_every', '3', '--logdir', first])
    runs = []
    for tag, extra in (('eager', []), ('graph', ['--graph'])):
        logdir = str(tmp_path / tag)
        _fresh_process(script, base + ['--max_steps', '8', '--evaluate_every', '8', '--resume', first, '--logdir', logdir]
                       + extra)
        runs.append(_state(logdir, names))
    _assert_same(runs[0], runs[1])


def test_graph_with_collectives_on_one_rank_rccl():
    """The multi-process launch path without Python on the critical step (train_gan.py:230-318 as one process per GPU):
    the D-step (overlapped and flat gradient exchange), the G-step and the StyleGAN2 D-step are captured WITH their RCCL
    collectives on a 1-rank group and must reproduce the eager steps bitwise (tests/dist_graph_worker.py, own process)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG_FILE='/dev/stderr')
    r = subprocess.run([sys.executable, os.path.join(here, 'dist_graph_worker.py'), str(port)], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'ALL OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
