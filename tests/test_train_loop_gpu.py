"""GPU smoke of the full training loop (D-step + G-step, warm-up, checkpoint + resume) through the CLI entry."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_gan_cli_runs_checkpoints_and_resumes(tmp_path):
    from contrad_amd.train_gan import main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin = os.path.join(root, 'configs', 'gan', 'cifar10', 'c10_b64.gin')
    logdir = str(tmp_path / 'run')
    main([gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--use_warmup', '--synthetic', '--max_steps', '4',
          '--print_every', '2', '--evaluate_every', '4', '--logdir', logdir])
    for f in ('gen.pt', 'dis.pt', 'optim.pt', 'log.txt'):
        assert os.path.exists(os.path.join(logdir, f)), f
    sd = torch.load(os.path.join(logdir, 'dis.pt'))
    assert 'main.0.weight_orig' in sd and 'main.0.weight_u' in sd and all(torch.isfinite(v).all() for v in sd.values())
    ck = torch.load(os.path.join(logdir, 'optim.pt'))
    assert ck['epoch'] == 4
    main([gin, 'sndcgan', '--mode=contrad', '--aug=simclr', '--synthetic', '--max_steps', '6', '--print_every', '1',
          '--resume', logdir])
    log = open(os.path.join(logdir, 'log.txt')).read()
    assert '[Steps       6]' in log and 'nan' not in log.lower()
