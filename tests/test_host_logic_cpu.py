"""Host-side logic that needs no GPU: RNG draw order of the augmentation samplers against the oracle (incl. simclr_hq
and simclr_hq_cutout), state-dict contracts of every architecture against the oracle's tables / the reference manifest,
the LR schedules and EMA schedule of the StyleGAN2 loops, the optimizer's graph-replay scalars, gin parsing."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import contrad_oracle as O
from oracle import stylegan2_oracle as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('mode,cfg,size', [('simclr', O.SIMCLR_CIFAR, 32), ('simclr_hq', O.SIMCLR_HQ_AFHQ, 64),
                                           ('simclr_hq_cutout', O.SIMCLR_HQ_CUTOUT_AFHQ, 96)])
def test_host_sampler_reproduces_the_reference_draw_order(mode, cfg, size):
    from contrad_amd import config
    from contrad_amd.augment import get_augment
    config.clear_config()
    files = [os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin')]
    if mode != 'simclr':
        files.append(os.path.join(config.CONFIG_ROOT, 'gan', 'stylegan2', 'afhq_dog_style64.gin'))
    config.parse_config_files_and_bindings(files)
    aug = get_augment(mode=mode)
    for seed in (0, 1, 2):
        B = 24
        torch.manual_seed(seed); np.random.seed(seed)
        P, cf, sigma = aug.sample(B, size, size)
        torch.manual_seed(seed); np.random.seed(seed)
        p = O.sample_simclr_params(B, size, size, cfg)
        th = p['theta']
        assert torch.equal(P[:, 0], th[:, 0, 0]) and torch.equal(P[:, 1], th[:, 1, 1])
        assert torch.equal(P[:, 2], th[:, 0, 2]) and torch.equal(P[:, 3], th[:, 1, 2])
        for col, key in ((4, 'flip_sign'), (5, 'jitter_mask'), (6, 'f_contrast'), (7, 'f_h'), (8, 'f_s'), (9, 'f_v'),
                         (10, 'gray_mask')):
            assert torch.equal(P[:, col], p[key]), (mode, key)
        assert cf == p['contrast_first'] and P[0, 15].item() == float(cf)
        if 'blur_mask' in p:
            assert torch.equal(P[:, 11], p['blur_mask']) and abs(sigma - p['sigma']) < 1e-15
        else:
            assert sigma is None
        if 'cut_mask' in p:
            assert torch.equal(P[:, 12], p['cut_mask'])
            assert torch.equal(P[:, 13], p['cut_h'].float()) and torch.equal(P[:, 14], p['cut_w'].float())
    with pytest.raises(NotImplementedError):
        get_augment(mode='diffaug')


def test_state_dict_contracts_of_every_architecture():
    from contrad_amd.models.gan import get_architecture
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'checkpoint_manifest.json')))
    G, D = get_architecture('sndcgan', (32, 32, 3))
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == O.sndcgan_d_param_shapes()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in G.state_dict().items()] == man['sndcgan']['gen']
    G, D = get_architecture('snresnet18', (32, 32, 3))
    shapes = O.snresnet18_param_shapes()
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes and list(D.state_dict()) == list(shapes)
    assert D.d_hidden == 1024 and D.d_penul == 512
    G, D = get_architecture('stylegan2', (32, 32, 3))
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == S.d_param_shapes(32, True)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == S.g_param_shapes(32, True)
    assert [[k, list(v.shape), str(v.dtype)] for k, v in D.state_dict().items()] == man['stylegan2']['dis']
    G, D = get_architecture('stylegan2_512', (512, 512, 3))
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == S.d_param_shapes(512, False, 1.0)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == S.g_param_shapes(512, False, 1.0)
    with pytest.raises(NotImplementedError):
        get_architecture('resnet50', (32, 32, 3))
    # no CPU fallback: the product refuses CPU tensors instead of silently computing somewhere else
    with pytest.raises(RuntimeError):
        D(torch.rand(2, 3, 512, 512))


def test_stylegan2_loop_schedules():
    """_update_warmup / _update_lr (train_stylegan2.py:86-103), the EMA constants (:303-308), option defaults (:126-144)."""
    from contrad_amd import config
    from contrad_amd.train_stylegan2 import IMAGE_SIZES, _update_lr, _update_warmup, get_options_dict, parse_args
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    _update_warmup(opt, 0, 3000, 2e-3)
    assert abs(opt.param_groups[0]['lr'] - min(1., 1 / (3000 + 1e-8)) * 2e-3) < 1e-18
    _update_warmup(opt, 5000, 3000, 2e-3)
    assert opt.param_groups[0]['lr'] == 2e-3
    _update_warmup(opt, 5, 0, 7.0)
    assert opt.param_groups[0]['lr'] == 2e-3                     # warmup 0: untouched
    assert _update_lr(opt, 1500, 64, 1000000, 2e-3) is None      # only every 1000 steps
    assert _update_lr(opt, 2000, 64, 0, 2e-3) is None            # halflife_lr 0: off
    lr = _update_lr(opt, 2000, 64, 1000000, 2e-3)
    assert abs(lr - 0.5 ** (2000 * 64 / 1000000) * 2e-3) < 1e-12 and opt.param_groups[0]['lr'] == lr
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'stylegan2', 'afhq_dog_style64.gin')])
    o = get_options_dict()
    assert (o['dataset'], o['batch_size'], o['lr'], o['lr_d'], tuple(o['beta']), o['n_critic'], o['warmup']) == \
        ('afhq_dog', 64, 0.0025, 0.0025, (0.0, 0.99), 1, 3000)
    assert IMAGE_SIZES[o['dataset']] == (512, 512, 3) and IMAGE_SIZES['cifar10_hflip'] == (32, 32, 3)
    P = parse_args(['x.gin', 'stylegan2_512', '--mode=contrad', '--aug=simclr_hq', '--lbd_r1=0.5', '--no_lazy'])
    assert (P.d_reg_every, P.style_mix, P.halflife_k, P.ema_start_k, P.lbd_r1, P.no_lazy) == (16, 0.9, 20, None, 0.5, True)
    # the reference's EMA decay: accum = 0.5 ** (batch / (halflife_k * 1000))
    assert abs(0.5 ** (64 / (20 * 1000)) - 0.99778429) < 1e-7


def test_fused_adam_graph_scalars_follow_the_launcher_arithmetic():
    """FusedAdam.hyper_values == what contrad_adam_step computes from its float arguments (bitwise-equal replay)."""
    import struct
    from contrad_amd.optim import FusedAdam
    f32 = lambda v: struct.unpack('f', struct.pack('f', v))[0]
    p = torch.nn.Parameter(torch.zeros(4))
    opt = FusedAdam([p], lr=2e-4, betas=(0.5, 0.999))
    opt.state[p] = {'step': 6, 'exp_avg': torch.zeros(4), 'exp_avg_sq': torch.zeros(4)}
    v0 = p._version
    h = opt.hyper_values(grad_scale=0.125)
    assert int(opt.state[p]['step']) == 7 and p._version == v0 + 1
    b1, b2 = f32(0.5), f32(0.999)
    assert h[0] == f32(2e-4) / (1.0 - b1 ** 7) and h[1] == 1.0 / math.sqrt(1.0 - b2 ** 7) and h[2] == 0.125


def test_packed_buffer_zero_fill_only_when_padding_exists():
    from contrad_amd.autograd_ops import _packed_buffer
    a = _packed_buffer((9 * 16, 32), 32, torch.device('cpu'))
    b = _packed_buffer((9 * 16, 4), 1, torch.device('cpu'))
    assert a.shape == (144, 32) and b.shape == (144, 4) and b.abs().sum().item() == 0.0


def test_bench_workload_table_matches_baseline_json():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert base['metric'].startswith('discriminator-step images/sec')
    c = bench.CONFIGS
    assert c['c10_b512']['batch'] == 512 and c['c10_b512']['batch_is_global'] and c['c10_b512']['flop_per_image'] == 4.28e9
    assert c['sg2_32']['batch'] == 64 and c['sg2_32']['d_reg_every'] == 1 and c['sg2_32']['lbd_r1'] == 0.1
    assert c['sg2_512']['batch'] == 16 and c['sg2_512']['d_reg_every'] == 16 and c['sg2_512']['aug'] == 'simclr_hq'
    for name, cfg in c.items():
        assert os.path.exists(os.path.join(ROOT, 'configs', *cfg['gin'])), name
    assert 1024 < bench._free_port() < 65536


def test_bench_graph_watchdog_prints_the_kept_eager_result_and_exits_zero():
    """bench.py's fallback order for N > 1: the eager result is kept, the capture runs under a
    per-rank deadline, past it rank 0 prints the line with the eager result and every rank leaves with exit code 0."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, time, json; sys.path.insert(0, %r)\n"
        "import bench\n"
        "results = {'c10_b512': {'value': 1.0, 'config': {'launch': 'hipGraph replay'}}}\n"
        "names = ['c10_b512', 'sg2_32']\n"
        "def emit():\n"
        "    out = results[names[0]]\n"
        "    rest = {n: results[n] for n in names[1:] if n in results}\n"
        "    if rest: out['other_configs'] = rest\n"
        "    print(json.dumps(out), flush=True)\n"
        "wd = bench._GraphWatchdog(int(sys.argv[1]), 0.3, results, emit)\n"
        "wd.keep('sg2_32', {'value': 2.0, 'config': {'launch': 'eager (graph capture timed out)'}})\n"
        "wd.arm('sg2_32')\n"
        "time.sleep(30)\n"
        "print('not reached')\n" % ROOT)
    for rank, want_line in ((0, True), (1, False)):
        r = subprocess.run([sys.executable, '-c', code, str(rank)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=120, text=True)
        assert r.returncode == 0 and 'not reached' not in r.stdout
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == (1 if want_line else 0)
        if want_line:
            out = json.loads(lines[0])
            assert out['value'] == 1.0 and out['other_configs']['sg2_32']['config']['launch'].startswith('eager (graph')
    # a disarmed watchdog never fires
    code2 = code.replace("time.sleep(30)", "wd.disarm('sg2_32'); time.sleep(1.0); print('reached')").replace("print('not reached')", "")
    r = subprocess.run([sys.executable, '-c', code2, '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, text=True)
    assert r.returncode == 0 and 'reached' in r.stdout and '{' not in r.stdout


def test_train_gan_accepts_every_flag_of_the_reference_cli():
    """train_gan.py:41-85: the reference's command lines must parse unchanged (logging flags are accepted and ignored)."""
    from contrad_amd.train_gan import parse_args
    a = parse_args(['configs/gan/cifar10/c10_b512.gin', 'sndcgan', '--mode', 'contrad', '--aug', 'simclr', '--use_warmup',
                    '--temp', '0.1', '--lbd_a', '1.0', '--no_fid', '--no_gif', '--n_eval_avg', '3', '--print_every', '50',
                    '--evaluate_every', '2000', '--save_every', '100000', '--comment', 'x', '--workers', '0',
                    '--world-size', '1', '--rank', '0', '--port', '40404'])
    assert a.mode == 'contrad' and a.no_fid and a.no_gif and a.n_eval_avg == 3 and a.world_size == 1 and a.rank == 0
