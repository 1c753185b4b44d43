"""INTEGRATION.md as an executable fixture (scope row B1): the five re-export modules of its section 2 are written to
a scratch directory exactly as printed there, imported under the REFERENCE's module names (models.gan, augment,
training.criterion, training.gan.contrad, third_party.gather_layer), and driven with the reference loop's call
sequence (train_gan.py:141-179: set_grad, _sample_generator, P.train_fn["D"], backward, optimizer step, G-step) --
with ``torch.optim.Adam`` as in the reference (train_gan.py:273-274)."""
import os
import re
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_NAMES = ('models', 'augment', 'training', 'third_party')


def _materialise(tmp):
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text.split('## 2.')[1].split('## 3.')[0]
    code = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    files = {}
    cur = None
    for line in code.splitlines():
        m = re.match(r'#\s+(\S+\.py)\s', line)
        if m:
            cur = m.group(1)
            files[cur] = []
        elif cur is not None:
            files[cur].append(line)
    assert set(files) == {'models/gan/__init__.py', 'augment/__init__.py', 'training/criterion.py',
                          'training/gan/contrad.py', 'training/gan/__init__.py', 'third_party/gather_layer.py'}, set(files)
    for rel, lines in files.items():
        path = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path, 'w').write('\n'.join(lines) + '\n')
        d = os.path.dirname(path)
        while os.path.abspath(d) != os.path.abspath(tmp):          # package markers up to the scratch root
            init = os.path.join(d, '__init__.py')
            if not os.path.exists(init):
                open(init, 'w').write('')
            d = os.path.dirname(d)


def test_reference_call_sequence_through_the_re_exports(tmp_path):
    tmp = str(tmp_path)
    _materialise(tmp)
    stale = [k for k in sys.modules if k.split('.')[0] in REF_NAMES]
    assert not stale, stale
    sys.path.insert(0, tmp)
    try:
        # ---- what the reference's train_gan.py imports (train_gan.py:19-30) ----
        from augment import get_augment
        from models.gan import get_architecture
        from training.gan import setup
        from training.criterion import nt_xent
        from training.gan.contrad import supcon_fake, loss_D_fn, loss_G_fn
        from third_party.gather_layer import GatherLayer
        import models.gan
        assert os.path.abspath(models.gan.__file__).startswith(os.path.abspath(tmp))
        from contrad_amd import config
        from contrad_amd.engine import set_grad            # utils.set_grad in the reference (utils.py:125-127)
        config.clear_config()
        config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                                os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                                os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b64.gin')])
        torch.manual_seed(0); np.random.seed(0)
        P = Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, penalty='none', distributed=False)
        P = setup(P)
        assert P.filename == 'contrad_simclr_L1.0_T0.1' and P.train_fn['D'] is loss_D_fn and P.train_fn['G'] is loss_G_fn
        generator, discriminator = get_architecture('sndcgan', (32, 32, 3), P=P)
        generator, discriminator = generator.cuda(), discriminator.cuda()
        P.augment_fn = get_augment(mode=P.aug).cuda()
        opt = {'loss': 'nonsat', 'batch_size': 16, 'n_critic': 1}
        opt_G = torch.optim.Adam(generator.parameters(), lr=2e-4, betas=(0.5, 0.999))      # train_gan.py:273-274
        opt_D = torch.optim.Adam(discriminator.parameters(), lr=2e-4, betas=(0.5, 0.999))
        before = [p.detach().clone() for p in discriminator.parameters()]
        gbefore = [p.detach().clone() for p in generator.parameters()]
        images = torch.rand(16, 3, 32, 32).cuda()

        def _sample_generator(G, num_samples, enable_grad=True):                            # train_gan.py:96-100
            latent_samples = G.sample_latent(num_samples)
            if enable_grad:
                return G(latent_samples)
            with torch.no_grad():
                return G(latent_samples)

        for step in range(2):                                                                 # train_gan.py:141-179
            generator.train(); discriminator.train()
            set_grad(generator, False); set_grad(discriminator, True)
            for _ in range(opt['n_critic']):
                gen_images = _sample_generator(generator, images.size(0), enable_grad=False)
                d_loss, aux = P.train_fn["D"](P, discriminator, opt, images, gen_images)
                loss = d_loss + aux['penalty']
                opt_D.zero_grad()
                loss.backward()
                opt_D.step()
            set_grad(generator, True); set_grad(discriminator, False)
            gen_images = _sample_generator(generator, images.size(0))
            g_loss = P.train_fn["G"](P, discriminator, opt, images, gen_images)
            opt_G.zero_grad()
            g_loss.backward()
            opt_G.step()
            vals = [g_loss.item(), d_loss.item(), aux['penalty'].item(), aux['d_real'].item(), aux['d_gen'].item()]
            assert all(np.isfinite(v) for v in vals), vals
        assert all(not torch.equal(a, p.detach()) for a, p in zip(before, discriminator.parameters()))
        assert any(not torch.equal(a, p.detach()) for a, p in zip(gbefore, generator.parameters()))
        # the stand-alone loss entry points the StyleGAN2 script imports (train_stylegan2_contraD.py:35-36)
        torch.manual_seed(0)
        z = torch.nn.functional.normalize(torch.randn(12, 16)).cuda()
        assert abs(nt_xent(z[:4], z[4:8], temperature=0.1).item() - 4.954558372497559) < 1e-4
        assert abs(supcon_fake(z[:4], z[4:8], z[8:], temperature=0.1).item() - 3.962817430496216) < 1e-4
        assert callable(GatherLayer.apply)
    finally:
        sys.path.remove(tmp)
        for k in [k for k in sys.modules if k.split('.')[0] in REF_NAMES]:
            del sys.modules[k]
