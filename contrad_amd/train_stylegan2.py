"""Training loops of the reference's ``train_stylegan2.py`` and ``train_stylegan2_contraD.py`` on the MI355X path.

Same CLI (``<gin_config> <architecture> --mode=contrad --aug=simclr[_hq] --lbd_r1 .. [--no_lazy --d_reg_every
--style_mix --halflife_k --ema_start_k --halflife_lr --use_warmup --resume --finetune ...]``), same step ordering
(train_stylegan2.py:147-233 / train_stylegan2_contraD.py:182-246): LR warm-up / decay -> EMA ``accumulate`` -> G-step
FIRST -> D-step with the lazy R1 penalty ``(0.5*lbd_r1) * r1 * d_reg_every`` every ``d_reg_every`` steps (``--no_lazy``
=> every step) -> extra critic iterations; same checkpoint files (gen.pt / dis.pt / gen_ema.pt / optim.pt).

The two scripts differ exactly where the reference's do:
  * ``train_stylegan2``        : the D-step re-uses the G-step's fake batch (detached) and makes ONE 3N-image D call
                                 through ``P.train_fn["D"]`` (train_stylegan2.py:184-212);
  * ``train_stylegan2_contraD``: fresh fakes under no_grad, fakes (N) and the two real views (2N) in SEPARATE D calls
                                 (G_D.forward, train_stylegan2_contraD.py:117-164).
The reference parallelises the second one with ``nn.DataParallel(G_D)`` (parameters re-broadcast on every forward, outputs
gathered on GPU 0).  Here both are one process per GPU: per-rank batch = batch_size / world, the embeddings travel in
ONE packed RCCL all-gather inside the loss, parameter gradients in flat all-reduces folded into the fused Adam -- the
global loss of the DataParallel formulation, without the per-step parameter broadcast.
FID / GIF / tensorboard side paths are out of scope.
"""
import os
import time
from argparse import ArgumentParser
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import config, ops
from .augment import get_augment
from .engine import (GradAllReducer, GraphedSG2DStep, GraphedSG2GStep, _sg2_fakes, loss_D_fn_separate, r1_loss, set_grad,
                     setup_grad_exchange)
from .hostio import THROTTLE
from .models.gan import get_architecture
from .optim import FusedAdam
from .training.gan import setup
from .training.gan.contrad import _GanGLoss

# datasets.py:10,57,99,115,131 of the reference
IMAGE_SIZES = {'cifar10': (32, 32, 3), 'cifar100': (32, 32, 3), 'cifar10_hflip': (32, 32, 3),
               'cifar100_hflip': (32, 32, 3), 'celeba128': (128, 128, 3), 'afhq_cat': (512, 512, 3),
               'afhq_dog': (512, 512, 3), 'afhq_wild': (512, 512, 3)}


def parse_args(argv=None, contrad_script=False):
    parser = ArgumentParser(description='Training script: StyleGAN2%s on MI355X (one process per GPU).'
                                        % (' + ContraD' if contrad_script else ''))
    parser.add_argument('gin_config', type=str, help='Path to the gin configuration file')
    parser.add_argument('architecture', type=str, help='Architecture')
    parser.add_argument('--mode', default='std', type=str, help='Training mode (only contrad is on this path)')
    parser.add_argument('--penalty', default='none', type=str)
    parser.add_argument('--aug', default='none', type=str, help='Augmentation (simclr | simclr_hq)')
    parser.add_argument('--use_warmup', action='store_true', help='Use warmup strategy on LR')
    parser.add_argument('--workers', default=8, type=int)
    parser.add_argument('--temp', default=0.1, type=float)
    parser.add_argument('--lbd_a', default=1.0, type=float)
    # StyleGAN2 options (train_stylegan2.py:61-75)
    parser.add_argument('--no_lazy', action='store_true', help='Do not use lazy regularization')
    parser.add_argument('--d_reg_every', type=int, default=16)
    parser.add_argument('--lbd_r1', type=float, default=10)
    parser.add_argument('--style_mix', default=0.9, type=float)
    parser.add_argument('--halflife_k', default=20, type=int)
    parser.add_argument('--ema_start_k', default=None, type=int)
    parser.add_argument('--halflife_lr', default=0, type=int)
    parser.add_argument('--no_fid', action='store_true')
    parser.add_argument('--no_gif', action='store_true')
    parser.add_argument('--n_eval_avg', default=3, type=int)
    parser.add_argument('--print_every', default=50, type=int)
    parser.add_argument('--evaluate_every', default=2000, type=int, help='checkpoint period (steps)')
    parser.add_argument('--save_every', default=100000, type=int)
    parser.add_argument('--comment', default='', type=str)
    parser.add_argument('--resume', default=None, type=str)
    parser.add_argument('--finetune', default=None, type=str)
    # additions
    parser.add_argument('--port', default=40405, type=int)
    parser.add_argument('--synthetic', action='store_true', help='uniform-random images instead of a dataset')
    parser.add_argument('--max_steps', default=None, type=int, help='override options.max_steps')
    parser.add_argument('--batch_size', default=None, type=int, help='override options.batch_size (global)')
    parser.add_argument('--logdir', default=None, type=str)
    parser.add_argument('--seed', default=0, type=int)
    parser.add_argument('--graph', action='store_true',
                        help='replay the D- and G-step from captured hipGraphs (collectives included; the ContraD script, whose '
                             'D-step draws its own fakes)')
    return parser.parse_args(argv)


def _update_warmup(optimizer, cur_step, warmup, lr):
    """train_stylegan2.py:86-91."""
    if warmup > 0:
        ratio = min(1., (cur_step + 1) / (warmup + 1e-8))
        for group in optimizer.param_groups:
            group['lr'] = ratio * lr


def _update_lr(optimizer, cur_step, batch_size, halflife_lr, lr, mult=1.0):
    """train_stylegan2.py:94-103."""
    if halflife_lr > 0 and (cur_step > 0) and (cur_step % 1000 == 0):
        ratio = (cur_step * batch_size) / halflife_lr
        lr_w = (0.5 ** ratio) * lr * mult
        for group in optimizer.param_groups:
            group['lr'] = lr_w
        return lr_w
    return None


@config.configurable('options')
def get_options_dict(dataset=config.REQUIRED, loss=config.REQUIRED, batch_size=32, fid_size=10000, max_steps=800000,
                     warmup=0, n_critic=1, lr=0.002, lr_d=None, beta=(.0, .99), lbd=10., lbd2=10.):
    """train_stylegan2.py:126-144."""
    if lr_d is None:
        lr_d = lr
    return {"dataset": dataset, "batch_size": batch_size, "fid_size": fid_size, "loss": loss, "max_steps": max_steps,
            "warmup": warmup, "n_critic": n_critic, "lr": lr, "lr_d": lr_d, "beta": beta, "lbd": lbd, "lbd2": lbd2}


@torch.no_grad()
def accumulate(model_dst, model_src, decay=0.999):
    """utils.accumulate (utils.py:130-143): dst = decay * dst + (1 - decay) * src for the parameters (one fused axpby
    launch per tensor), buffers copied."""
    params_dst = dict(model_dst.named_parameters())
    params_src = dict(model_src.named_parameters())
    for k, p in params_dst.items():
        ops.axpby_(p.data, params_src[k].data, decay, 1.0 - decay)
        torch.autograd.graph.increment_version(p)
    buf_src = dict(model_src.named_buffers())
    for k, b in model_dst.named_buffers():
        b.copy_(buf_src[k])


def sample_generator(G, num_samples, style_mix=0.9, enable_grad=True):
    """_sample_generator (train_stylegan2.py:116-123)."""
    with torch.set_grad_enabled(enable_grad):
        return _sg2_fakes(G, num_samples, style_mix)     # z, mixing latent and per-layer noise in one device draw


def loss_G_nonsat(d_gen):
    """_loss_G_fn (train_stylegan2_contraD.py:112-114)."""
    return _GanGLoss.apply(d_gen, 'nonsat')


def _opt_step(opt, reducer):
    world = reducer() if reducer is not None else 1
    opt.step(grad_scale=1.0 / world) if world > 1 else opt.step()


# train_stylegan2_contraD.py calls G_D.forward WITHOUT its style_mix argument (:207,218 -> the default 0.9 of :128): the
# ``--style_mix`` flag only names the log directory there (:349).  Reproduced as is.
CONTRAD_SCRIPT_STYLE_MIX = 0.9


class GraphedCritic(object):
    """``--graph`` for train_stylegan2_contraD.py: its D-step (fresh fakes, two D calls, lazy R1) is exactly
    engine.d_step_stylegan2_contrad, so it is replayed from engine.GraphedSG2DStep -- captured at the first D-step after
    the optimizer holds state; the lazy-R1 steps run eagerly inside it.  Same random numbers as the eager iteration."""

    def __init__(self):
        self.step = None
        self.gstep = None
        self.eager_d = self.eager_g = 0       # eager steps seen IN THIS PROCESS

    @staticmethod
    def _may_capture(seen, optimizer):
        """One eager step in this process AND optimizer state (after ``--resume`` the state exists at once, but a fresh
        process's first step does first-use host work -- constant uploads, workspace allocation, module loads -- that
        must not fall inside a stream capture)."""
        return seen >= 1 and len(optimizer.state) > 0

    def generator(self, P, opt, G, D, opt_G, images):
        """The generator step from its own captured graph (engine.GraphedSG2GStep); None while it runs eagerly."""
        if self.gstep is None:
            if not self._may_capture(self.eager_g, opt_G):
                self.eager_g += 1
                return None
            self.gstep = GraphedSG2GStep(P, G, D, opt_G, opt, images.size(0), images.size(2), images.size(3),
                                         style_mix=CONTRAD_SCRIPT_STYLE_MIX)
        return self.gstep()

    def __call__(self, P, opt, G, D, opt_D, images, step):
        if self.step is None:
            if not self._may_capture(self.eager_d, opt_D):
                self.eager_d += 1
                return None
            if P.mode != 'contrad':
                raise NotImplementedError("--graph captures the ContraD D-step (--mode contrad), not '%s'" % P.mode)
            self.step = GraphedSG2DStep(P, G, D, opt_D, opt, images, contrad_script=True, style_mix=CONTRAD_SCRIPT_STYLE_MIX,
                                        warmup=0)
        self.step.load_images(images)
        return self.step(step)


def train_iteration(P, opt, G, D, g_ema, opt_G, opt_D, loader, step, reducers, contrad_script, graphed=None):
    """One iteration of train_stylegan2.py:147-233 (contrad_script False) / train_stylegan2_contraD.py:182-246 (True).
    Returns the loss tensors (no host sync)."""
    THROTTLE.begin()
    red_G, red_D = reducers
    style_mix = CONTRAD_SCRIPT_STYLE_MIX if contrad_script else P.style_mix
    d_regularize = (step % P.d_reg_every == 0) and (P.lbd_r1 > 0)
    if P.use_warmup:
        _update_warmup(opt_G, step, opt["warmup"], opt["lr"])
        _update_warmup(opt_D, step, opt["warmup"], opt["lr_d"])
    lr_note = None
    if (not P.use_warmup) or step > opt["warmup"]:
        cur_lr_g = _update_lr(opt_G, step, opt["global_batch_size"], P.halflife_lr, opt["lr"])
        cur_lr_d = _update_lr(opt_D, step, opt["global_batch_size"], P.halflife_lr, opt["lr_d"])
        if cur_lr_d and cur_lr_g:
            lr_note = (cur_lr_g, cur_lr_d)
    do_ema = (step * opt['global_batch_size']) > (P.ema_start_k * 1000)
    accumulate(g_ema, G, P.accum if do_ema else 0)

    G.train(); D.train()
    images, _labels = next(loader)
    N = images.size(0)
    out = {}

    # ---- generator step first ----
    set_grad(G, True); set_grad(D, False)
    g_loss = graphed.generator(P, opt, G, D, opt_G, images) if graphed is not None else None
    if g_loss is None:
        gen_images = sample_generator(G, N, style_mix=style_mix, enable_grad=True)
        if contrad_script:      # G_D.forward(train_G=True): D(augment(G(z)), sg_linear=False, ...) -> d_gen
            d_gen, _aux = D(P.augment_fn(gen_images), sg_linear=False, projection=True, projection2=True)
            g_loss = loss_G_nonsat(d_gen)
        else:
            g_loss = P.train_fn["G"](P, D, opt, images, gen_images)
        opt_G.zero_grad()
        g_loss.backward()
        _opt_step(opt_G, red_G)
    out['G_loss'] = g_loss.detach()

    # ---- discriminator step ----
    set_grad(G, False); set_grad(D, True)

    def d_loss_of(images, gen_images):
        if contrad_script:
            return loss_D_fn_separate(P, D, opt, images, gen_images)
        return P.train_fn["D"](P, D, opt, images, gen_images)

    done = graphed(P, opt, G, D, opt_D, images, step) if graphed is not None else None
    if done is not None:
        d_loss, aux = done
        if 'r1' in aux:
            out['D_r1'] = aux['r1'].detach()
    else:
        if contrad_script:
            gen_images = sample_generator(G, N, style_mix=style_mix, enable_grad=False)
        d_loss, aux = d_loss_of(images, gen_images.detach())
        loss = d_loss + aux['penalty']
        if d_regularize:
            r1 = r1_loss(D, images, P.augment_fn)
            loss = loss + (0.5 * P.lbd_r1) * r1 * P.d_reg_every
            out['D_r1'] = r1.detach()
        opt_D.zero_grad()
        loss.backward()
        _opt_step(opt_D, red_D)
    for _ in range(opt['n_critic'] - 1):
        images, _labels = next(loader)
        gen_images = sample_generator(G, images.size(0), style_mix=style_mix, enable_grad=False)
        d_loss, aux = d_loss_of(images, gen_images)
        opt_D.zero_grad()
        (d_loss + aux['penalty']).backward()
        _opt_step(opt_D, red_D)
    G.eval(); D.eval()
    THROTTLE.end()
    out.update({'D_loss': d_loss.detach(), 'D_penalty': aux['penalty'].detach(), 'D_real': aux['d_real'].detach(),
                'D_gen': aux['d_gen'].detach(), 'lr_note': lr_note})
    return out


def _synthetic_loader(batch, image_size, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    h, w, c = image_size
    pool = [torch.rand(batch, c, h, w, generator=g).to(device) for _ in range(4)]     # resident, cycled
    i = 0
    while True:
        yield pool[i % len(pool)], None
        i += 1


def _dataset_loader(name, batch, rank, world, workers):
    import torchvision
    import torchvision.transforms as T
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    root = os.environ.get('DATA_DIR', 'data/')
    if name.startswith('cifar'):
        cls = torchvision.datasets.CIFAR100 if name.startswith('cifar100') else torchvision.datasets.CIFAR10
        tf = [T.RandomHorizontalFlip()] if name.endswith('hflip') else []
        ds = cls(root, train=True, download=False, transform=T.Compose(tf + [T.ToTensor()]))
    elif name.startswith('afhq_'):
        ds = torchvision.datasets.ImageFolder(os.path.join(root, 'afhq/%s/train' % name[5:]),
                                              T.Compose([T.RandomHorizontalFlip(), T.ToTensor()]))
    elif name == 'celeba128':
        ds = torchvision.datasets.ImageFolder(os.path.join(root, 'CelebAMask-HQ/CelebA-128-split/train'), T.ToTensor())
    else:
        raise NotImplementedError(name)
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True)
    loader = DataLoader(ds, shuffle=False, pin_memory=True, num_workers=workers, batch_size=batch, sampler=sampler,
                        drop_last=True)
    epoch = 0
    while True:
        for images, targets in loader:
            yield images.cuda(non_blocking=True), targets
        epoch += 1
        sampler.set_epoch(epoch)


def main(argv=None, contrad_script=False):
    P = parse_args(argv, contrad_script)
    if P.comment:
        P.comment = '_' + P.comment
    P.gin_stem = Path(P.gin_config).stem
    P = setup(P)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(P.port))
        dist.init_process_group('nccl', device_id=dev)
    P.rank, P.distributed = rank, world > 1

    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'), P.gin_config])
    options = get_options_dict()
    if P.max_steps is not None:
        options['max_steps'] = P.max_steps
    if P.batch_size is not None:
        options['batch_size'] = P.batch_size
    if options['loss'] != 'nonsat' and contrad_script:
        raise NotImplementedError('train_stylegan2_contraD.py hard-codes the non-saturating loss (:105,:113)')
    if options['dataset'] not in IMAGE_SIZES:
        raise NotImplementedError("dataset '%s'" % options['dataset'])
    image_size = IMAGE_SIZES[options['dataset']]
    if options['batch_size'] % world:
        raise ValueError('batch_size %d is not divisible by the %d ranks' % (options['batch_size'], world))
    options['global_batch_size'] = options['batch_size']            # the schedules count GLOBAL images
    options['batch_size'] = options['batch_size'] // world
    if P.no_lazy:
        P.d_reg_every = 1
    if P.ema_start_k is None:
        P.ema_start_k = P.halflife_k
    P.accum = 0.5 ** (options['global_batch_size'] / (P.halflife_k * 1000))

    torch.manual_seed(P.seed); np.random.seed(P.seed)               # identical initial weights on all ranks
    G, D = get_architecture(P.architecture, image_size, P=P)
    g_ema, _ = get_architecture(P.architecture, image_size, P=P)
    if P.resume:
        G.load_state_dict(torch.load(f"{P.resume}/gen.pt", map_location='cpu'))
        D.load_state_dict(torch.load(f"{P.resume}/dis.pt", map_location='cpu'))
        g_ema.load_state_dict(torch.load(f"{P.resume}/gen_ema.pt", map_location='cpu'))
    if P.finetune:
        D.load_state_dict(torch.load(f"{P.finetune}/dis.pt", map_location='cpu'), strict=False)
        D.reset_parameters(D.linear)
        P.comment += 'ft'
    G, D, g_ema = G.to(dev), D.to(dev), g_ema.to(dev)
    g_ema.eval()
    torch.manual_seed(P.seed + 1000 * (rank + 1)); np.random.seed(P.seed + 1000 * (rank + 1))
    torch.cuda.manual_seed(P.seed + 1000 * (rank + 1))
    P.augment_fn = get_augment(mode=P.aug).to(dev)

    opt_G = FusedAdam(G.parameters(), lr=options["lr"], betas=tuple(options["beta"]))
    opt_D = FusedAdam(D.parameters(), lr=options["lr_d"], betas=tuple(options["beta"]))
    starting_step = 1
    if P.resume:
        ck = torch.load(f"{P.resume}/optim.pt", map_location=dev)
        opt_G.load_state_dict(ck['optim_G']); opt_D.load_state_dict(ck['optim_D'])
        starting_step = ck['epoch'] + 1
    desc = f"R{P.lbd_r1}_mix{P.style_mix}_H{P.halflife_k}"
    if P.halflife_lr > 0:
        desc += f"_lr{P.halflife_lr / 1000000:.1f}M"
    desc += "_NoLazy" if P.no_lazy else "_Lazy"
    sub = 'gan_dp' if contrad_script else 'gan'
    logdir = P.logdir or P.resume or f'logs/{sub}/st_{P.gin_stem}/{P.architecture}/{P.filename}_{desc}{P.comment}'
    log_file = None
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        log_file = open(os.path.join(logdir, 'log.txt'), 'a')

    def log(msg):
        if rank == 0:
            print(msg, flush=True)
            log_file.write(msg + '\n'); log_file.flush()

    reducers = (None, None)
    if world > 1:
        reducers = (GradAllReducer(G.parameters()), setup_grad_exchange(D))      # D: weights exchanged inside the backward
    use_synth = P.synthetic
    if not use_synth:
        try:
            import torchvision  # noqa: F401
        except ImportError:
            log('torchvision not available -> --synthetic')
            use_synth = True
    loader = _synthetic_loader(options['batch_size'], image_size, dev, P.seed + rank) if use_synth else \
        _dataset_loader(options['dataset'], options['batch_size'], rank, world, P.workers)
    log(f"# Params - G: {sum(p.numel() for p in G.parameters())}, D: {sum(p.numel() for p in D.parameters())}")
    log(str(options))
    log(f"Use G moving average: {P.accum}")

    graphed = None
    if P.graph:
        if not contrad_script:
            log('--graph: train_stylegan2_contraD.py only (train_stylegan2.py feeds the D-step the G-step\'s fakes) '
                '-> eager')
        elif P.mode != 'contrad':
            log("--graph captures the ContraD D-step (--mode contrad), not '%s' -> eager" % P.mode)
        else:
            graphed = GraphedCritic()
    t0 = time.time()
    for step in range(starting_step, options['max_steps'] + 1):
        losses = train_iteration(P, options, G, D, g_ema, opt_G, opt_D, loader, step, reducers, contrad_script,
                                 graphed)
        if losses['lr_note']:
            log('LR Updated: [G %.5f] [D %.5f]' % losses['lr_note'])
        if step % P.print_every == 0:
            vals = {k: float(v) for k, v in losses.items() if torch.is_tensor(v)}      # the only host sync of the loop
            log('[Steps %7d] [G %.3f] [D %.3f] [pen %.3f]%s [%.1f img/s]' %
                (step, vals['G_loss'], vals['D_loss'], vals['D_penalty'],
                 (' [r1 %.4g]' % vals['D_r1']) if 'D_r1' in vals else '',
                 P.print_every * options['global_batch_size'] / max(time.time() - t0, 1e-9)))
            t0 = time.time()
        if step % P.evaluate_every == 0 and rank == 0:
            torch.save(G.state_dict(), logdir + '/gen.pt')
            torch.save(D.state_dict(), logdir + '/dis.pt')
            torch.save(g_ema.state_dict(), logdir + '/gen_ema.pt')
            if step % P.save_every == 0:
                torch.save(G.state_dict(), logdir + f'/gen_{step}.pt')
                torch.save(D.state_dict(), logdir + f'/dis_{step}.pt')
                torch.save(g_ema.state_dict(), logdir + f'/gen_ema_{step}.pt')
            torch.save({'epoch': step, 'optim_G': opt_G.state_dict(), 'optim_D': opt_D.state_dict()},
                       logdir + '/optim.pt')
    if world > 1:
        dist.destroy_process_group()
    return logdir


if __name__ == '__main__':
    main()
