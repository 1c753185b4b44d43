"""Training loop of the reference's ``train_gan.py`` on the MI355X path: same CLI
(``<gin_config> <architecture> --mode=contrad --aug=simclr [--use_warmup --temp --lbd_a --resume ...]``), same step
ordering (train_gan.py:141-179: warm-up, set_grad toggling, D-step, G-step), same checkpoint files
(gen.pt / dis.pt / optim.pt, :211-225), one process per GPU.

Differences by design: data parallelism is ``contrad_amd.engine`` (packed RCCL embedding all-gather inside the loss,
flat gradient all-reduce folded into the fused Adam) instead of DistributedDataParallel; the per-step ``dist.barrier()``
of the reference (:227) is dropped (the all-reduce already synchronises); FID / GIF / tensorboard side paths are out
of scope -- losses are logged to stdout / log.txt.  Datasets: ``--synthetic`` (default when
torchvision is absent) feeds uniform-random CIFAR-shaped batches; otherwise torchvision CIFAR-10/100 as the reference.
"""
import os
import time
from argparse import ArgumentParser
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import config
from .augment import get_augment
from .engine import GradAllReducer, GraphedDStep, GraphedGStep, OverlappedGradReducer, sample_generator, set_grad
from .hostio import THROTTLE
from .models.gan import get_architecture
from .optim import FusedAdam
from .training.gan import setup

IMAGE_SIZES = {'cifar10': (32, 32, 3), 'cifar100': (32, 32, 3), 'cifar10_hflip': (32, 32, 3)}


def parse_args(argv=None):
    parser = ArgumentParser(description='Training script: ContraD on MI355X (one process per GPU).')
    parser.add_argument('gin_config', type=str, help='Path to the gin configuration file')
    parser.add_argument('architecture', type=str, help='Architecture')
    parser.add_argument('--mode', default='std', type=str, help='Training mode (only contrad is on this path)')
    parser.add_argument('--penalty', default='none', type=str)
    parser.add_argument('--aug', default='none', type=str, help='Augmentation (simclr | simclr_hq)')
    parser.add_argument('--use_warmup', action='store_true', help='Use warmup strategy on LR')
    parser.add_argument('--temp', default=0.1, type=float)
    parser.add_argument('--lbd_a', default=1.0, type=float)
    # train_gan.py:55-60: FID / GIF logging is outside the hot path (out of scope) -- the flags are accepted
    # so that the reference's command lines run unchanged
    parser.add_argument('--no_fid', action='store_true', help='accepted for CLI compatibility (FIDs are never tracked here)')
    parser.add_argument('--no_gif', action='store_true', help='accepted for CLI compatibility (no GIFs are written here)')
    parser.add_argument('--n_eval_avg', default=3, type=int, help='accepted for CLI compatibility')
    parser.add_argument('--print_every', default=50, type=int)
    parser.add_argument('--evaluate_every', default=2000, type=int, help='checkpoint period (steps)')
    parser.add_argument('--save_every', default=100000, type=int)
    parser.add_argument('--comment', default='', type=str)
    parser.add_argument('--resume', default=None, type=str)
    parser.add_argument('--finetune', default=None, type=str)
    parser.add_argument('--workers', default=0, type=int)
    # train_gan.py:79-82 (NODE count / node rank of the reference's mp.spawn launch): accepted; this build is one process
    # per GPU of ONE node and takes rank / world size from the launcher's environment (RANK / WORLD_SIZE)
    parser.add_argument('--world-size', default=1, type=int, help='accepted for CLI compatibility (nodes; single-node build)')
    parser.add_argument('--rank', default=0, type=int, help='accepted for CLI compatibility (node rank)')
    parser.add_argument('--port', default=40404, type=int)
    # additions
    parser.add_argument('--synthetic', action='store_true', help='uniform-random images instead of a dataset')
    parser.add_argument('--max_steps', default=None, type=int, help='override options.max_steps')
    parser.add_argument('--logdir', default=None, type=str)
    parser.add_argument('--seed', default=0, type=int)
    parser.add_argument('--graph', action='store_true',
                        help='replay the D- and G-step from captured hipGraphs (simclr pipeline; with several ranks the RCCL collectives are captured too)')
    return parser.parse_args(argv)


def _update_warmup(optimizer, cur_step, warmup, lr):
    """train_gan.py:88-93."""
    if warmup > 0:
        ratio = min(1., (cur_step + 1) / warmup)
        for group in optimizer.param_groups:
            group['lr'] = ratio * lr


@config.configurable('options')
def get_options_dict(dataset=config.REQUIRED, loss=config.REQUIRED, batch_size=64, fid_size=10000, max_steps=200000,
                     warmup=0, n_critic=1, lr=2e-4, lr_d=None, beta=(.5, .999), lbd=10., lbd2=10.):
    """train_gan.py:103-121."""
    if lr_d is None:
        lr_d = lr
    return {"dataset": dataset, "batch_size": batch_size, "fid_size": fid_size, "loss": loss, "max_steps": max_steps,
            "warmup": warmup, "n_critic": n_critic, "lr": lr, "lr_d": lr_d, "beta": beta, "lbd": lbd, "lbd2": lbd2}


def _synthetic_loader(batch, image_size, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    h, w, c = image_size
    while True:
        yield torch.rand(batch, c, h, w, generator=g).to(device, non_blocking=True), None


def _dataset_loader(name, batch, rank, world, workers):
    import torchvision
    import torchvision.transforms as T
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    root = os.environ.get('DATA_DIR', './data')
    cls = torchvision.datasets.CIFAR100 if name == 'cifar100' else torchvision.datasets.CIFAR10
    tf = [T.RandomHorizontalFlip()] if name.endswith('hflip') else []
    ds = cls(root, train=True, download=False, transform=T.Compose(tf + [T.ToTensor()]))
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank)
    loader = DataLoader(ds, shuffle=False, pin_memory=True, num_workers=workers, batch_size=batch, sampler=sampler)
    epoch = 0
    while True:
        for images, targets in loader:
            yield images.cuda(non_blocking=True), targets
        epoch += 1
        sampler.set_epoch(epoch)


class GraphedCritic(object):
    """``--graph``: the critic iteration of ``train_step`` replayed from ONE captured hipGraph (engine.GraphedDStep) and
    the generator step from a second one (engine.GraphedGStep).  Each is captured lazily at its first occurrence AFTER
    its optimizer holds state (i.e. from the second iteration on); every
    replay consumes exactly the host random numbers the eager iteration would, so a run with ``--graph`` produces
    bitwise the checkpoints of a run without (tests/test_graph_gpu.py)."""

    def __init__(self):
        self.step = None
        self.gstep = None
        self.eager_d = self.eager_g = 0       # eager steps seen IN THIS PROCESS (see _may_capture)

    @staticmethod
    def _may_capture(seen, optimizer):
        """Capture only after one eager step has run in this process AND the optimizer holds state.  The state alone is
        not enough: after ``--resume`` it is non-empty at once, but the first step of a fresh process still does
        first-use host work that must not land inside a stream capture (blocking host -> device copies of cached
        constants, first-time workspace allocations, kernel module loads)."""
        return seen >= 1 and len(optimizer.state) > 0

    def generator(self, P, opt, G, D, opt_G, images):
        """The generator step of the iteration from its own captured graph (engine.GraphedGStep); None while it has to
        run eagerly (first iteration of the process)."""
        if self.gstep is None:
            if not self._may_capture(self.eager_g, opt_G):
                self.eager_g += 1
                return None
            self.gstep = GraphedGStep(P, G, D, opt_G, opt, images.size(0), images.size(2), images.size(3))
        return self.gstep()

    def __call__(self, P, opt, G, D, opt_D, images):
        if self.step is None:
            if not self._may_capture(self.eager_d, opt_D):
                self.eager_d += 1
                return None                       # first iteration of this process: eager
            if P.mode != 'contrad':
                raise NotImplementedError("--graph captures the ContraD critic iteration (--mode contrad), not '%s'"
                                          % P.mode)
            self.step = GraphedDStep(P, G, D, opt_D, opt, images, warmup=0)
        self.step.load_images(images)
        return self.step()


def train_step(P, opt, G, D, opt_G, opt_D, loader, step, reducers, graphed=None):
    """One iteration of train_gan.py:141-179.  ``loader`` yields (images, labels); a fresh real batch is drawn for
    every critic iteration (train_gan.py:153-155) and the last one feeds the G-step.  Returns the loss tensors (no
    host sync)."""
    THROTTLE.begin()                          # host stays at most one step ahead of the GPU (hostio.py)
    G.train(); D.train()
    if P.use_warmup:
        _update_warmup(opt_G, step, opt["warmup"], opt["lr"])
        _update_warmup(opt_D, step, opt["warmup"], opt["lr_d"])
    red_G, red_D = reducers
    set_grad(G, False); set_grad(D, True)
    for _ in range(opt['n_critic']):
        images, _labels = next(loader)
        done = graphed(P, opt, G, D, opt_D, images) if graphed is not None else None
        if done is not None:
            d_loss, aux = done
            continue
        gen_images = sample_generator(G, images.size(0), enable_grad=False)
        d_loss, aux = P.train_fn["D"](P, D, opt, images, gen_images)
        loss = d_loss + aux['penalty']
        opt_D.zero_grad()
        loss.backward()
        comm = getattr(D, '_grad_comm', None)
        world = comm.world() if comm is not None else (red_D() if red_D is not None else 1)
        opt_D.step(grad_scale=1.0 / world) if world > 1 else opt_D.step()
    set_grad(G, True); set_grad(D, False)
    g_loss = graphed.generator(P, opt, G, D, opt_G, images) if graphed is not None else None
    if g_loss is None:
        gen_images = sample_generator(G, images.size(0))
        g_loss = P.train_fn["G"](P, D, opt, images, gen_images)
        opt_G.zero_grad()
        g_loss.backward()
        world = red_G() if red_G is not None else 1
        opt_G.step(grad_scale=1.0 / world) if world > 1 else opt_G.step()
    THROTTLE.end()
    # detached: a loss that keeps last iteration's autograd graph (and its AccumulateGrad nodes) alive would tie the next
    # D-step to the stream that graph ran on -- which breaks a hipGraph capture
    return {'G_loss': g_loss.detach(), 'D_loss': d_loss.detach(), 'D_penalty': aux['penalty'].detach(),
            'D_real': aux['d_real'].detach(), 'D_gen': aux['d_gen'].detach()}


def main(argv=None):
    P = parse_args(argv)
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
    options['batch_size'] = options['batch_size'] // world                  # train_gan.py:247
    if options['dataset'] not in IMAGE_SIZES:
        raise NotImplementedError("dataset '%s' (train_gan.py drives %s; the StyleGAN2 high-resolution configs run "
                                  "through train_stylegan2_contraD.py)" % (options['dataset'], sorted(IMAGE_SIZES)))
    image_size = IMAGE_SIZES[options['dataset']]

    torch.manual_seed(P.seed); np.random.seed(P.seed)                       # identical initial weights on all ranks
    G, D = get_architecture(P.architecture, image_size, P=P)
    if P.resume:
        G.load_state_dict(torch.load(f"{P.resume}/gen.pt", map_location='cpu'))
        D.load_state_dict(torch.load(f"{P.resume}/dis.pt", map_location='cpu'))
    if P.finetune:
        D.load_state_dict(torch.load(f"{P.finetune}/dis.pt", map_location='cpu'), strict=False)
        D.reset_parameters(D.linear)
        P.comment += 'ft'
    G, D = G.to(dev), D.to(dev)
    torch.manual_seed(P.seed + 1000 * (rank + 1)); np.random.seed(P.seed + 1000 * (rank + 1))

    opt_G = FusedAdam(G.parameters(), lr=options["lr"], betas=tuple(options["beta"]))
    opt_D = FusedAdam(D.parameters(), lr=options["lr_d"], betas=tuple(options["beta"]))
    starting_step = 1
    if P.resume:
        ck = torch.load(f"{P.resume}/optim.pt", map_location=dev)
        opt_G.load_state_dict(ck['optim_G']); opt_D.load_state_dict(ck['optim_D'])
        starting_step = ck['epoch'] + 1
    logdir = P.logdir or P.resume or f'logs/gan/{P.gin_stem}/{P.architecture}/{P.filename}{P.comment}'
    log_file = None
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        log_file = open(os.path.join(logdir, 'log.txt'), 'a')

    def log(msg):
        if rank == 0:
            print(msg, flush=True)
            log_file.write(msg + '\n'); log_file.flush()

    P.augment_fn = get_augment(mode=P.aug).to(dev)
    reducers = (None, None)
    if world > 1:
        reducers = (GradAllReducer(G.parameters()), None)
        if hasattr(D, 'enable_grad_overlap'):
            D.enable_grad_overlap(OverlappedGradReducer())
        else:
            reducers = (reducers[0], GradAllReducer(D.parameters()))
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

    graphed = None
    if P.graph:
        if P.mode != 'contrad':
            log("--graph captures the ContraD critic iteration (--mode contrad), not '%s' -> eager" % P.mode)
        else:
            graphed = GraphedCritic()
    t0 = time.time()
    for step in range(starting_step, options['max_steps'] + 1):
        losses = train_step(P, options, G, D, opt_G, opt_D, loader, step, reducers, graphed)
        if step % P.print_every == 0:
            vals = {k: float(v.detach()) for k, v in losses.items()}              # the only host sync of the loop
            log('[Steps %7d] [G %.3f] [D %.3f] [pen %.3f] [%.1f img/s]' %
                (step, vals['G_loss'], vals['D_loss'], vals['D_penalty'],
                 P.print_every * options['batch_size'] * world / max(time.time() - t0, 1e-9)))
            t0 = time.time()
        if step % P.evaluate_every == 0 and rank == 0:
            torch.save(G.state_dict(), logdir + '/gen.pt')
            torch.save(D.state_dict(), logdir + '/dis.pt')
            if step % P.save_every == 0:
                torch.save(G.state_dict(), logdir + f'/gen_{step}.pt')
                torch.save(D.state_dict(), logdir + f'/dis_{step}.pt')
            torch.save({'epoch': step, 'optim_G': opt_G.state_dict(), 'optim_D': opt_D.state_dict()},
                       logdir + '/optim.pt')
    if world > 1:
        dist.destroy_process_group()
    return logdir


if __name__ == '__main__':
    main()
