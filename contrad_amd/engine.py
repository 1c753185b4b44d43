"""The discriminator step of the reference training loop (train_gan.py:153-163) as a reusable function, plus the
data-parallel gradient exchange that replaces DistributedDataParallel's reducer (train_gan.py:311-313).

Data parallelism: one process per GPU; each rank owns N_local real images, draws its own fakes and
augmentation parameters; ONE packed RCCL all-gather of embeddings inside the loss (training/gan/contrad.py), then a
SUM all-reduce of the parameter gradients whose 1/world_size is folded into the fused Adam kernel's grad_scale.
The discriminator's backward hands autograd views of two flat buffers (all weight gradients, all bias gradients),
so the exchange is two large collectives over xGMI instead of DDP's 25 MB buckets.
"""
import os

import torch
import torch.distributed as dist

from . import ops as _ops
from .hostio import THROTTLE


# Test hook: run every collective code path on a 1-rank process group (set by bench.py --force-dist / tests).
FORCE_DIST = False


def dist_on():
    """True when this process is one rank of a multi-rank job (or the 1-rank test hook is set)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_DIST)


_dist_on = dist_on


class GradAllReducer(object):
    """Sum-all-reduce ``p.grad`` for all params, coalescing gradients that are views of one storage into a single
    collective over the covering span (the HIP discriminators return their weight gradients as views of one flat
    buffer).  Gradients that stay separate small tensors (the ~40 FusedLeakyReLU / head biases of the StyleGAN2
    discriminator: a latency-bound collective each) are packed into ONE staging buffer, reduced together and copied
    back (``torch._foreach_copy_``: two launches instead of 40 collectives)."""

    SMALL = 1 << 18          # floats: spans below 1 MB are packed

    def __init__(self, params):
        self.params = [p for p in params]

    def spans(self):
        groups = {}
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if not g.is_contiguous():
                raise RuntimeError('GradAllReducer needs contiguous gradients')
            key = g.untyped_storage().data_ptr()
            lo = g.storage_offset()
            hi = lo + g.numel()
            if key in groups:
                t, a, b = groups[key]
                groups[key] = (t, min(a, lo), max(b, hi))
            else:
                groups[key] = (g, lo, hi)
        out = []
        for g, lo, hi in groups.values():
            out.append(g.as_strided((hi - lo,), (1,), lo))
        return out

    def __call__(self):
        if not _dist_on():
            return 1
        spans = self.spans()
        small = [t for t in spans if t.numel() < self.SMALL]
        if len(small) > 1:
            stage = torch.cat([t.reshape(-1) for t in small])
            dist.all_reduce(stage)
            torch._foreach_copy_(small, list(torch.split(stage, [t.numel() for t in small])))
            spans = [t for t in spans if t.numel() >= self.SMALL]
        for flat in spans:
            dist.all_reduce(flat)          # SUM; the mean's 1/W goes into Adam's grad_scale
        return dist.get_world_size()


def setup_grad_exchange(D, overlap=True):
    """The data-parallel gradient exchange of a discriminator: returns the reducer to call after the backward, or None
    when the network exchanges everything inside its backward.  ``D_SNDCGAN``: per-layer collectives inside the fused
    backward (all parameters).  ``ResidualDiscriminatorP``: the packed weight gradients are all-reduced from autograd hooks
    as they are produced (heads and the deep levels first -- their exchange hides behind the long high-resolution part of
    the backward), the bias gradients in one packed collective afterwards."""
    if overlap and hasattr(D, 'enable_grad_overlap'):
        rest = D.enable_grad_overlap(OverlappedGradReducer())
        return GradAllReducer(rest) if rest else None
    return GradAllReducer(D.parameters())


class OverlappedGradReducer(object):
    """Gradient exchange overlapped with the backward pass (what DDP's bucketed reducer does for the reference,
    train_gan.py:311-313), at the granularity the fused discriminator backward produces gradients: each layer's
    PACKED weight-gradient slab is all-reduced (SUM, async on RCCL's stream) as soon as its wgrad kernel is enqueued
    -- the three merged head layers (2/3 of D_SNDCGAN's parameters) first, so their exchange hides behind the whole
    trunk backward.  The spectral-norm weight-gradient transform that follows is linear in the packed gradient and
    uses identical weights on every rank, so reducing before it is equivalent to reducing the final gradients."""

    def __init__(self):
        self.handles = []

    def active(self):
        return _dist_on() or (dist.is_available() and dist.is_initialized() and getattr(self, 'force', False))


    def reduce_async(self, t):
        if self.active():
            self.handles.append(dist.all_reduce(t, async_op=True))

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []

    def world(self):
        return dist.get_world_size() if self.active() else 1


def sample_generator(G, num_samples, enable_grad=True):
    """_sample_generator of train_gan.py:96-100."""
    latent = G.sample_latent(num_samples)
    with torch.set_grad_enabled(enable_grad):
        return G(latent)


def set_grad(model, flag=True):
    """utils.set_grad (utils.py:125-127)."""
    for p in model.parameters():
        p.requires_grad = flag


def d_step(P, G, D, opt_D, options, images, reducer=None):
    """One discriminator step, exactly train_gan.py:153-163 (minus the four logging .item() syncs): fakes under
    no_grad -> loss_D_fn -> zero_grad -> backward -> [gradient all-reduce] -> Adam.  Returns (d_loss, aux)."""
    THROTTLE.begin()                          # host stays at most one step ahead of the GPU (hostio.py)
    gen_images = sample_generator(G, images.size(0), enable_grad=False)
    d_loss, aux = P.train_fn["D"](P, D, options, images, gen_images)
    loss = d_loss + aux['penalty']
    opt_D.zero_grad()
    loss.backward()
    comm = getattr(D, '_grad_comm', None)
    if comm is not None:                      # gradients were exchanged inside the backward (overlapped)
        world = comm.world()
    else:
        world = reducer() if reducer is not None else 1
    if world > 1:
        opt_D.step(grad_scale=1.0 / world)
    else:
        opt_D.step()
    THROTTLE.end()
    return d_loss, aux


# ----------------------------------------------------------------------------------------------------------
# StyleGAN2 discriminator steps
# ----------------------------------------------------------------------------------------------------------
def r1_loss(D, images, augment_fn):
    """r1_loss (train_stylegan2.py:106-113): mean_n sum_chw (d D(aug(x)) / d aug(x))^2, differentiable in D's
    parameters (double backward through the HIP op family of contrad_amd.autograd_ops)."""
    from . import autograd_ops as A
    images_aug = augment_fn(images).detach()
    images_aug.requires_grad = True
    d_real = D(images_aug)
    with A.input_grad_only():        # this backward is asked for d / d images only: skip the parameter gradients
        grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=images_aug, create_graph=True, retain_graph=True)
    return A.SumSqMeanFn.apply(grad_real)          # grad_real.pow(2).reshape(N, -1).sum(1).mean(), two launches


def _sg2_fakes(G, N, style_mix):
    """G(G.sample_latent(N), style_mix) (train_stylegan2.py:116-123); contrad_amd's Generator draws everything random of
    that forward in one device launch (Generator.draw_inputs), any other generator goes through the reference API."""
    if hasattr(G, 'draw_inputs'):
        return G(**G.draw_inputs(N, style_mix))
    return G(G.sample_latent(N), style_mix=style_mix)


def d_step_stylegan2(P, G, D, opt_D, options, images, step, reducer=None, style_mix=0.9):
    """D-step of train_stylegan2.py:199-212 (BASELINE config 4): single 3N-image D call via loss_D_fn, plus the
    R1 penalty every ``P.d_reg_every`` steps weighted (0.5*lbd_r1)*r1*d_reg_every (``--no_lazy`` => every step)."""
    THROTTLE.begin()
    with torch.no_grad():
        gen_images = _sg2_fakes(G, images.size(0), style_mix)
    d_loss, aux = P.train_fn["D"](P, D, options, images, gen_images)
    loss = d_loss + aux['penalty']
    if (step % P.d_reg_every == 0) and P.lbd_r1 > 0:
        r1 = r1_loss(D, images, P.augment_fn)
        loss = torch.add(loss, r1, alpha=(0.5 * P.lbd_r1) * P.d_reg_every)      # one launch (and one in the backward)
        aux['r1'] = r1
    opt_D.zero_grad()
    loss.backward()
    world = reducer() if reducer is not None else 1
    opt_D.step(grad_scale=1.0 / world) if world > 1 else opt_D.step()
    THROTTLE.end()
    return d_loss, aux


# One discriminator call over [two real views | fakes] with the minibatch-stddev groups kept inside each segment (True, the
# product path: D.call_merged) or the reference's two calls; the switch serves the same-box A/B runs of bench.py (tools/dev/r5d.sh, profiles/r05_ab_merged_demod.txt).
_MERGED_CALLS = os.environ.get('CONTRAD_DEV_MERGED', '1') != '0'


def loss_D_fn_separate(P, D, options, images, gen_images):
    """ContraD discriminator loss with the call structure of train_stylegan2_contraD.py (G_D.forward :138-164 and
    _loss_D_fn :95-109): the fakes (N) and the two real views (2N) are augmented SEPARATELY and go through D in two
    calls; the losses act on the concatenated embeddings.  Same return contract as contrad.loss_D_fn."""
    from .training.gan.contrad import _ContraDContrastive, _GanDLoss
    N = images.size(0)
    if _MERGED_CALLS and hasattr(D, 'call_merged') and getattr(P.augment_fn, 'supports_out', False):
        # the two calls as ONE pass over 3N images (only the minibatch-stddev statistics see the call boundary), laid out
        # [real views 2N | fakes N] -- the order the losses want -- by letting each augmentation write its slice of one
        # buffer: no concatenation of the batches, no split + re-concatenation of the outputs (forward and backward).
        # Draw order as in the reference: the fakes' augmentation parameters first, then the real views'.
        both = torch.empty((3 * N,) + tuple(images.shape[1:]), device=images.device, dtype=torch.float32)
        P.augment_fn(gen_images.detach(), out=both[2 * N:])
        P.augment_fn(torch.cat([images, images], dim=0), out=both[:2 * N])
        d_all, aux = D.call_merged(both, [2 * N, N], sg_linear=True, projection=True, projection2=True)
        proj, proj2 = aux['projection'], aux['projection2']
    else:
        aug_f = P.augment_fn(gen_images.detach())                   # (same RNG draw order as the two reference calls)
        aug_r = P.augment_fn(torch.cat([images, images], dim=0))
        if hasattr(D, 'call_batches'):
            (d_gen, aux_g), (d_real2, aux_r) = D.call_batches([aug_f, aug_r], sg_linear=True, projection=True,
                                                              projection2=True)
        else:
            d_gen, aux_g = D(aug_f, sg_linear=True, projection=True, projection2=True)
            d_real2, aux_r = D(aug_r, sg_linear=True, projection=True, projection2=True)
        proj = torch.cat([aux_r['projection'], aux_g['projection']], dim=0)
        proj2 = torch.cat([aux_r['projection2'], aux_g['projection2']], dim=0)
        d_all = torch.cat([d_real2, d_gen], dim=0)
    simclr, sup = _ContraDContrastive.apply(proj, proj2, N, P.temp, bool(P.distributed))
    gan, d_real_m, d_gen_m = _GanDLoss.apply(d_all, N, options['loss'])
    return simclr + P.lbd_a * sup, {'penalty': gan, 'd_real': d_real_m, 'd_gen': d_gen_m}


def d_step_stylegan2_contrad(P, G, D, opt_D, options, images, step, reducer=None, style_mix=0.9):
    """D-step of train_stylegan2_contraD.py:148-164,218-226 (BASELINE config 5): loss_D_fn_separate + lazy R1 on
    its own D call."""
    N = images.size(0)
    THROTTLE.begin()
    with torch.no_grad():
        gen_images = _sg2_fakes(G, N, style_mix)
    d_loss, aux = loss_D_fn_separate(P, D, options, images, gen_images)
    loss = d_loss + aux['penalty']
    if (step % P.d_reg_every == 0) and P.lbd_r1 > 0:
        r1 = r1_loss(D, images, P.augment_fn)
        loss = torch.add(loss, r1, alpha=(0.5 * P.lbd_r1) * P.d_reg_every)      # one launch (and one in the backward)
        aux['r1'] = r1
    opt_D.zero_grad()
    loss.backward()
    world = reducer() if reducer is not None else 1
    opt_D.step(grad_scale=1.0 / world) if world > 1 else opt_D.step()
    THROTTLE.end()
    return d_loss, aux


def _quiesce_before_capture(*modules):
    """Called right before a stream capture; returns the ``capture_error_mode`` to capture with.

    With a process group, every EAGER collective issued so far is still listed in ProcessGroupNCCL's watchdog thread until
    that thread has seen it complete (it polls every 100 ms).  The capture pulls RCCL's own stream into capture mode, and
    under the default 'global' capture mode ``hipEventQuery`` from ANY thread is then refused ("operation not permitted on
    an event last recorded in a capturing stream") -- the watchdog rethrows and the whole job aborts.  Seen once in nine
    runs of tests/dist_graph_worker.py, on a cold box where the watchdog had not drained yet.  Deterministic part: with a
    process group the capture runs in 'thread_local' mode, in which only THIS thread's calls are policed, so the watchdog
    thread may keep polling whatever it still holds.  Belt and braces: the device is drained first and the watchdog gets a
    moment to retire the eager works (there is no API to wait for that)."""
    for m in modules:                       # no autograd graph of an eager step may stay referenced across the capture
        if hasattr(m, 'drop_pack_cache'):
            m.drop_pack_cache()
    torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        import time
        time.sleep(1.0)          # (the full second stays until thread_local mode is proven on W > 1 hardware)
        torch.cuda.synchronize()
        return 'thread_local'
    return 'global'


# ----------------------------------------------------------------------------------------------------------
# the SNDCGAN D-step as one hipGraph
# ----------------------------------------------------------------------------------------------------------
class GraphedDStep(object):
    """``d_step`` (train_gan.py:153-163) captured ONCE into a hipGraph and replayed every iteration.

    One D-step is ~120 kernel launches; enqueued from Python they cost ~2.8 ms of host time -- hidden behind 17.7 ms of
    GPU work at batch 512, but most of the 3.5 ms step one rank of an 8-GPU job runs at batch 64.
    Replaying a captured graph takes the host out: per step it only draws the random numbers (latents, augmentation
    parameters -- on the host, in the reference's RNG order, exactly as the eager path), hands them over into STATIC
    device tensors, refreshes three Adam scalars, and launches the graph.  What had to move from launch arguments into
    device memory for that: the colour-op order of the augmentation (column 15 of the parameter block) and Adam's
    step-dependent scalars (``contrad_adam_step_dev``).

    With a process group (one rank of a data-parallel job, train_gan.py:230-318) the collectives are captured WITH the
    step: G's SyncBN statistics all-reduce, the packed embedding all-gather inside the loss and the per-layer gradient
    all-reduces the fused backward overlaps with itself (OverlappedGradReducer: RCCL's own stream forks from and joins the
    capture stream through events, which a stream capture records as graph edges); Adam's 1/W travels in the device
    scalars.  A per-rank batch of 64 is ~3.4 ms of GPU work against ~2.8 ms of Python launch time in the eager path
     -- this is what takes the host off that critical path.  Every rank must replay the same graph
    the same number of times (as every rank must issue the same eager collectives).
    """

    def __init__(self, P, G, D, opt_D, options, images, warmup=3):
        from .augment import SimCLRAugment
        from . import ops
        self.P, self.G, self.D, self.opt, self.options = P, G, D, opt_D, options
        self.aug = P.augment_fn
        if not isinstance(self.aug, SimCLRAugment) or self.aug.p_blur is not None:
            raise NotImplementedError('GraphedDStep: the simclr pipeline')
        self.dist = dist_on()
        self.reducer = None
        if self.dist and getattr(D, '_grad_comm', None) is None:
            self.reducer = GradAllReducer(D.parameters())       # flat collectives after the backward
        self.images = images.clone()                    # static input: load_images() hands over a new real batch
        N = images.size(0)
        dev = images.device
        self.N = N
        for _ in range(warmup if len(opt_D.state) else max(warmup, 1)):     # optimizer state, workspaces, pools
            d_step(P, G, D, opt_D, options, images, self.reducer)
        self.z = torch.zeros(N, G.nz, device=dev)
        self.params = torch.zeros(3 * N, ops.AUG_NPARAM, device=dev)
        self.hyper = torch.ones(3, device=dev)
        torch.cuda.synchronize()          # (capture records launches, it does not run them: the inputs stay untouched)
        G.invalidate_cache()              # the step re-packs G's weights itself: G moves between D-steps in training
        self.graph = torch.cuda.CUDAGraph()
        self._scratch = {}           # this graph's own conv scratch (ops.private_workspace)
        mode = _quiesce_before_capture(self.D)
        if _ops.SEQUENCE is not None:
            _ops.SEQUENCE.append('capture')          # (bench.py --shape-table: the launch order of the captured step follows)
        with _ops.private_workspace(self._scratch), torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.d_loss, self.aux = self._body()
        G.invalidate_cache()              # (what the capture allocated is filled by the first replay, not now)
        torch.cuda.synchronize()

    def load_images(self, images):
        self.images.copy_(images)

    def _refresh_inputs(self):
        from .hostio import upload
        G, N = self.G, self.N
        dev = self.images.device
        z = torch.empty(N, G.nz).uniform_(-1, 1)                        # G.sample_latent's draw (sndcgan.py:50-52)
        Pm, cf, _ = self.aug.sample(3 * N, self.images.shape[2], self.images.shape[3])
        Pm[:, 15] = float(cf)
        self.z.copy_(upload(z, dev))
        self.params.copy_(upload(Pm, dev))
        world = dist.get_world_size() if self.dist else 1
        self.hyper.copy_(upload(torch.tensor([self.opt.hyper_values(1.0 / world)], dtype=torch.float32), dev).view(3))

    def _body(self):
        from . import ops
        from .training.gan.contrad import _ContraDContrastive, _GanDLoss
        P, G, D, N = self.P, self.G, self.D, self.N
        with torch.no_grad():
            gen = G(self.z)
            cat = torch.cat([self.images, self.images, gen], dim=0)
            aug = ops.simclr_augment(cat, self.params, -1, self.aug.r_c is not None)
        d_all, aux = D(aug, sg_linear=True, projection=True, projection2=True)
        simclr, sup = _ContraDContrastive.apply(aux['projection'], aux['projection2'], N, P.temp,
                                                bool(P.distributed) and self.dist)
        gan, d_real, d_gen = _GanDLoss.apply(d_all, N, self.options['loss'])
        d_loss = simclr + P.lbd_a * sup
        self.opt.zero_grad(set_to_none=True)
        (d_loss + gan).backward()
        if self.reducer is not None:
            self.reducer()                      # (the overlapped reducer exchanged inside the backward already)
        self.opt.step_captured(self.hyper)
        # detached: the step's autograd graph must not outlive the capture (its AccumulateGrad nodes are tied to the
        # capture stream; an eager backward later on would have to synchronise with it)
        return d_loss.detach(), {'penalty': gan.detach(), 'd_real': d_real.detach(), 'd_gen': d_gen.detach()}

    def __call__(self):
        THROTTLE.begin()
        self._refresh_inputs()
        self.graph.replay()
        THROTTLE.end()
        return self.d_loss, self.aux


class _StaticAugment(object):
    """Stand-in for ``P.augment_fn`` inside a captured step: the k-th call of a step applies the k-th parameter block.
    ``refresh()`` draws every block on the host with the pipeline's own sampler, in call order (the reference's RNG
    order), and hands it into static device memory; the colour-op order travels in column 15, the Gaussian taps of
    simclr_hq in a static kernel tensor."""

    def __init__(self, aug, sizes, H, W, device):
        from . import ops
        self.aug, self.sizes, self.H, self.W, self.dev = aug, list(sizes), H, W, device
        self.blocks = [torch.zeros(n, ops.AUG_NPARAM, device=device) for n in self.sizes]
        self.radius = int((H // 10) / 2) if aug.p_blur is not None else 0
        self.taps = [torch.zeros(2 * self.radius + 1, device=device) for _ in self.sizes] if self.radius > 0 else None
        self.k = 0

    def refresh(self):
        from .hostio import upload
        for i, n in enumerate(self.sizes):
            Pm, _cf, sigma = self.aug.sample(n, self.H, self.W)
            self.blocks[i].copy_(upload(Pm, self.dev))
            if self.taps is not None:
                _r, g = self.aug.blur_kernel(self.H, sigma)
                self.taps[i].copy_(upload(g.view(1, -1), self.dev).view(-1))

    supports_out = True          # (``out=``: the last stage writes into the caller's buffer, see SimCLRAugment.apply)

    def __call__(self, x, out=None):
        from . import ops
        i, self.k = self.k, self.k + 1
        if x.shape[0] != self.sizes[i]:
            raise RuntimeError('captured step: augmentation call %d saw %d images, planned %d' % (i, x.shape[0], self.sizes[i]))
        if x.requires_grad and torch.is_grad_enabled():      # generator step: gradient flows through the augmentation
            from .augment import _SimCLRFn
            if out is not None:
                raise RuntimeError('augmentation: out= is a forward-only option')
            return _SimCLRFn.apply(x.contiguous().float(), self.blocks[i], -1, self.aug.r_c is not None,
                                   (self.radius, self.taps[i]) if self.taps is not None else None,
                                   self.aug.cutout_length if self.aug.p_cutout is not None else None)
        res = ops.simclr_augment(x.detach().contiguous().float(), self.blocks[i], -1, self.aug.r_c is not None,
                                 out=None if self.taps is not None else out)
        if self.taps is not None:
            res = ops.gaussian_blur_masked(res, self.blocks[i], self.taps[i], self.radius, out=out)
        if self.aug.p_cutout is not None:
            ops.cutout_masked_(res, self.blocks[i], self.aug.cutout_length)
        return res


class _StaticSG2Inputs(object):
    """Everything random the StyleGAN2 generator consumes in one forward, as static device tensors: latents, mixing
    latents, mixing layers, per-layer noise.  ``refresh()`` draws them in the forward's own order (device draws for the
    latents and the noise, CPU-generator draws for the mixing mask, generator.py:233-234,252-259,91-92)."""

    def __init__(self, G, N, device, style_mix):
        self.G, self.N, self.dev, self.style_mix = G, N, device, style_mix
        self.flat = torch.zeros(sum(G.input_sizes(N)), device=device)        # z | mixing latent | per-layer noise
        self.mix_layer = torch.zeros(N, device=device)
        self.mixing = G.training and style_mix > 0
        z, z_mix, noise = G.input_views(self.flat, N)
        self.kw = dict(input=z, style_mix=style_mix, noise=noise)
        if self.mixing:
            self.kw['_mix'] = (z_mix, self.mix_layer)

    def refresh(self):
        """Generator.draw_inputs' draws, in its order, into the static tensors: one device normal_(), then the CPU draws."""
        from .hostio import upload
        G, N = self.G, self.N
        self.flat.normal_()
        if self.mixing:
            nomix = torch.rand(N) >= self.style_mix
            mix_layer = torch.randint(G.n_latent, (N,)).masked_fill(nomix, G.n_latent)
            self.mix_layer.copy_(upload(mix_layer.float().view(-1, 1), self.dev).view(-1))

    def forward(self):
        return self.G(**self.kw)


class GraphedSG2DStep(object):
    """The StyleGAN2 D-steps (``d_step_stylegan2`` / ``d_step_stylegan2_contrad``) as ONE captured hipGraph.

    These steps are 500-800 launches of mostly small kernels (the R1 double backward alone builds ~300 nodes); enqueued
    from Python they cost 19-27 ms of host time per step depending on the host core -- the same order as the 19 ms of GPU
    work of BASELINE config 4, which therefore flipped between GPU-bound (20.5 ms) and host-bound (27 ms) from run to run.
    Everything random is drawn OUTSIDE the graph, in the eager path's order -- device draws (latents, mixing latents,
    per-layer noise) into static tensors, host draws (mixing layers, augmentation blocks, blur taps) handed over -- so a
    replay consumes exactly the random numbers the eager step would: ``tests/test_graph_gpu.py`` checks bitwise equality.
    With lazy R1 (``d_reg_every`` > 1) the graph holds the plain step and the R1 steps run eagerly."""

    def __init__(self, P, G, D, opt_D, options, images, contrad_script, style_mix=0.9, warmup=2):
        import argparse
        self.dist = dist_on()       # one rank of a data-parallel job: the embedding all-gather (inside the loss) and the
        self.reducer = None         # gradient all-reduces (from the backward's hooks and / or after it) are captured
        if self.dist:
            rest = D.overlap_rest() if getattr(D, '_pack_comm', None) is not None else None
            self.reducer = GradAllReducer(rest if rest is not None else D.parameters())
        self.P, self.G, self.D, self.opt, self.options, self.images = P, G, D, opt_D, options, images.clone()
        self.contrad_script, self.style_mix = contrad_script, style_mix
        self.eager = d_step_stylegan2_contrad if contrad_script else d_step_stylegan2
        N, dev = images.size(0), images.device
        self.N = N
        self.r1_in_graph = P.lbd_r1 > 0 and P.d_reg_every == 1
        for s in range(1, (warmup if len(opt_D.state) else max(warmup, 1)) + 1):     # optimizer state, workspaces
            self.eager(P, G, D, opt_D, options, images, s if P.d_reg_every == 1 else 1, self.reducer, style_mix)
        H, W = images.shape[2], images.shape[3]
        sizes = ([N, 2 * N] if contrad_script else [3 * N]) + ([N] if self.r1_in_graph else [])
        self.saug = _StaticAugment(P.augment_fn, sizes, H, W, dev)
        self.Pg = argparse.Namespace(**vars(P))
        self.Pg.augment_fn = self.saug
        self.gin = _StaticSG2Inputs(G, N, dev, style_mix)
        self.hyper = torch.ones(3, device=dev)
        torch.cuda.synchronize()
        G.invalidate_cache()              # as GraphedDStep: the packed tables are rebuilt inside the step
        self.graph = torch.cuda.CUDAGraph()
        self._scratch = {}           # this graph's own conv scratch (ops.private_workspace)
        mode = _quiesce_before_capture(self.D)
        if _ops.SEQUENCE is not None:
            _ops.SEQUENCE.append('capture')          # (bench.py --shape-table: the launch order of the captured step follows)
        with _ops.private_workspace(self._scratch), torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.d_loss, self.aux = self._body()
        G.invalidate_cache()
        torch.cuda.synchronize()

    def load_images(self, images):
        self.images.copy_(images)

    def _refresh(self):
        from .hostio import upload
        dev = self.images.device
        self.gin.refresh()
        self.saug.refresh()
        world = dist.get_world_size() if self.dist else 1
        self.hyper.copy_(upload(torch.tensor([self.opt.hyper_values(1.0 / world)], dtype=torch.float32), dev).view(3))

    def _body(self):
        P, G, D = self.Pg, self.G, self.D
        self.saug.k = 0
        with torch.no_grad():
            gen = self.gin.forward()
        if self.contrad_script:
            d_loss, aux = loss_D_fn_separate(P, D, self.options, self.images, gen)
        else:
            d_loss, aux = P.train_fn["D"](P, D, self.options, self.images, gen)
        loss = d_loss + aux['penalty']
        if self.r1_in_graph:
            r1 = r1_loss(D, self.images, self.saug)
            loss = torch.add(loss, r1, alpha=(0.5 * P.lbd_r1) * P.d_reg_every)     # (as the eager step)
            aux['r1'] = r1
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.reducer is not None:
            self.reducer()
        self.opt.step_captured(self.hyper)
        return d_loss.detach(), {k: v.detach() for k, v in aux.items()}     # (see GraphedDStep._body)

    def __call__(self, step):
        P = self.P
        if P.lbd_r1 > 0 and P.d_reg_every > 1 and step % P.d_reg_every == 0:      # lazy-R1 step: eager
            return self.eager(P, self.G, self.D, self.opt, self.options, self.images, step, self.reducer, self.style_mix)
        THROTTLE.begin()
        self._refresh()
        self.graph.replay()
        THROTTLE.end()
        return self.d_loss, self.aux


# ----------------------------------------------------------------------------------------------------------
# the generator steps as captured hipGraphs (the other half of a training iteration)
# ----------------------------------------------------------------------------------------------------------
def _hyper_upload(opt, hyper):
    from .hostio import upload
    world = dist.get_world_size() if dist_on() else 1              # Adam's 1/W of the data-parallel mean
    hyper.copy_(upload(torch.tensor([opt.hyper_values(1.0 / world)], dtype=torch.float32), hyper.device).view(3))


class GraphedGStep(object):
    """The SNDCGAN generator step (train_gan.py:170-179: fakes WITH gradient -> ``loss_G_fn`` = D(augment(G(z))) ->
    backward through D, the augmentation and G -> Adam on G) as one captured hipGraph.  The caller has set
    ``set_grad(G, True); set_grad(D, False)`` and both networks to train mode, and ``opt_G`` holds state (one eager step
    has run).  Per replay the host draws the latents and the augmentation block in the eager order."""

    def __init__(self, P, G, D, opt_G, options, N, H, W):
        import argparse
        from . import ops
        from .augment import SimCLRAugment
        if not isinstance(P.augment_fn, SimCLRAugment):
            raise NotImplementedError('GraphedGStep: simclr-family pipeline')
        self.dist = dist_on()      # data-parallel rank: SyncBN's statistics exchange and G's gradient all-reduce are captured
        self.reducer = GradAllReducer(G.parameters()) if self.dist else None
        if not len(opt_G.state):
            raise RuntimeError('GraphedGStep: capture after the first eager generator step (Adam state)')
        self.P, self.G, self.D, self.opt, self.options, self.N = P, G, D, opt_G, options, N
        dev = next(G.parameters()).device
        self.saug = _StaticAugment(P.augment_fn, [N], H, W, dev)
        self.Pg = argparse.Namespace(**vars(P))
        self.Pg.augment_fn = self.saug
        self.z = torch.zeros(N, G.nz, device=dev)
        self.hyper = torch.ones(3, device=dev)
        torch.cuda.synchronize()
        G.invalidate_cache()
        self.graph = torch.cuda.CUDAGraph()
        self._scratch = {}           # this graph's own conv scratch (ops.private_workspace)
        mode = _quiesce_before_capture(self.D)
        if _ops.SEQUENCE is not None:
            _ops.SEQUENCE.append('capture')          # (bench.py --shape-table: the launch order of the captured step follows)
        with _ops.private_workspace(self._scratch), torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.g_loss = self._body()
        G.invalidate_cache()
        torch.cuda.synchronize()

    def _body(self):
        self.saug.k = 0
        gen = self.G(self.z)
        g_loss = self.Pg.train_fn["G"](self.Pg, self.D, self.options, None, gen)
        self.opt.zero_grad(set_to_none=True)
        g_loss.backward()
        if self.reducer is not None:
            self.reducer()
        self.opt.step_captured(self.hyper)
        return g_loss.detach()

    def __call__(self):
        from .hostio import upload
        THROTTLE.begin()
        z = torch.empty(self.N, self.G.nz).uniform_(-1, 1)             # G.sample_latent's draw (sndcgan.py:50-52)
        self.z.copy_(upload(z, self.z.device))
        self.saug.refresh()
        _hyper_upload(self.opt, self.hyper)
        self.graph.replay()
        THROTTLE.end()
        return self.g_loss


class GraphedSG2GStep(object):
    """The StyleGAN2 generator step of train_stylegan2_contraD.py:138-146,207 (D with ``sg_linear=False`` and both
    projections on the augmented fakes, non-saturating loss) as one captured hipGraph; same contract as GraphedGStep."""

    def __init__(self, P, G, D, opt_G, options, N, H, W, style_mix=0.9):
        self.dist = dist_on()
        self.reducer = GradAllReducer(G.parameters()) if self.dist else None
        if not len(opt_G.state):
            raise RuntimeError('GraphedSG2GStep: capture after the first eager generator step (Adam state)')
        self.P, self.G, self.D, self.opt, self.options, self.N = P, G, D, opt_G, options, N
        dev = next(G.parameters()).device
        self.saug = _StaticAugment(P.augment_fn, [N], H, W, dev)
        self.gin = _StaticSG2Inputs(G, N, dev, style_mix)
        self.hyper = torch.ones(3, device=dev)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._scratch = {}           # this graph's own conv scratch (ops.private_workspace)
        mode = _quiesce_before_capture(self.D)
        if _ops.SEQUENCE is not None:
            _ops.SEQUENCE.append('capture')          # (bench.py --shape-table: the launch order of the captured step follows)
        with _ops.private_workspace(self._scratch), torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.g_loss = self._body()
        G.invalidate_cache()
        torch.cuda.synchronize()

    def _body(self):
        from .training.gan.contrad import _GanGLoss
        self.saug.k = 0
        gen = self.gin.forward()
        d_gen, _aux = self.D(self.saug(gen), sg_linear=False, projection=True, projection2=True)
        g_loss = _GanGLoss.apply(d_gen, 'nonsat')
        self.opt.zero_grad(set_to_none=True)
        g_loss.backward()
        if self.reducer is not None:
            self.reducer()
        self.opt.step_captured(self.hyper)
        return g_loss.detach()

    def __call__(self):
        THROTTLE.begin()
        self.gin.refresh()
        self.saug.refresh()
        _hyper_upload(self.opt, self.hyper)
        self.graph.replay()
        THROTTLE.end()
        return self.g_loss
