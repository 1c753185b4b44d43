#!/usr/bin/env python
"""Headline benchmark: discriminator-step images/sec (ContraD, SimCLR aug) on synthetic batches, random-init weights.

    python bench.py [--gpus N --steps K --warmup W] [--config c10_b512|sg2_32|sg2_512|all]

``--gpus N`` (N > 1) spawns one process per GPU itself (re-exec under ``torch.distributed.run`` on 127.0.0.1, as the
reference's ``mp.spawn`` does, train_gan.py:331) unless it already runs inside such a launch (WORLD_SIZE set).  With N > 1
every workload is first timed with eager launches; capture + replay of the step with its RCCL collectives comes second,
under a watchdog: the line always carries a number for every workload.

Workloads (BASELINE.json configs):
  c10_b512  SNDCGAN + ContraD, CIFAR-10 32x32, GLOBAL batch 512, simclr aug (configs[1]; [2] = the same over N GPUs)
            -- the configuration the metric is quoted on: the top-level fields of the JSON line.  "strong" scaling.
  sg2_32    StyleGAN2 small32 + ContraD, 32x32, batch 64 per GPU, R1 every step (train_stylegan2.py --no_lazy) (configs[3])
  sg2_512   StyleGAN2_512 + ContraD, 512x512, batch 16 per GPU, simclr_hq aug, lazy R1 every 16th step
            (train_stylegan2_contraD.py semantics: separate N and 2N discriminator calls) (configs[4])
With ``--config all`` (the default) the two StyleGAN2 workloads run after the headline one and are reported under
``"other_configs"`` of the same single JSON line, each with its own ``roofline`` (and ``cpu_baseline`` at N=1).

One "step" = one full D-step of the reference loop (train_gan.py:153-163): no-grad G forward for N fakes ->
SimCLR-augment 3N images -> D forward -> NT-Xent + SupCon + non-saturating GAN loss [+ R1] -> backward -> [embedding
all-gather / gradient all-reduce over RCCL] -> Adam on D.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import re
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA = 157.3         # TFLOP/s, MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)

# flop_per_image: SURVEY.md 8(d) -- algorithmic FLOPs of one D-step per real image (FlopCounterMode on the reference
# path), R1 amortised over its period.
CONFIGS = {
    'c10_b512': dict(arch='sndcgan', size=32, batch=512, batch_is_global=True, aug='simclr',
                     gin=('gan', 'cifar10', 'c10_b512.gin'), flop_per_image=4.28e9, g_flop_per_image=1.53e9, d_reg_every=0, lbd_r1=0.0,
                     steps=20, warmup=5,
                     workload="SNDCGAN + ContraD D-step, CIFAR-10 32x32, global batch %d, simclr aug, nonsat loss, "
                              "Adam(2e-4,(0.5,0.999)), random-init weights"),
    'sg2_32': dict(arch='stylegan2', size=32, batch=64, batch_is_global=False, aug='simclr',
                   gin=('gan', 'stylegan2', 'c10_style64.gin'), flop_per_image=14.29e9 + 8.6e9, g_flop_per_image=16.5e9, d_reg_every=1,
                   lbd_r1=0.1, steps=10, warmup=3,
                   workload="StyleGAN2(small32) + ContraD D-step, 32x32, batch %d per GPU, simclr aug, R1 every step "
                            "(--no_lazy, lbd_r1 0.1), single 3N discriminator call (train_stylegan2.py), "
                            "Adam(2e-3,(0,0.99)), random-init weights"),
    'sg2_512': dict(arch='stylegan2_512', size=512, batch=16, batch_is_global=False, aug='simclr_hq',
                    gin=('gan', 'stylegan2', 'afhq_dog_style64.gin'), flop_per_image=388.7e9 + 235e9 / 16,
                    flop_plain=388.7e9, flop_r1=235e9, g_flop_per_image=302.5e9,
                    d_reg_every=16, lbd_r1=0.5, steps=16, warmup=3,
                    workload="StyleGAN2_512 (channel multiplier 1) + ContraD D-step, AFHQ-shaped 512x512, batch %d per "
                             "GPU, simclr_hq aug (crop 0.08-1, jitter 0.8/0.8/0.8/0.2, gaussian blur k=51), lazy R1 every "
                             "16th step (lbd_r1 0.5), separate N and 2N discriminator calls "
                             "(train_stylegan2_contraD.py), Adam(2.5e-3,(0,0.99)), random-init weights"),
}


def _threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 32))      # the small convs stop scaling (and oversubscribe) beyond that


def _time_cpu(step, budget_s=20.0, max_steps=5):
    t0 = time.perf_counter()
    step(1)
    warm = time.perf_counter() - t0
    if warm > budget_s:                    # one step already exceeds the sample budget: it IS the sample
        return warm, 1
    steps = max(1, min(max_steps, int(budget_s / max(warm, 1e-3))))      # bound the sample to ~20 s of CPU work
    t0 = time.perf_counter()
    for t in range(steps):
        step(t + 2)
    return (time.perf_counter() - t0) / steps, steps


def cpu_baseline_c10(n=64):
    """BASELINE.json configs[0] on the host cores: the oracle's (= reference algorithm's) PyTorch-CPU D-step."""
    from oracle import contrad_oracle as O
    torch.manual_seed(0); np.random.seed(0)
    torch.set_num_threads(_threads())
    sd = O.det_fill(O.sndcgan_d_param_shapes(), seed=1)
    gsd = O.det_fill(O.sndcgan_g_param_shapes(), seed=2)
    params = [k for k in sd if k.endswith('weight_orig') or k.endswith('bias')]
    for k in params:
        sd[k].requires_grad_()
    m = {k: torch.zeros_like(sd[k]) for k in params}
    v = {k: torch.zeros_like(sd[k]) for k in params}
    x = torch.rand(n, 3, 32, 32)

    def step(t):
        with torch.no_grad():
            fake = O.sndcgan_g_forward(gsd, O.sample_latent_sndcgan(n))
        p = O.sample_simclr_params(3 * n, 32, 32, O.SIMCLR_CIFAR)
        aug = O.simclr_apply(torch.cat([x, x, fake]), p)
        closs, gloss, _, _ = O.contrad_loss_d(lambda z: O.sndcgan_d_forward(sd, z, sg_linear=True)[:3], aug, n)
        for k in params:
            sd[k].grad = None
        (closs + gloss).backward()
        with torch.no_grad():
            for k in params:
                O.adam_step(sd[k], sd[k].grad, m[k], v[k], t, 2e-4, 0.5, 0.999)

    dt, steps = _time_cpu(step)
    return {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "SNDCGAN ContraD D-step, 32x32, batch %d, %d timed steps after 1 warm-up "
                      "(oracle = PyTorch-CPU restatement of the reference path, %.3f s/step)" % (n, steps, dt)}


def cpu_baseline_sg2(name, n):
    """The oracle's StyleGAN2 D-step on the host cores -- the same step the GPU side times: generator forward under
    no_grad (fresh latents, per-layer noise and style mixing each step), augmentation, D forward / backward, losses
    [+ R1], Adam."""
    from oracle import contrad_oracle as O
    from oracle import stylegan2_oracle as S
    cfg = CONFIGS[name]
    size = cfg['size']
    small32 = cfg['arch'] == 'stylegan2'
    cm = 2 if small32 else 1
    torch.manual_seed(0); np.random.seed(0)
    torch.set_num_threads(_threads())
    sd = S.det_fill_d(S.d_param_shapes(size, small32, cm))
    params = [k for k in sd if not k.endswith('kernel')]
    for k in params:
        sd[k].requires_grad_()
    m = {k: torch.zeros_like(sd[k]) for k in params}
    v = {k: torch.zeros_like(sd[k]) for k in params}
    x = torch.rand(n, 3, size, size)
    gshapes = S.g_param_shapes(size, small32, cm)
    gsd = S.fill_kernels(S.det_fill_g(gshapes), gshapes)
    log_size = int(np.log2(size))
    n_latent, n_noise = 2 * log_size - 2, 2 * log_size - 3
    aug_cfg = O.SIMCLR_CIFAR if name == 'sg2_32' else O.SIMCLR_HQ_AFHQ
    lr = 2e-3 if small32 else 2.5e-3
    every = cfg['d_reg_every']

    def d3(z):
        if name == 'sg2_32':                   # train_stylegan2.py: one 3N call
            return S.d_forward(sd, z, size, sg_linear=True)[:3]
        # train_stylegan2_contraD.py: the two real views (2N) and the fakes (N) in separate calls (the minibatch-stddev
        # groups then live inside each call); outputs re-assembled in [view1, view2, fakes] order
        r = S.d_forward(sd, z[:2 * n], size, sg_linear=True)[:3]
        f = S.d_forward(sd, z[2 * n:], size, sg_linear=True)[:3]
        return tuple(torch.cat([a, b]) for a, b in zip(r, f))

    def step(t):
        with torch.no_grad():             # Generator.forward in train mode, style_mix 0.9 (generator.py:236-291)
            noise = [torch.randn(n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)) for i in range(n_noise)]
            mix_layer = torch.where(torch.rand(n) < 0.9, torch.randint(1, n_latent, (n,)), torch.full((n,), n_latent))
            fake = S.g_forward(gsd, torch.randn(n, 512), size, noise, mix=(torch.randn(n, 512), mix_layer))
        if name == 'sg2_32':
            aug = O.simclr_apply(torch.cat([x, x, fake]), O.sample_simclr_params(3 * n, size, size, aug_cfg))
        else:
            aug_f = O.simclr_apply(fake, O.sample_simclr_params(n, size, size, aug_cfg))
            aug_r = O.simclr_apply(torch.cat([x, x]), O.sample_simclr_params(2 * n, size, size, aug_cfg))
            aug = torch.cat([aug_r, aug_f])
        closs, gloss, _, _ = O.contrad_loss_d(d3, aug, n)
        loss = closs + gloss
        if t % every == 0 or name == 'sg2_32':
            pr = O.sample_simclr_params(n, size, size, aug_cfg)
            r1 = S.r1_penalty(lambda z: S.d_forward(sd, z, size)[0], O.simclr_apply(x, pr).detach())
            loss = loss + 0.5 * cfg['lbd_r1'] * r1 * every
        for k in params:
            sd[k].grad = None
        loss.backward()
        with torch.no_grad():
            for k in params:
                O.adam_step(sd[k], sd[k].grad, m[k], v[k], t, lr, 0.0, 0.99)

    dt, steps = _time_cpu(step, max_steps=3)
    return {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s ContraD D-step (G forward no-grad + augment + D fwd/bwd + losses%s + Adam), %dx%d, batch %d, "
                      "%d timed steps after 1 warm-up (oracle = PyTorch-CPU restatement, "
                      "%.3f s/step)" % (cfg['arch'], ' + R1 every step' if name == 'sg2_32' else ', no R1 step in the sample',
                                        size, size, n, steps, dt)}


class _GraphWatchdog(object):
    """Per-rank deadline around the graph capture + replay of one workload (world size > 1 only; main() has the order)."""

    def __init__(self, rank, timeout, results, emit):
        self.rank, self.timeout, self.results, self.emit = rank, timeout, results, emit
        self.kept, self.timer = {}, None

    def keep(self, name, out):               # rank 0: the eager result of the workload that is about to capture
        self.kept[name] = out

    def arm(self, name):
        import threading
        self.timer = threading.Timer(self.timeout, self._fire, args=(name,))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self, name):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None
        self.kept.pop(name, None)

    def _fire(self, name):
        sys.stderr.write('bench.py: rank %d: %s did not get through graph capture + replay within %.0f s; reporting the '
                         'eager result\n' % (self.rank, name, self.timeout))
        sys.stderr.flush()
        if self.rank == 0 and name in self.kept:
            self.results[name] = self.kept[name]
            self.emit()
        sys.stdout.flush()
        os._exit(0)


def _gloo_row_gather():
    """--dev-backend gloo: gloo has no all_gather_into_tensor for device tensors; the list form gives the same rows."""
    import contrad_amd.third_party.gather_layer as gl
    import contrad_amd.training.gan.contrad as cd

    def gather_rows(x):
        outs = [torch.empty_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, x.contiguous())
        return torch.stack(outs, 0)
    gl.all_gather_rows = gather_rows
    cd.all_gather_rows = gather_rows


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pmc_traffic(config, kernel):
    """HBM traffic per launch of ``kernel`` from the newest committed rocprofv3 --pmc summary of this config
    (profiles/rNN_<config>_n1_pmc.json; separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)."""
    prof = os.path.join(ROOT, 'profiles')
    cands = []
    for f in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        if f.endswith('_pmc.json') and (('_%s_' % config) in f or (config == 'c10_b512' and '_bench_n1_' in f)):
            cands.append(f)
    for f in reversed(cands):
        try:
            pmc = json.load(open(os.path.join(prof, f)))
            ent = pmc.get('kernels', {}).get(kernel)
            if ent is None:
                # rocprofv3 prints igemm_lean_kernel<MODE, BM, BN, false|true> (4th argument: balanced strided order);
                # the engine names an instance by its first three
                want = kernel.replace(' ', '')
                for k, v in pmc.get('kernels', {}).items():
                    kk = re.sub(r'wino(22|23|44n?)?::', '', k.replace(' ', '').replace('(anonymousnamespace)::', ''))
                    kk = re.sub(r'^(wino(?:22|23|44n?)_kernel<\d+),\d+>', r'\1>', kk)
                    kk = re.sub(r'^wino23_kernel<\d+>', 'wino23_kernel', kk)
                    if kk == want or kk == want[:-1] + ',false>':
                        ent = v
                        break
            if ent is not None:
                if pmc.get('csrc_fingerprint') != _csrc_fingerprint():
                    # counters cannot be read inside the timed process; a summary collected on OTHER kernel sources is
                    # not this build's traffic -> null, and say why
                    return None, 'stale: profiles/%s was collected on kernel sources %s, this build is %s' % (
                        f, pmc.get('csrc_fingerprint'), _csrc_fingerprint()), None
                return ent['traffic_bytes_per_launch'], 'profiles/' + f, ent
        except (OSError, ValueError, KeyError):
            pass
    return None, None, None


def _write_shape_table(path, name, cfg, warm_prof, marks, n_local):
    """One row per (kernel instance, layer shape) of the conv engine, from the event-bracketed eager warm-up steps:
    launches per step, algorithmic GFLOP per launch (2*N*Ho*Wo*K*C*KH*KW), workgroups of the main launch -- the key
    tools/rocpd_rows.py joins a rocprofv3 kernel trace on (the trace names only the template instance).  With lazy R1 the
    first warm-up step is the R1 step: its launches go into their own section."""
    nsteps = len(marks) - 1
    if nsteps <= 0:
        return
    lazy = cfg['d_reg_every'] > 1
    sections = {}
    for i in range(nsteps):
        sec = 'r1_step' if (lazy and i == 0) else 'plain_step'
        if not lazy and nsteps > 1 and i == 0:
            continue                                   # first eager step: cold caches
        d = sections.setdefault(sec, {'steps': 0, 'rows': {}})
        d['steps'] += 1
        # the launch ORDER of one step: rocpd_rows.py aligns the trace's igemm dispatches of a step with it
        d['sequence'] = [[q[0], list(q[4]), q[5], round(q[1] / 1e9, 4), round(q[6], 4)] for q in warm_prof[marks[i]:marks[i + 1]]]
        for kname, flops, e0, e1, shape, blocks, _executed in warm_prof[marks[i]:marks[i + 1]]:
            r = d['rows'].setdefault((kname, shape, blocks), [0, 0.0, flops])
            r[0] += 1
            r[1] += e0.elapsed_time(e1)
    out = {'config': name, 'per_gpu_batch': n_local, 'flop_rule': '2*N*Ho*Wo*K*C*KH*KW per launch',
           'shape_fields': ['N', 'H', 'W', 'C', 'K', 'KH', 'KW', 'stride', 'pad'],
           'note': 'bracket_us = HIP events around the C-ABI call in eager warm-up steps (a WGRAD bracket spans the '
                   'main kernel and its reduce kernel); grid_blocks = workgroups of the main igemm launch', 'sections': {}}
    for sec, d in sections.items():
        rows = []
        for (kname, shape, blocks), (cnt, ms, flops) in d['rows'].items():
            rows.append({'kernel': kname, 'shape': list(shape), 'grid_blocks': blocks,
                         'launches_per_step': cnt / d['steps'], 'gflop_per_launch': round(flops / 1e9, 4),
                         'bracket_us': round(ms / cnt * 1e3, 2), 'bracket_tflops': round(flops * cnt / (ms * 1e-3) / 1e12, 2)})
        rows.sort(key=lambda r: -r['gflop_per_launch'] * r['launches_per_step'])
        out['sections'][sec] = {'steps_sampled': d['steps'], 'rows': rows, 'sequence': d['sequence'],
                                'sequence_fields': ['kernel', 'shape', 'grid_blocks', 'gflop', 'executed_share']}
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)


def _csrc_fingerprint():
    """sha256 over the kernel sources: a committed PMC summary names the fingerprint it was collected with, so a summary
    that predates a kernel change is recognised as stale instead of being quoted."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'contrad_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def run_config(name, args, world, rank, dev, multi, keep=None, arm=None, disarm=None, graph=None, eager_line=None,
               overlap=None):
    from contrad_amd import config, ops
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import (GradAllReducer, OverlappedGradReducer, d_step, d_step_stylegan2,
                                    d_step_stylegan2_contrad, set_grad)
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup

    cfg = CONFIGS[name]
    steps = args.steps if args.steps is not None else cfg['steps']
    warmup = args.warmup if args.warmup is not None else cfg['warmup']
    if cfg['d_reg_every'] > 1 and steps % cfg['d_reg_every']:
        # lazy R1 (a side workload, never the top-level line): a window that is not a multiple of the period holds the wrong
        # share of R1 steps (20 steps: 1 / 20 instead of 1 / 16, 0.7 % optimistic -- VERDICT r5): round the window UP
        steps = -(-steps // cfg['d_reg_every']) * cfg['d_reg_every']
    batch = args.dev_local_batch or cfg['batch']
    if cfg['batch_is_global']:
        assert batch % world == 0
        n_local, global_batch, scaling = batch // world, batch, 'strong'          # train_gan.py:247
    else:
        n_local, global_batch, scaling = batch, batch * world, 'weak'
    size = cfg['size']

    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, *cfg['gin'])])
    opt = config.get_bindings('options')
    torch.manual_seed(0); np.random.seed(0)                   # identical weights on every rank (DDP's broadcast)
    G, D = get_architecture(cfg['arch'], (size, size, 3))
    torch.manual_seed(0 + rank); np.random.seed(0 + rank)
    G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(mode='contrad', aug=cfg['aug'], temp=0.1, lbd_a=1.0, distributed=multi,
                           lbd_r1=cfg['lbd_r1'], d_reg_every=max(cfg['d_reg_every'], 1))
    P = setup(P)
    P.augment_fn = get_augment(mode=P.aug).to(dev)
    options = {'loss': opt['loss'], 'batch_size': n_local}
    opt_D = FusedAdam(D.parameters(), lr=opt['lr'], betas=tuple(opt['beta']))
    reducer = None
    if multi:
        from contrad_amd.engine import setup_grad_exchange
        # per-layer collectives hidden behind the backward (D_SNDCGAN: everything; ResidualDiscriminatorP: the packed
        # weight gradients, the biases in one packed collective afterwards); CONTRAD_NO_OVERLAP: flat collectives after it
        if overlap is None:
            overlap = not os.environ.get('CONTRAD_NO_OVERLAP')
        reducer = setup_grad_exchange(D, overlap=overlap)
    set_grad(G, False); set_grad(D, True)
    images = torch.rand(n_local, 3, size, size, device=dev)     # synthetic batch, resident in HBM

    # lazy R1: the first warm-up step IS an R1 step, so that its ~25 GB of double-backward buffers sit in the caching
    # allocator before the timed window (a cold hipMalloc of that size inside the window cost up to 0.5 s on a busy box)
    counter = [cfg['d_reg_every'] - 1 if cfg['d_reg_every'] > 1 else 0]
    graphed = [None]
    # hipGraph replay also with a process group: the collectives are captured with the step (engine.GraphedDStep)
    use_graph = (graph or args.graph) != 'off'
    graph_failed = False
    if name == 'c10_b512':
        def one_step():
            if graphed[0] is not None:
                return graphed[0]()
            return d_step(P, G, D, opt_D, options, images, reducer)
    else:
        fn = d_step_stylegan2 if name == 'sg2_32' else d_step_stylegan2_contrad

        def one_step():
            counter[0] += 1
            if graphed[0] is not None:
                return graphed[0](counter[0])
            return fn(P, G, D, opt_D, options, images, counter[0], reducer)

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up: every conv-engine launch is bracketed by HIP events (on the launch stream) -> per-kernel table and the
    # choice of the dominant kernel.  Timed region: only the dominant kernel is bracketed -- an event pair costs
    # ~10 us of stream time and ~60 of them per step would take 6 % off the number being measured.
    ops.PROFILE = []
    marks = [0]
    for _ in range(warmup):
        one_step()
        marks.append(len(ops.PROFILE))
    torch.cuda.synchronize()
    warm_prof = ops.PROFILE
    ops.PROFILE = []
    wsteps = max(1, warmup // 2)                                     # the later half of the warm-up (clocks ramped)
    wagg = {}
    for kname, flops, e0, e1, _shape, _blocks, executed in warm_prof[marks[max(0, warmup - wsteps)]:]:
        a = wagg.setdefault(kname, [0.0, 0.0, 0, 0.0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += flops
        a[2] += 1
        a[3] += flops * executed
    if args.shape_table and rank == 0:
        _write_shape_table(args.shape_table, name, cfg, warm_prof, marks, n_local)
    del warm_prof
    dom_name = max(wagg.items(), key=lambda kv: kv[1][0])[0] if wagg else None
    ops.PROFILE_ONLY = dom_name
    peak_box = [None]                     # peak HBM of the D-step (read before the generator-step side measurement)

    def timed_region():
        """EXACTLY `steps` steps between two barrier + synchronize brackets; MAX over ranks."""
        if cfg['d_reg_every'] > 1:      # lazy R1: start right after an R1 step, so K steps hold exactly K // period of them
            counter[0] = 0
        barrier()
        t0 = time.perf_counter()
        if os.environ.get('CONTRAD_BENCH_STEP_TIMES'):          # dev: per-step wall times (synchronises every step)
            for i in range(steps):
                ts = time.perf_counter()
                d_loss, aux = one_step()
                torch.cuda.synchronize()
                sys.stderr.write('%s step %d: %.2f ms\n' % (name, i + 1, (time.perf_counter() - ts) * 1e3))
        else:
            for _ in range(steps):
                d_loss, aux = one_step()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        finite = bool(torch.isfinite(d_loss).item() and torch.isfinite(aux['penalty']).item())
        return dt, finite

    def make_out(dt, finite, prof, nprof_steps, launch, dom, wagg_, wsteps_):
        """This workload's JSON object from one timed region (rank 0 only)."""
        ms = dt / steps * 1e3
        value = global_batch * steps / dt
        graph_run = launch == 'hipGraph replay'
        # dominant kernel = the conv-engine instance with the largest summed device time (chosen on the warm-up
        # steps); its launches in the timed region are the `achieved` figure
        if dom is None:                                             # --warmup 0: everything was bracketed
            tagg = {}
            for n_, f_, e0, e1, _s, _b, ex_ in prof:
                a = tagg.setdefault(n_, [0.0, 0.0, 0, 0.0])
                a[0] += e0.elapsed_time(e1) * 1e-3; a[1] += f_; a[2] += 1; a[3] += f_ * ex_
            dom = max(tagg.items(), key=lambda kv: kv[1][0])[0]
            wagg_, wsteps_ = tagg, steps
            prof = [q for q in prof if q[0] == dom]
        tsum = sum(q[2].elapsed_time(q[3]) for q in prof) * 1e-3
        fsum = sum(q[1] for q in prof)
        xsum = sum(q[1] * q[6] for q in prof)          # flops the kernel issued (pixel-major tiles skip padding taps)
        cnt = max(len(prof), 1)
        achieved = fsum / max(tsum, 1e-12) / 1e12
        executed = xsum / max(tsum, 1e-12) / 1e12
        conv_time_per_step = sum(a[0] for a in wagg_.values()) / wsteps_
        issued_share_step = sum(a[3] for a in wagg_.values()) / max(sum(a[1] for a in wagg_.values()), 1.0)
        traffic, traffic_src, pmc_ent = _pmc_traffic(name, dom) if world == 1 else (None, None, None)
        fpi = cfg['flop_per_image']
        if cfg['d_reg_every'] > 1:      # lazy R1: price the R1 steps actually inside the timed window, not 1 / period
            fpi = cfg['flop_plain'] + cfg['flop_r1'] * (steps // cfg['d_reg_every']) / steps
        clock = (pmc_ent or {}).get('effective_clock_GHz')
        # `achieved` / `frac`: the multiply-adds the kernel ISSUES per second against the fp32 MFMA peak -- a roofline
        # fraction in the strict sense (<= 1 by construction).  `nominal_*`: the same time priced on the dense layer's count
        # 2*N*Ho*Wo*K*C*KH*KW (what every earlier round quoted as `frac`); it exceeds the issued figure where
        # pixel-major tiles skip tap-positions that read zero padding, i.e. it is a speed-up over the dense algorithm,
        # not a utilisation.
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(executed, 2), "peak": PEAK_FP32_MFMA,
                    "unit": "TFLOP/s", "frac": round(executed / PEAK_FP32_MFMA, 4),
                    "nominal_achieved": round(achieved, 2), "nominal_frac": round(achieved / PEAK_FP32_MFMA, 4),
                    "issued_share_of_nominal": round(xsum / max(fsum, 1.0), 4),
                    "traffic": traffic,
                    "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "effective_clock_GHz": None if clock is None else round(clock, 3),
                    "peak_at_effective_clock": None if clock is None else round(PEAK_FP32_MFMA * clock / 2.4, 1),
                    "frac_at_effective_clock": None if clock is None else round(executed / (PEAK_FP32_MFMA * clock / 2.4), 4),
                    "mfma_busy_fraction": (pmc_ent or {}).get('mfma_busy_fraction'),
                    "l2_hit_rate": (pmc_ent or {}).get('l2_hit_rate'),
                    "launches_per_step": cnt / nprof_steps, "avg_launch_ms": round(tsum / cnt * 1e3, 4),
                    "algorithmic_gflop_per_launch": round(xsum / cnt / 1e9, 2),
                    "nominal_gflop_per_launch": round(fsum / cnt / 1e9, 2),
                    "flop_convention": "achieved / frac price the multiply-adds the kernel issues (Winograd: the transform-domain "
                                       "products -- F(4x4,3x3), contrad_conv2d_path == 9 / 11: 1/4 of the dense layer's; F(2x2,3x3) / "
                                       "F(3x3,2x2), path 7: 4/9; F(2x2,2x2) on the phases of the strided layers, path 8: 9/16 (4x4 stride 2), path 10: 25/36 (3x3 stride 2, zero planes skipped); "
                                       "pixel-major tiles, path 3, skip the tap-positions that read zero padding on "
                                       "the 4x4 / 8x8 maps: contrad_conv2d_executed_fraction); nominal_* price the same time on the "
                                       "dense layer 2*N*Ho*Wo*K*C*KH*KW of the reference (SURVEY.md 8d) and may exceed the "
                                       "issued figure; peak = 157.3 TFLOP/s at 2.4 GHz, the clock the counters measured "
                                       "under this kernel is effective_clock_GHz (from the PMC summary in traffic_source)",
                    "schema": 2,    # 1 (rounds 1 - 4): achieved / frac on the NOMINAL count; 2: on the multiply-adds issued
                    "bracket": "HIP events around the C-ABI call on its stream" +
                               (" (igemm WGRAD kernel + its wgrad_reduce_kernel)" if "<2," in dom else "") +
                               (" (filter kernel + main kernel: the filter transform G g G^T is redone by every call)"
                                if dom.startswith(("wino_kernel", "wino22_kernel", "wino44_kernel")) else "") +
                               (" (wino_wgrad_kernel + wgrad_reduce_kernel)" if dom.startswith("wino_wgrad") else "") +
                               ("; the timed region replays one captured hipGraph per step, so the bracketed launches are "
                                "those of %d eager steps run right after it" % nprof_steps if graph_run else ""),
                    "conv_engine_share_of_step": round(conv_time_per_step / (dt / steps), 3),
                    "all_kernels_warmup": {k: {"tflops": round(v[3] / v[0] / 1e12, 1),
                                               "nominal_tflops": round(v[1] / v[0] / 1e12, 1),
                                               "ms_per_step": round(v[0] / wsteps_ * 1e3, 3)}
                                           for k, v in sorted(wagg_.items())},
                    "step_level": {"nominal_achieved": round(value / world * fpi / 1e12, 2),
                                   "nominal_frac": round(value / world * fpi / 1e12 / PEAK_FP32_MFMA, 4),
                                   "frac": round(value / world * fpi / 1e12 / PEAK_FP32_MFMA * issued_share_step, 4),
                                   "issued_share_of_nominal": round(issued_share_step, 4),
                                   "flop_per_image": fpi,
                                   "note": "whole step (every kernel, every gap) priced at SURVEY's nominal FLOPs per "
                                           "image; frac scales it by the issued / nominal share of the conv launches"}}
        out = {"metric": "discriminator-step images/sec (ContraD, SimCLR aug)", "value": round(value, 1),
               "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": cfg['workload'] % batch, "name": name,
                          "global_batch": global_batch, "per_gpu_batch": n_local,
                          "parallelism": "dp%d" % world, "losses_finite": finite,
                          "launch": launch,
                          "grad_exchange": None if not multi else ("overlapped with the backward (per-layer collectives)" if overlap
                                                                   else "after the backward (flat collectives)"),
                          "rccl_ranks": dist.get_world_size() if multi else 1,
                          "backend": dist.get_backend() if multi else None,
                          "peak_hbm_gib": round((peak_box[0] or torch.cuda.max_memory_allocated(dev)) / 2 ** 30, 2)},
               "roofline": roofline}
        if cfg['d_reg_every'] > 1:
            out["config"]["r1_steps_in_window"] = steps // cfg['d_reg_every']
        return out

    # With more than one rank main() runs every workload EAGERLY first (graph='off') and keeps those lines; this call
    # (graph on) only adds the replayed figure: a hipGraph with captured RCCL collectives has never been built on several
    # ranks at once on this stack, a capture that HANGS cannot be caught, and one that RAISES leaves the process in a state
    # nothing else should be measured in (measured with gloo: the stream stays "capture invalidated", the device RNG stays
    # in capture mode).  From arm() to disarm() a per-rank watchdog (--graph-timeout) reports instead of waiting.
    eager = None
    if use_graph and world > 1:
        if rank == 0 and eager_line is not None:
            line = json.loads(json.dumps(eager_line))
            line["config"]["launch"] = 'eager (graph capture timed out)'
            line["config"]["graph_hung"] = True        # explicit: the process leaves through os._exit(0)
            keep(name, line)
        arm(name)

    launch = 'hipGraph replay' if use_graph else 'eager'
    if use_graph:
        # the timed region replays ONE captured hipGraph per step (engine.GraphedDStep); events cannot sit inside a
        # graph, so the dominant kernel is bracketed on eager steps run right after the timed region instead
        from contrad_amd.engine import GraphedDStep, GraphedSG2DStep
        ops.PROFILE = None
        if args.shape_table and rank == 0:
            ops.SEQUENCE = []                # launch order of the step as it is CAPTURED (tools/rocpd_rows.py)
        try:
            if os.environ.get('CONTRAD_BENCH_FAKE_CAPTURE_HANG') and world > 1:      # test hook: a capture that never returns
                while True:
                    time.sleep(0.5)
            if name == 'c10_b512':
                graphed[0] = GraphedDStep(P, G, D, opt_D, options, images, warmup=1)
            else:
                graphed[0] = GraphedSG2DStep(P, G, D, opt_D, options, images, contrad_script=(name == 'sg2_512'), warmup=1)
            for _ in range(2):
                one_step()
            if cfg['d_reg_every'] > 1:
                # torch.cuda.graph() empties the caching allocator before capturing: run the eager lazy-R1 step once
                # more so that its ~25 GB of buffers are cached again before the timed window (a cold process paid 0.5 s
                # of hipMalloc inside the window for it: 96 instead of 67 ms per step over the 16-step window)
                graphed[0](cfg['d_reg_every'])
            if ops.SEQUENCE is not None and 'capture' in ops.SEQUENCE:
                seq = ops.SEQUENCE[len(ops.SEQUENCE) - ops.SEQUENCE[::-1].index('capture'):]
                per_step = [q for q in seq if q != 'capture']
                try:
                    tab = json.load(open(args.shape_table))
                    sec = tab['sections'].get('plain_step')
                    if sec is not None and len(per_step) % max(len(sec['sequence']), 1) == 0:
                        sec['graph_sequence'] = per_step[:len(sec['sequence'])]      # (the capture body runs the step once)
                        json.dump(tab, open(args.shape_table, 'w'), indent=1)
                except (OSError, ValueError, KeyError):
                    pass
            ops.SEQUENCE = None
        except Exception as e:              # capture not available on this stack: the eager launch sequence is the same work
            ops.SEQUENCE = None
            sys.stderr.write('bench.py: hipGraph capture failed (%r); timing the eager launch sequence\n' % (e,))
            graphed[0], use_graph = None, False
            launch = 'eager (graph capture failed: %s)' % type(e).__name__
            ops.PROFILE = []
            graph_failed = True
            # torch.cuda.graph's side stream is still the current stream (its __exit__ raised before restoring it), and the
            # runtime holds a pending error that the next synchronising call re-raises
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
            for _ in range(4):
                try:
                    torch.cuda.synchronize()
                    break
                except Exception:
                    pass
    if graph_failed and world > 1:
        # the eager line of this workload exists already (main(), phase 1); nothing more is measured in this process
        disarm(name)
        ops.PROFILE, ops.PROFILE_ONLY = None, None
        out = None
        if rank == 0 and eager_line is not None:
            out = json.loads(json.dumps(eager_line))
            out["config"]["launch"] = launch
        return out, False
    dt, finite = timed_region()
    if use_graph:                           # same kernels, same shapes, eager launches with the dominant kernel bracketed
        graphed[0] = None
        ops.PROFILE = []
        for _ in range(max(3, steps // 4)):
            one_step()
        torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    nprof_steps = max(3, steps // 4) if use_graph else steps
    if use_graph and world > 1:
        disarm(name)
    ops.PROFILE_ONLY = None
    peak_box[0] = torch.cuda.max_memory_allocated(dev)

    # ---- generator step, reported separately -- never part of `value` ----
    g_step = None
    if not args.no_g_step and not multi:
        try:
            from contrad_amd.training.gan.contrad import loss_G_fn
            opt_G = FusedAdam(G.parameters(), lr=opt['lr'], betas=tuple(opt['beta']))
            set_grad(G, True); set_grad(D, False)

            def g_once():
                if name == 'c10_b512':
                    gen = G(G.sample_latent(n_local))
                else:
                    gen = G(G.sample_latent(n_local), style_mix=0.9)
                g_loss = loss_G_fn(P, D, options, images, gen)
                opt_G.zero_grad()
                g_loss.backward()
                opt_G.step()
                return g_loss
            for _ in range(2):
                g_once()
            torch.cuda.synchronize()
            g_launch = 'eager'
            if use_graph:          # one captured hipGraph per generator step (engine.GraphedGStep / GraphedSG2GStep)
                try:
                    from contrad_amd.engine import GraphedGStep, GraphedSG2GStep
                    if name == 'c10_b512':
                        gg = GraphedGStep(P, G, D, opt_G, options, n_local, size, size)
                    else:
                        gg = GraphedSG2GStep(P, G, D, opt_G, options, n_local, size, size, style_mix=0.9)
                    g_once = gg
                    g_launch = 'hipGraph replay'
                    for _ in range(2):
                        g_once()
                    torch.cuda.synchronize()
                except Exception as e:
                    sys.stderr.write('bench.py: generator-step graph capture failed (%r); eager launches\n' % (e,))
                    torch.cuda.set_stream(torch.cuda.default_stream(dev))
            tg = time.perf_counter()
            reps = 10
            for _ in range(reps):
                gl = g_once()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg) / reps
            gfpi = cfg['g_flop_per_image']
            g_step = {"ms_per_step": round(tg * 1e3, 3), "images_per_sec": round(n_local / tg, 1),
                      "finite": bool(torch.isfinite(gl).item()), "launch": g_launch,
                      "roofline": {"step_level": {"nominal_achieved": round(n_local / tg * gfpi / 1e12, 2),
                                                  "nominal_frac": round(n_local / tg * gfpi / 1e12 / PEAK_FP32_MFMA, 4),
                                                  "flop_per_image": gfpi,
                                                  "note": "whole generator step priced at SURVEY.md 8(d)'s nominal FLOPs per "
                                                          "image against the fp32 MFMA peak (the Winograd layers issue 1/4 ... 9/16 of "
                                                          "their share)"}},
                      "what": "generator step (G forward with grad -> augment -> D -> loss_G_fn -> backward through D, the "
                              "augmentation and G -> Adam on G), %s, %d timed steps; not part of `value`" % (g_launch, reps)}
            set_grad(G, False); set_grad(D, True)
            del opt_G
        except Exception as e:                      # never let the side measurement take the headline down
            sys.stderr.write('bench.py: generator-step measurement skipped (%r)\n' % (e,))
    out = None
    if rank == 0:
        out = make_out(dt, finite, prof, nprof_steps, launch, dom_name, wagg, wsteps)
        if eager_line is not None:
            out["config"]["eager_ms_per_step"] = eager_line["ms_per_step"]
        if g_step is not None:
            out["g_step"] = g_step
    # release this workload's memory before the next one
    del G, D, opt_D, images, P
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    return out, True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 20 / 10 / 16 for c10_b512 / sg2_32 / sg2_512)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default: 5 / 3 / 3)')
    ap.add_argument('--config', default='all', choices=sorted(CONFIGS) + ['all'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-g-step', action='store_true', help='skip the separately reported generator-step timing')
    ap.add_argument('--shape-table', default=None,
                    help='write the per-(kernel, layer shape) table of the conv engine (from the bracketed warm-up) here')
    ap.add_argument('--graph', default='on', choices=['on', 'off'],
                    help='replay the D-step as one captured hipGraph (engine.GraphedDStep / GraphedSG2DStep; with a process group the '
                         'RCCL collectives are captured with it)')
    ap.add_argument('--force-dist', action='store_true',
                    help='dev: run all collective code paths on a 1-rank RCCL group (single GPU)')
    ap.add_argument('--dev-local-batch', type=int, default=0,
                    help='dev: single-GPU run at this batch (what one rank of an N-GPU job sees); not the headline config')
    ap.add_argument('--graph-timeout', type=float, default=240.0,
                    help='N > 1: seconds a workload may spend on capturing its step and timing the replay (every workload has '
                         'been timed eagerly before); past that every rank stops and rank 0 prints the line with the eager results')
    ap.add_argument('--exchange', default='auto', choices=['auto', 'overlap', 'serial'],
                    help="N > 1: the gradient exchange overlapped with the backward (per-layer collectives on RCCL's stream), after it "
                         "(flat collectives), or -- auto -- whichever of the two EAGER measurements is faster for the workload: "
                         "the Winograd kernels' blocks need whole CUs, and what a concurrent RCCL ring costs them has never been "
                         "measured on hardware (DESIGN.md section 6)")
    ap.add_argument('--dev-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="dev: process-group backend.  'gloo' lets several ranks share ONE GPU (RCCL refuses that), so the "
                         "self-launch, the barriers, the MAX over ranks and the fallback order can be exercised on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # self-launch: one process per GPU over RCCL, rendezvous on 127.0.0.1 (the container hostname may not resolve)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '4')
        raise SystemExit(subprocess.call(cmd, env=env))

    # (dmabuf IPC: the hosts of this pool support nothing else, and RCCL / tensor sharing across processes fail without it;
    # the HIP runtime reads it at its first call, which comes after this line)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('bench.py --gpus %d inside a launch of WORLD_SIZE=%d' % (args.gpus, world))
    if args.dev_backend == 'gloo':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    multi = world > 1 or args.force_dist
    # RCCL writes its banner and warnings to STDOUT, unterminated, possibly in the middle of our line: keep stdout to the
    # one JSON line by sending them to stderr (and never ask for the version banner)
    if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':
        os.environ['NCCL_DEBUG'] = 'WARN'
    os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29566')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.dev_backend == 'gloo':
            dist.init_process_group('gloo')
            _gloo_row_gather()
        else:
            dist.init_process_group('nccl', device_id=dev)     # "nccl" is RCCL on ROCm
        if args.force_dist:
            import contrad_amd.engine as _eng
            _eng.FORCE_DIST = True

    names = ['c10_b512', 'sg2_32', 'sg2_512'] if args.config == 'all' else [args.config]
    if args.dev_local_batch:
        names = names[:1]
    results = {}

    def emit():
        out = results[names[0]]
        rest = {n: results[n] for n in reversed(names[1:]) if n in results}      # sg2_512, then sg2_32
        if rest:
            out["other_configs"] = rest
        # the LAST key of the line: one short object per workload (a reader that keeps only the tail of the line still has
        # every workload's value -- VERDICT r5)
        out.pop("summary", None)
        summ = {}
        for n in names:
            r = results.get(n)
            if r and "error" not in r:
                summ[n] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                           "dominant_kernel": r["roofline"]["kernel"], "frac": r["roofline"]["frac"],
                           "step_frac": r["roofline"]["step_level"]["frac"],
                           "g_step_ms": (r.get("g_step") or {}).get("ms_per_step")}
        out["summary"] = summ
        print(json.dumps(out), flush=True)

    # Order with more than one rank: (1) EVERY workload is run and timed with eager launches and its
    # line kept; (2) then, workload by workload, the step is captured with its collectives and the replay is timed under a
    # watchdog -- a replay that completes replaces the eager line (the eager figure stays beside it); (3) a capture that
    # raises ends phase 2 (the process state is not trusted any more), one that hangs past --graph-timeout makes every rank
    # leave through os._exit(0) after rank 0 has printed the line: in both cases everything not replayed keeps its eager line.
    wd = _GraphWatchdog(rank, args.graph_timeout, results, emit)
    two_phase = world > 1 and args.graph != 'off'

    def guarded(name, first, **kw):
        if first:
            return run_config(name, args, world, rank, dev, multi, wd.keep, wd.arm, wd.disarm, **kw)
        try:                                     # a side workload must never take the headline line down
            return run_config(name, args, world, rank, dev, multi, wd.keep, wd.arm, wd.disarm, **kw)
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}, True

    chosen = {}                                  # workload -> gradient exchange overlapped with the backward?
    fixed = {'overlap': True, 'serial': False}.get(args.exchange) if world > 1 else None
    for i, name in enumerate(names):
        results[name], _ok = guarded(name, i == 0, graph='off' if two_phase else None, overlap=fixed)
        chosen[name] = fixed
        if world > 1 and args.exchange == 'auto':
            # the same workload once more, eagerly, with the exchange after the backward: keep the faster of the two lines
            # (every rank must take the same decision: rank 0 decides, the others are told)
            alt, _ok2 = guarded(name, False, graph='off', overlap=False)
            pick = torch.zeros(1, device=dev)
            if rank == 0:
                a, b = results[name] or {}, alt or {}
                if 'ms_per_step' in b and ('ms_per_step' not in a or b['ms_per_step'] < 0.97 * a['ms_per_step']):
                    pick.fill_(1.0)
            dist.broadcast(pick, 0)
            serial = bool(pick.item() > 0.5)
            if rank == 0 and 'ms_per_step' in (results[name] or {}) and 'ms_per_step' in (alt or {}):
                both = {"overlapped": results[name]['ms_per_step'], "after_the_backward": alt['ms_per_step']}
                if serial:
                    results[name] = alt
                results[name]["config"]["eager_ms_per_step_by_exchange"] = both
            chosen[name] = (not serial)
    if two_phase:
        for i, name in enumerate(names):
            if rank == 0 and "error" in (results[name] or {}):
                continue
            try:
                out, ok = run_config(name, args, world, rank, dev, multi, wd.keep, wd.arm, wd.disarm, graph='on',
                                     eager_line=results[name], overlap=chosen.get(name))
            except Exception as e:               # (anything else that goes wrong in the graph phase: keep the eager lines)
                sys.stderr.write('bench.py: graph phase of %s failed (%r); keeping the eager lines\n' % (name, e))
                break
            if rank == 0 and out is not None:
                results[name] = out
            if not ok:
                break
    if multi:
        try:
            dist.barrier()
            dist.destroy_process_group()         # before the JSON line: nothing RCCL prints can follow or split it
        except Exception as e:                   # (after a failed capture the process group may be unusable: still print)
            sys.stderr.write('bench.py: process-group shutdown failed (%r)\n' % (e,))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            for name in names:
                if "error" in results[name]:
                    continue
                if name == 'c10_b512':
                    results[name]["cpu_baseline"] = cpu_baseline_c10()
                elif name == 'sg2_32':
                    results[name]["cpu_baseline"] = cpu_baseline_sg2(name, 16)
                else:
                    results[name]["cpu_baseline"] = cpu_baseline_sg2(name, 2)
        emit()


if __name__ == '__main__':
    main()
