#!/usr/bin/env python
"""Headline benchmark: discriminator-step images/sec (ContraD, SimCLR aug), SNDCGAN on CIFAR-10-shaped synthetic
data, global batch 512 (BASELINE.json configs[1]; configs[2] = the same global batch over N GPUs).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

One "step" = one full D-step of the reference loop (train_gan.py:153-163): no-grad G forward for N fakes ->
SimCLR-augment 3N images -> D forward -> NT-Xent + SupCon + non-saturating GAN loss -> backward -> [embedding
all-gather / gradient all-reduce over RCCL] -> Adam on D.  Inputs are resident in HBM before the timed region;
weights are random-init.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GLOBAL_BATCH = 512
FLOP_PER_IMAGE = 4.28e9        # algorithmic FLOPs of one SNDCGAN D-step per real image
PEAK_FP32_MFMA = 157.3         # TFLOP/s, MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)


def cpu_baseline(n=64, steps=5):
    """BASELINE.json configs[0] on the host cores: the oracle's (= reference algorithm's) PyTorch-CPU D-step."""
    from oracle import contrad_oracle as O
    torch.manual_seed(0); np.random.seed(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))      # the 32x32 convs stop scaling (and oversubscribe) beyond that
    torch.set_num_threads(threads)
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

    t0 = time.perf_counter()
    step(1)
    warm = time.perf_counter() - t0
    steps = max(1, min(steps, int(20.0 / max(warm, 1e-3))))      # bound the sample to ~20 s of CPU work
    t0 = time.perf_counter()
    for t in range(steps):
        step(t + 2)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "SNDCGAN ContraD D-step, 32x32, batch %d, %d timed steps after 1 warm-up "
                      "(oracle = PyTorch-CPU restatement of the reference path, %.3f s/step)" % (n, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='dev: run all collective code paths on a 1-rank RCCL group (single GPU)')
    ap.add_argument('--dev-local-batch', type=int, default=0,
                    help='dev: single-GPU run at this batch (what one rank of an N-GPU job sees); not the headline config')
    args = ap.parse_args()
    global GLOBAL_BATCH
    if args.dev_local_batch:
        GLOBAL_BATCH = args.dev_local_batch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, args.gpus))
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
        dist.init_process_group('nccl', device_id=dev)     # "nccl" is RCCL on ROCm
        if args.force_dist:
            import contrad_amd.engine as _eng
            _eng.FORCE_DIST = True

    from contrad_amd import config, ops
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import GradAllReducer, OverlappedGradReducer, d_step, set_grad
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup

    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b512.gin')])
    opt = config.get_bindings('options')
    assert (opt['batch_size'] == GLOBAL_BATCH or args.dev_local_batch) and GLOBAL_BATCH % world == 0
    n_local = GLOBAL_BATCH // world                          # train_gan.py:247

    torch.manual_seed(0 + rank); np.random.seed(0 + rank)
    G, D = get_architecture('sndcgan', (32, 32, 3))
    if world > 1:                                            # identical weights on every rank (DDP's broadcast)
        torch.manual_seed(0)
        G, D = get_architecture('sndcgan', (32, 32, 3))
        torch.manual_seed(0 + rank)
    G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=multi)
    P = setup(P)
    P.augment_fn = get_augment(mode=P.aug).to(dev)
    options = {'loss': opt['loss'], 'batch_size': n_local}
    opt_D = FusedAdam(D.parameters(), lr=opt['lr'], betas=tuple(opt['beta']))
    reducer = None
    if multi:
        if os.environ.get('CONTRAD_NO_OVERLAP'):
            reducer = GradAllReducer(D.parameters())          # two flat collectives after the backward
        else:
            D.enable_grad_overlap(OverlappedGradReducer())      # per-layer collectives hidden behind the backward
    set_grad(G, False); set_grad(D, True)
    images = torch.rand(n_local, 3, 32, 32, device=dev)      # synthetic CIFAR-shaped batch, resident in HBM

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
    for _ in range(args.warmup):
        d_step(P, G, D, opt_D, options, images, reducer)
        marks.append(len(ops.PROFILE))
    torch.cuda.synchronize()
    warm_prof = ops.PROFILE
    ops.PROFILE = []
    wsteps = max(1, args.warmup // 2)                                # the later half of the warm-up (clocks ramped)
    wagg = {}
    for name, flops, e0, e1 in warm_prof[marks[max(0, args.warmup - wsteps)]:]:
        a = wagg.setdefault(name, [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += flops
        a[2] += 1
    ops.PROFILE_ONLY = max(wagg.items(), key=lambda kv: kv[1][0])[0] if wagg else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d_loss, aux = d_step(P, G, D, opt_D, options, images, reducer)
    barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    dom_name, ops.PROFILE_ONLY = ops.PROFILE_ONLY, None
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = GLOBAL_BATCH * args.steps / dt
    finite = bool(torch.isfinite(d_loss).item() and torch.isfinite(aux['penalty']).item())

    if rank == 0:
        # dominant kernel = the conv-engine instance with the largest summed device time (chosen on the warm-up
        # steps); its launches in the timed region are the `achieved` figure
        if dom_name is None:                                        # --warmup 0: everything was bracketed
            tagg = {}
            for n_, f_, e0, e1 in prof:
                a = tagg.setdefault(n_, [0.0, 0.0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3; a[1] += f_; a[2] += 1
            dom_name = max(tagg.items(), key=lambda kv: kv[1][0])[0]
            wagg, wsteps = tagg, args.steps
            prof = [q for q in prof if q[0] == dom_name]
        tsum = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof) * 1e-3
        fsum = sum(f for _, f, _, _ in prof)
        cnt = len(prof)
        name = dom_name
        achieved = fsum / tsum / 1e12
        conv_time_per_step = sum(a[0] for a in wagg.values()) / wsteps
        # HBM traffic per launch of the dominant kernel: from the committed rocprofv3 --pmc passes of this same
        # command (profiles/r01_bench_n1_pmc.*; separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bench_n1_pmc.json')))
            ent = pmc.get('kernels', {}).get(name) or (pmc if pmc.get('kernel') == name else None)
            if ent is not None and world == 1:
                traffic, traffic_src = ent['traffic_bytes_per_launch'], 'profiles/r01_bench_n1_pmc.json'
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA, 4), "traffic": traffic,
                    "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "launches_per_step": cnt / args.steps, "avg_launch_ms": round(tsum / cnt * 1e3, 4),
                    "algorithmic_gflop_per_launch": round(fsum / cnt / 1e9, 2),
                    "bracket": "HIP events around the C-ABI call on its stream" +
                               (" (igemm WGRAD kernel + its wgrad_reduce_kernel)" if "<2," in name else ""),
                    "conv_engine_share_of_step": round(conv_time_per_step / (dt / args.steps), 3),
                    "all_kernels_warmup": {k: {"tflops": round(v[1] / v[0] / 1e12, 1),
                                               "ms_per_step": round(v[0] / wsteps * 1e3, 3)}
                                           for k, v in sorted(wagg.items())},
                    "step_level": {"achieved": round(value / world * FLOP_PER_IMAGE / 1e12, 2),
                                   "frac": round(value / world * FLOP_PER_IMAGE / 1e12 / PEAK_FP32_MFMA, 4),
                                   "flop_per_image": FLOP_PER_IMAGE}}
        out = {"metric": "discriminator-step images/sec (ContraD, SimCLR aug)", "value": round(value, 1),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "SNDCGAN + ContraD D-step, CIFAR-10 32x32, global batch %d, simclr aug, "
                                      "nonsat loss, Adam(2e-4,(0.5,0.999)), random-init weights" % GLOBAL_BATCH,
                          "global_batch": GLOBAL_BATCH, "per_gpu_batch": n_local,
                          "parallelism": "dp%d" % world, "losses_finite": finite},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if multi:
        dist.barrier()
        dist.destroy_process_group()             # before the JSON line: nothing RCCL prints can follow or split it
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
