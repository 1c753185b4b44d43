"""Dev: sweep tile x split count per (mode, layer shape) of the bench workloads against the plan the library picks itself.

usage (GPU box):  python tools/tune_plans.py profiles/r04_sg2_32_n1_shapes.json [more shape tables ...] [--min-us 15]

Reads the shape tables `bench.py --shape-table` writes (rows: kernel, shape = N,H,W,C,K,KH,KW,s,p, launches_per_step), runs
every distinct (mode, shape) of the lean igemm kernels under the library's own plan and under every candidate forced through
`contrad_dev_plan_override` (exported by libcontrad_hip_dev.so only), checks that the forced result equals the default one,
and prints per shape the best candidate and what it would save per step.  Output feeds the decision whether a rule of the
plan functions (igemm.hip: pick_tile / split_plan / wgrad_plan) is wrong for a family of shapes -- not a lookup table.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('CONTRAD_HIP_LIB', os.path.join(ROOT, 'contrad_amd', 'csrc', 'libcontrad_hip_dev.so'))
import torch                                  # noqa: E402
from contrad_amd import ops, _lib             # noqa: E402

TILES = [(128, 128), (64, 128), (128, 64), (64, 64)]
ITERS, WARM = 8, 3


def timeit(fn):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3      # us


def rows_of(paths):
    seen = {}
    for path in paths:
        table = json.load(open(path))
        for _, sec in table['sections'].items() if isinstance(table['sections'], dict) else table['sections']:
            for r in sec['rows']:
                m = re.match(r'igemm_lean_kernel<(\d), (\d+), (\d+)', r['kernel'])
                if not m:
                    continue
                key = (int(m.group(1)), tuple(r['shape']))
                ent = seen.setdefault(key, {'w': 0.0, 'tile': (int(m.group(2)), int(m.group(3))), 'cfg': table['config']})
                ent['w'] = max(ent['w'], r['launches_per_step'])
    return seen


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith('--') and sys.argv[i - 1] not in ('--dump', '--min-us')]
    min_us = float(sys.argv[sys.argv.index('--min-us') + 1]) if '--min-us' in sys.argv else 15.0
    dev = torch.device('cuda')
    import ctypes
    override = getattr(_lib.lib()._dll, 'contrad_dev_plan_override')     # dev library only: not in include/contrad_hip.h
    override.restype, override.argtypes = None, [ctypes.c_int] * 3
    out, dump = [], []
    for (mode, shape), ent in sorted(rows_of(args).items()):
        N, H, W, C, K, KH, KW, s, p = shape
        if C % 16 or K % 16:
            continue
        Ho, Wo = ops.out_size(H, KH, s, p), ops.out_size(W, KW, s, p)
        x = torch.randn(N, H, W, C, device=dev)
        gy = torch.randn(N, Ho, Wo, K, device=dev)
        wp = torch.randn(KH * KW * C, K, device=dev) * 0.05
        if mode == 0:
            y = torch.empty(N, Ho, Wo, K, device=dev)
            fn, res = (lambda: ops.conv2d_fwd(x, wp, None, K, KH, KW, s, p, 0.2, 1.0, out=y)), y
        elif mode == 1:
            dx = torch.empty(N, H, W, C, device=dev)
            fn, res = (lambda: ops.conv2d_dgrad(gy, wp, (N, H, W, C), KH, KW, s, p, out=dx)), dx
        else:
            dw = torch.empty_like(wp)
            fn, res = (lambda: ops.conv2d_wgrad(x, gy, KH, KW, s, p, out=dw)), dw
        override(0, 0, 0)
        timeit(fn)                               # (clocks, caches: the first timing of a shape reads up to 8 % slow)
        t0 = timeit(fn)
        if t0 < min_us:
            continue
        ref = res.clone()
        if mode == 2:
            tm_tn = [(-(-KH * KW * C // bm)) * (-(-K // bn)) for bm, bn in TILES]
            splits_for = lambda i: sorted({max(1, b // tm_tn[i]) for b in (384, 512, 768, 1024, 1536, 2048)})
        elif s == 1:
            splits_for = lambda i: [1, 2, 3, 4, 6, 8, 12, 16]
        else:
            splits_for = lambda i: [1]
        best = (t0, None)
        cands = []
        for i, (bm, bn) in enumerate(TILES):
            if bn > 64 and K <= 64 and mode != 1:
                continue
            if bn > 64 and C <= 64 and mode == 1:
                continue
            for sp in splits_for(i):
                override(bm, bn, sp)
                try:
                    t = timeit(fn)
                except RuntimeError:
                    continue
                err = ((res - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
                if err > 1e-4:
                    print('!! mismatch', mode, shape, bm, bn, sp, err, flush=True)
                    continue
                cands.append((t, (bm, bn, sp)))
        override(0, 0, 0)
        t0 = min(t0, timeit(fn))                 # the library's own plan once more, after the candidates
        # the three fastest candidates again, interleaved with nothing else: a candidate must win twice
        for t, c in sorted(cands)[:3]:
            override(*c)
            t = max(t, timeit(fn))
            if t < best[0]:
                best = (t, c)
        override(0, 0, 0)
        dump.append({'cfg': ent['cfg'], 'mode': mode, 'shape': list(shape), 'plan_tile': list(ent['tile']), 'w': ent['w'], 't_plan_us': t0,
                     'cands': [[c[0], c[1], c[2], t] for t, c in cands]})
        gain = (t0 - best[0]) * ent['w']
        out.append((gain, ent['cfg'], mode, shape, ent['tile'], ent['w'], t0, best))
        print('%-8s mode %d %-38s plan %3dx%-3d x%.0f/step  %8.1f us -> best %-16s %8.1f us  (%4.1f %%, %6.1f us/step)' % (
            ent['cfg'], mode, ','.join(map(str, shape)), ent['tile'][0], ent['tile'][1], ent['w'], t0, best[1], best[0],
            100.0 * (t0 - best[0]) / t0, gain), flush=True)
        del x, gy, wp
    if '--dump' in sys.argv:
        json.dump(dump, open(sys.argv[sys.argv.index('--dump') + 1], 'w'))
    print('== total potential per step: %.1f us over %d shapes' % (sum(o[0] for o in out), len(out)))


if __name__ == '__main__':
    main()
