"""dev: which lines of contrad_amd/ launch ATen kernels in a D-step?  One eager step under torch.profiler with Python
stacks; device-kernel launches that are not ours (at::native::*, rocclr copies) are attributed to the innermost frame
inside contrad_amd/ (or bench.py).  usage: python tools/dev/aten_sources.py sg2_32|sg2_512|c10_b512"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(name):
    from contrad_amd import config
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import d_step, d_step_stylegan2, d_step_stylegan2_contrad, set_grad
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup
    cfg = bench.CONFIGS[name]
    dev = torch.device('cuda', 0)
    size, n = cfg['size'], cfg['batch']
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, *cfg['gin'])])
    opt = config.get_bindings('options')
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture(cfg['arch'], (size, size, 3))
    G, D = G.to(dev).train(), D.to(dev).train()
    P = setup(argparse.Namespace(mode='contrad', aug=cfg['aug'], temp=0.1, lbd_a=1.0, distributed=False,
                                 lbd_r1=cfg['lbd_r1'], d_reg_every=max(cfg['d_reg_every'], 1)))
    P.augment_fn = get_augment(mode=P.aug).to(dev)
    options = {'loss': opt['loss'], 'batch_size': n}
    opt_D = FusedAdam(D.parameters(), lr=opt['lr'], betas=tuple(opt['beta']))
    set_grad(G, False); set_grad(D, True)
    images = torch.rand(n, 3, size, size, device=dev)
    if name == 'c10_b512':
        step = lambda s: d_step(P, G, D, opt_D, options, images, None)
    else:
        fn = d_step_stylegan2 if name == 'sg2_32' else d_step_stylegan2_contrad
        step = lambda s: fn(P, G, D, opt_D, options, images, s, None)
    for s in (1, 2, 3):
        step(s)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    by = collections.defaultdict(lambda: [0, 0])
    skip = ('aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::detach', 'aten::alias', 'aten::t', 'aten::expand',
            'aten::permute', 'aten::slice', 'aten::select', 'aten::as_strided', 'aten::empty', 'aten::transpose',
            'aten::split', 'aten::unsqueeze', 'aten::squeeze', 'aten::empty_like', 'aten::empty_strided',
            'aten::split_with_sizes', 'aten::_local_scalar_dense', 'aten::is_', 'aten::stride', 'aten::size',
            'aten::new_empty', 'aten::unbind', 'aten::record_stream', 'aten::lift_fresh', 'aten::narrow')

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            nm = str(func._schema.name)
            if nm.startswith(skip):
                return out
            frame = '<no contrad_amd frame: C++ autograd of an ATen op / engine accumulation>'
            for fr in reversed(traceback.extract_stack()):
                if 'contrad_amd/' in fr.filename and 'aten_sources' not in fr.filename:
                    frame = '%s:%d %s' % (fr.filename.split('contrad_amd/')[-1], fr.lineno, fr.name)
                    break
            numel = 0
            o = out[0] if isinstance(out, (tuple, list)) and out else out
            if torch.is_tensor(o):
                numel = o.numel()
            if frame.startswith('<no'):
                # C++-side op: name the autograd node being evaluated (its own ATen backward, or the engine summing the
                # gradient contributions of a tensor with several consumers while that node's outputs are routed)
                node = torch._C._current_autograd_node() if hasattr(torch._C, '_current_autograd_node') else None
                frame = '<C++ autograd, node %s> shape %s' % (node.name() if node is not None else 'none',
                                                               tuple(o.shape) if torch.is_tensor(o) else '-')
            k = (nm, frame)
            by[k][0] += 1
            by[k][1] += numel
            return out
    with Mode():
        step(5)                      # a plain step (lazy R1: not an R1 step)
        torch.cuda.synchronize()
    print('%s: ATen ops dispatched in one eager step (count, summed output elements), by innermost contrad_amd frame' % name)
    for (nm, fr), (c, n_) in sorted(by.items(), key=lambda kv: -kv[1][0])[:90]:
        print('%5d %14d  %-26s %s' % (c, n_, nm, fr))


if __name__ == '__main__':
    main(sys.argv[1])
