"""dev: which ingredient of the training loop breaks the capture (run each variant in its own process)."""
import subprocess
import sys

VARIANTS = ['w1', 'w1_gstep', 'w2_gstep', 'w1_warm']

if len(sys.argv) > 1:
    import argparse, os
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
    from test_graph_gpu import _setup
    from contrad_amd.engine import GraphedDStep, d_step, set_grad, sample_generator
    from contrad_amd.optim import FusedAdam
    v = sys.argv[1]
    P, G, D, opt, x = _setup(64)
    opt_G = FusedAdam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    nw = 2 if v.startswith('w2') else 1
    for _ in range(nw):
        d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
    if 'gstep' in v:
        set_grad(G, True); set_grad(D, False)
        gen = sample_generator(G, 64)
        g_loss = P.train_fn["G"](P, D, {'loss': 'nonsat'}, x, gen)
        opt_G.zero_grad(); g_loss.backward(); opt_G.step()
        set_grad(G, False); set_grad(D, True)
    if 'warm' in v:
        for g in opt.param_groups:
            g['lr'] = 1e-5
    g = GraphedDStep(P, G, D, opt, {'loss': 'nonsat'}, x, warmup=0)
    for _ in range(3):
        dl, aux = g()
    torch.cuda.synchronize()
    print(v, 'ok', dl.item())
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        print(v, 'rc', r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-300:].replace('\n', ' | '))
