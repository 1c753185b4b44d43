import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import d_step, set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import contrad
dev = torch.device('cuda')
n = 512
G, D = get_architecture('sndcgan', (32, 32, 3)); G, D = G.to(dev).train(), D.to(dev).train()
P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(scale=(0.2, 1.0)), train_fn={'D': contrad.loss_D_fn})
opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)); set_grad(G, False)
x = torch.rand(n, 3, 32, 32, device=dev)
for _ in range(6): d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter(); ts = []
for i in range(24):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
    ts.append(time.perf_counter() - t0)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
print('host enqueue done times (ms):', ' '.join('%.0f' % (t * 1e3) for t in ts))
print('gpu step durations (ms):', ' '.join('%.1f' % evs[i].elapsed_time(evs[i + 1]) for i in range(24)))
print('mem allocated %.1f GB reserved %.1f GB' % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
st = torch.cuda.memory_stats()
print('num_alloc_retries', st.get('num_alloc_retries'), 'segments', st.get('segment.all.allocated'), 'cudaMalloc calls', st.get('num_device_alloc'), 'frees', st.get('num_device_free'))

# ---- which host call blocks? ----
import contrad_amd._lib as L
import contrad_amd.ops as O2
lib = L.lib()
orig_call = lib.call
slow = []
def timed_call(name, *a):
    t = time.perf_counter(); orig_call(name, *a); dt = time.perf_counter() - t
    if dt > 2e-3: slow.append((name, dt * 1e3))
lib.call = timed_call
import torch.cuda
orig_empty = torch.empty
def timed_empty(*a, **k):
    t = time.perf_counter(); r = orig_empty(*a, **k); dt = time.perf_counter() - t
    if dt > 2e-3: slow.append(('torch.empty%s' % (a[:1],), dt * 1e3))
    return r
torch.empty = timed_empty
torch.cuda.synchronize()
for i in range(12):
    t = time.perf_counter(); d_step(P, G, D, opt, {'loss': 'nonsat'}, x); dt = time.perf_counter() - t
    print('step %d host %.1f ms; slow calls: %s' % (i, dt * 1e3, slow)); slow.clear()

import faulthandler
lib.call = orig_call; torch.empty = orig_empty
torch.cuda.synchronize()
f = open('/tmp/tb.txt', 'w')
faulthandler.dump_traceback_later(0.02, repeat=True, file=f)
for i in range(9):
    d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
faulthandler.cancel_dump_traceback_later()
f.close()
import collections, re
txt = open('/tmp/tb.txt').read()
blocks = txt.split('Timeout (')
c = collections.Counter()
for b in blocks[1:]:
    # innermost frames of each thread
    for th in b.split('Thread ')[1:] + b.split('Current thread ')[1:]:
        lines = [l.strip() for l in th.splitlines() if l.strip().startswith('File')]
        if lines:
            c[' <- '.join(lines[:3])] += 1
for k, v in c.most_common(8):
    print(v, k)
