"""Dev: is the D-step host-bound?  Times the enqueue loop against the synchronised total and cProfiles the host side."""
import cProfile, os, pstats, sys, time, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd import config
from contrad_amd.augment import get_augment
from contrad_amd.engine import d_step, set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import setup

B = int(os.environ.get('HT_BATCH', '512'))
dev = torch.device('cuda')
config.clear_config()
config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                        os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                        os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b512.gin')])
opt = config.get_bindings('options')
torch.manual_seed(0); np.random.seed(0)
G, D = get_architecture('sndcgan', (32, 32, 3))
G, D = G.to(dev).train(), D.to(dev).train()
P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=False))
P.augment_fn = get_augment(mode=P.aug).to(dev)
options = {'loss': opt['loss'], 'batch_size': B}
opt_D = FusedAdam(D.parameters(), lr=opt['lr'], betas=tuple(opt['beta']))
set_grad(G, False); set_grad(D, True)
images = torch.rand(B, 3, 32, 32, device=dev)
for _ in range(5):
    d_step(P, G, D, opt_D, options, images, None)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    d_step(P, G, D, opt_D, options, images, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('batch %d: host enqueue %.2f ms/step, synchronised total %.2f ms/step' % (B, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
ms = torch.cuda.memory_stats()
print('allocator: segments allocated %d freed %d, retries %d, reserved %.2f GB, peak allocated %.2f GB' % (
    ms.get('segment.all.allocated', -1), ms.get('segment.all.freed', -1), ms.get('num_alloc_retries', -1),
    ms.get('reserved_bytes.all.current', 0) / 2**30, ms.get('allocated_bytes.all.peak', 0) / 2**30))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    d_step(P, G, D, opt_D, options, images, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
