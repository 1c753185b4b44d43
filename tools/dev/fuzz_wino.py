"""dev: random shapes through the forced Winograd entry points (F(4x4,3x3) both cout-block widths, F(2x2,3x3), the strided kernels)
against PyTorch-CPU; prints the worst relative error per kernel family.  usage: python tools/dev/fuzz_wino.py [cases] [seed]"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from contrad_amd import ops  # noqa: E402


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main(cases=60, seed=0):
    rnd = random.Random(seed)
    dev = torch.device('cuda')
    worst = {}
    for it in range(cases):
        fam = rnd.choice(['f44', 'f44', 'f44', 'f22', 'k4s2', 'k3s2'])
        g = torch.Generator().manual_seed(seed * 1000 + it)
        if fam in ('f44', 'f22'):
            H = rnd.choice([4, 8, 16, 32, 64]) if fam == 'f44' else rnd.choice([4, 8, 16, 32])
            W = H if H < 32 else rnd.choice([32, 64])
            if fam == 'f44' and W >= 32 and H < 16:
                H = 16
            C = rnd.choice([32, 64, 96, 128]) if fam == 'f44' else rnd.choice([16, 32, 48, 64])
            K = rnd.choice([32, 64, 96, 128, 160]) if fam == 'f44' else rnd.choice([64, 128])
            N = rnd.choice([1, 2, 3, 5, 9, 33, 70])
            mode = rnd.choice([0, 1])
            x = torch.randn(N, H, W, C if mode == 0 else K, generator=g)
            w = torch.randn(K, C, 3, 3, generator=g) * 0.1
            if mode == 1 and ((fam == 'f44' and (K % 32 or C % 32)) or (fam == 'f22' and (K % 16 or C % 64))):
                continue
            wp = ops.pack_weight(w).to(dev)
            try:
                y = ops.conv2d_wino(mode, x.to(dev), wp, C, K, f44=(fam == 'f44'))
            except RuntimeError:
                continue
            ref = (F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1) if mode == 0 else
                   F.conv_transpose2d(x.permute(0, 3, 1, 2), w, padding=1)).permute(0, 2, 3, 1)
            key = '%s mode %d%s' % (fam, mode, ' (32-wide)' if fam == 'f44' and (((K if mode == 0 else C) % 64) or W == 4) else '')
        elif fam == 'k4s2':
            G = rnd.choice([4, 8, 16]); C = rnd.choice([8, 16, 24, 32]); K = rnd.choice([64, 128]); N = rnd.choice([1, 3, 9, 40])
            x = torch.randn(N, 2 * G, 2 * G, C, generator=g); w = torch.randn(K, C, 4, 4, generator=g) * 0.1
            y = ops.conv2d_wino(0, x.to(dev), ops.pack_weight(w).to(dev), C, K, k4s2=True)
            ref = F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2, padding=1).permute(0, 2, 3, 1)
            key = 'k4s2 forward'
        else:
            G = rnd.choice([4, 8, 16, 32, 64]); C = rnd.choice([16, 32, 48]); K = rnd.choice([64, 128]); N = rnd.choice([1, 2, 5, 9, 40])
            x = torch.randn(N, 2 * G + 1, 2 * G + 1, C, generator=g); w = torch.randn(K, C, 3, 3, generator=g) * 0.1
            y = ops.conv2d_wino(0, x.to(dev), ops.pack_weight(w).to(dev), C, K, k3s2=True)
            ref = F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2).permute(0, 2, 3, 1)
            key = 'k3s2 forward'
        e = rel(y.cpu(), ref)
        n, m = worst.get(key, (0, 0.0))
        worst[key] = (n + 1, max(m, e))
        assert e < 1e-4, (key, tuple(x.shape), C, K, e)
    for k in sorted(worst):
        print('%-28s %3d cases  worst rel err %.2e' % (k, worst[k][0], worst[k][1]))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
