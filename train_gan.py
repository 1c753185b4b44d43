#!/usr/bin/env python
"""Entry point with the reference's command line (train_gan.py:41-85):

    python train_gan.py configs/gan/cifar10/c10_b512.gin sndcgan --mode=contrad --aug=simclr --use_warmup
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_gan.py <same arguments>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from contrad_amd.train_gan import main  # noqa: E402

if __name__ == '__main__':
    main()
