#!/usr/bin/env python
"""Entry point with the reference's name and CLI: `python train_stylegan2.py <gin> <architecture> --mode=contrad ...`
(one process per GPU; launch N ranks with torch.distributed.run).  The loop lives in contrad_amd/train_stylegan2.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from contrad_amd.train_stylegan2 import main  # noqa: E402

if __name__ == '__main__':
    main()
