"""TEST INFRASTRUCTURE ONLY.

CPU restatement (pure PyTorch-CPU / numpy) of the ContraD discriminator-step hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``contrad_amd``) never does.
"""
