"""contrad_amd -- MI355X-native (gfx950) implementation of ContraD's discriminator-step hot path.

Host orchestration is Python on PyTorch-ROCm (device memory, streams, autograd graph, torch.distributed);
every compute stage between "3N images in HBM" and "losses + parameter gradients + Adam" is a hand-written
HIP kernel in ``csrc/`` behind the C ABI of ``include/contrad_hip.h`` (``libcontrad_hip.so``, loaded with
ctypes).  There is no CPU or PyTorch fallback: ops raise if the library is missing.
"""
__version__ = "0.1.0"
