"""Host -> device hand-over of the per-step random draws (latents, augmentation parameters).

The draws happen on the host in the reference's RNG order; what this module controls is only HOW the few hundred KB
reach the GPU.  A synchronous pageable ``tensor.to(device)`` makes the host wait until the GPU has drained everything
queued before it -- twice per step the GPU then idles until the next kernels are enqueued (0.6 ms of a 4.1 ms step at
a per-rank batch of 64, 0.5 of 18.1 ms at 512).  ``upload()`` instead copies into a small ring of pinned staging
buffers and lets a kernel pull the data over PCIe from the device-mapped pinned memory (stream-ordered, no runtime
copy), and ``StepThrottle`` keeps the host at most ONE step ahead of the GPU.

Two ROCm 7.2 observations are built in (tools/host_time.py, bench.py --dev-local-batch):
  * a single asynchronous hand-over of more than ~128 KB -- hipMemcpyAsync from pinned memory or the pull kernel alike --
    makes every following step take 2-3x as long (33-53 instead of 18 ms at batch 512, where the latent block is
    256 KB); pieces of 64 KB do not.  Cause not understood; big tensors therefore go in 64 KB pieces.
  * unbounded host run-ahead (>= 3 steps queued) makes the runtime drain its queue every third step; hence the throttle.

Measured step times, synchronous -> asynchronous (1 GPU, SNDCGAN D-step, per-rank batch 1024 / 512 / 256 / 128 / 64):
37.5 -> 33.9, 18.1 -> 17.6, 10.0 -> 9.5, 6.1 -> 5.5, 4.1 -> 3.5 ms.   CONTRAD_SYNC_UPLOADS=1 restores the synchronous path.
"""
import os

import torch

_SYNC = os.environ.get('CONTRAD_SYNC_UPLOADS', '0') == '1'
_CHUNK = int(os.environ.get('CONTRAD_UPLOAD_CHUNK', str(64 * 1024)))     # bytes per asynchronous piece
_RING = 8                      # staging slots per (shape, dtype); a slot is only reused after its own pull has completed
_rings = {}


def _pull(t, device):
    """One piece: host tensor -> pinned ring slot -> device tensor written by contrad_pull_host on the current stream."""
    from . import ops
    key = (tuple(t.shape), t.dtype)
    ring = _rings.get(key)
    if ring is None:
        ring = _rings[key] = [[[torch.empty(t.shape, dtype=t.dtype).pin_memory(), None] for _ in range(_RING)], 0]
    slots, i = ring
    ring[1] = (i + 1) % _RING
    buf, done = slots[i]
    if done is not None:
        done.synchronize()     # (returns at once unless the caller is more than _RING uploads ahead of the GPU)
    buf.copy_(t)
    out = torch.empty(t.shape, dtype=t.dtype, device=device)
    ops.lib().call('contrad_pull_host', buf.data_ptr(), out.data_ptr(), t.numel(), ops._stream())
    slots[i][1] = torch.cuda.Event()
    slots[i][1].record()
    return out


def upload(t, device):
    """CPU float32 tensor -> device tensor, stream-ordered on the current stream, without a host-side wait."""
    if _SYNC or t.dtype != torch.float32 or torch.device(device).type != 'cuda':
        return t.to(device)
    nbytes = t.numel() * t.element_size()
    if nbytes <= _CHUNK or t.dim() != 2:
        return _pull(t.contiguous(), device) if nbytes <= 2 * _CHUNK else t.to(device)
    rows = max(1, _CHUNK // (t.shape[1] * t.element_size()))
    return torch.cat([_pull(t[i:i + rows].contiguous(), device) for i in range(0, t.shape[0], rows)], 0)


class StepThrottle(object):
    """Bounds the host's run-ahead to one step: ``begin()`` at the top of a step waits for the end of the step before the
    previous one, ``end()`` marks the end of this step's launches."""

    def __init__(self):
        self.events = []
        self.depth = 0                      # re-entrant: a step function called inside a throttled iteration is a no-op

    def begin(self):
        self.depth += 1
        if self.depth > 4:                  # no step nests this deep: an exception left the counter stuck -- start over
            self.depth = 1
        if self.depth > 1:
            return
        while len(self.events) >= 2:
            self.events.pop(0).synchronize()

    def end(self):
        self.depth = max(self.depth - 1, 0)
        if _SYNC or self.depth > 0:
            return
        e = torch.cuda.Event()
        e.record()
        self.events.append(e)


THROTTLE = StepThrottle()
