"""Fused multi-tensor Adam on the HIP kernel (one launch per 64 tensors), numerically torch.optim.Adam as the
reference constructs it (train_gan.py:273-274: lr, betas, eps=1e-8, no weight decay / amsgrad).  The state layout
(`step`, `exp_avg`, `exp_avg_sq`) matches torch's so ``optim.pt`` checkpoints interchange (train_gan.py:219-223)."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] = int(st['step']) + 1
                by_step.setdefault(st['step'], []).append(p)
            for step, ps in by_step.items():
                ops.adam_step([p.data for p in ps], [p.grad.data for p in ps],
                              [self.state[p]['exp_avg'] for p in ps], [self.state[p]['exp_avg_sq'] for p in ps],
                              step, group['lr'], beta1, beta2, group['eps'], grad_scale)
                # the kernel writes through raw pointers: tell autograd's version counters, so that everything keyed
                # on ``p._version`` (the generators' packed-weight caches) sees the update
                for p in ps:
                    torch.autograd.graph.increment_version(p)
        return loss

    # ---- hipGraph support: the launch sequence is captured once, the step-dependent scalars live in device memory ----
    def hyper_values(self, grad_scale=1.0):
        """Advance the step count of the parameters the captured update touches (those that had a gradient when
        ``step_captured`` was recorded -- the same set the eager ``step()`` advances; all parameters with state before
        any capture) by one and return [lr / bc1, 1 / sqrt(bc2), grad_scale] of the (single) param group for THAT step
        -- what ``step_captured`` reads from its device tensor."""
        import math
        import struct
        group = self.param_groups[0]
        # the C launcher receives the betas as floats and forms the bias corrections in double from those
        f32 = lambda v: struct.unpack('f', struct.pack('f', v))[0]
        beta1, beta2 = (f32(b) for b in group['betas'])
        step = None
        captured = getattr(self, '_captured_params', None)
        for p in (captured if captured is not None else group['params']):
            st = self.state.get(p)
            if not st:                       # never received a gradient: no optimizer state, not part of the update
                continue
            st['step'] = int(st['step']) + 1
            step = st['step']
            torch.autograd.graph.increment_version(p)
        if step is None:
            raise RuntimeError('hyper_values: no parameter carries optimizer state yet (run eager steps first)')
        bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
        return [f32(group['lr']) / bc1, 1.0 / math.sqrt(bc2), float(grad_scale)]

    @torch.no_grad()
    def step_captured(self, hyper):
        """The update as it is captured into a graph: same kernel, scalars from ``hyper`` (device, 3 floats).  All
        parameters must already carry optimizer state (run eager steps first) and share one step count."""
        if len(self.param_groups) != 1:
            raise RuntimeError('step_captured: one param group')
        group = self.param_groups[0]
        ps = [p for p in group['params'] if p.grad is not None]
        steps = {int(self.state[p]['step']) for p in ps}
        if len(steps) != 1:
            raise RuntimeError('step_captured: parameters at different step counts')
        if getattr(self, '_captured_params', None) is not None and \
                [id(p) for p in self._captured_params] != [id(p) for p in ps]:
            raise RuntimeError('step_captured: a second capture over a different parameter set (the step counts of the '
                               'two replays would drift apart)')
        self._captured_params = ps
        beta1, beta2 = group['betas']
        ops.adam_step_dev([p.data for p in ps], [p.grad.data for p in ps], [self.state[p]['exp_avg'] for p in ps],
                          [self.state[p]['exp_avg_sq'] for p in ps], hyper, beta1, beta2, group['eps'])
