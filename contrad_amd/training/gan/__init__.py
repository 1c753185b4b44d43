"""Counterpart of training/gan/__init__.py:4-29: ``setup(P)`` resolves ``--mode`` to the loss functions."""
from importlib import import_module


def setup(P):
    if P.mode != 'contrad':
        # std / aug / aug_both / simclr_only are other baselines, outside the hot-path scope
        raise NotImplementedError("training mode '%s' (only --mode=contrad is on the MI355X hot path)" % P.mode)
    mod = import_module('.contrad', __package__)
    P.filename = f"{P.mode}_{P.aug}_L{P.lbd_a}_T{P.temp}"
    P.train_fn = {"G": mod.loss_G_fn, "D": mod.loss_D_fn}
    return P
