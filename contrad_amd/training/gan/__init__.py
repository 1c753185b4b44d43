"""Counterpart of training/gan/__init__.py:4-29: ``setup(P)`` resolves ``--mode`` to the loss functions."""
from importlib import import_module


def setup(P):
    if P.mode == 'contrad':
        P.filename = f"{P.mode}_{P.aug}_L{P.lbd_a}_T{P.temp}"
    elif P.mode == 'simclr_only':
        P.filename = f"{P.mode}_{P.aug}_T{P.temp}"
    else:
        # std / aug / aug_both are the non-contrastive baselines, outside the hot-path scope
        raise NotImplementedError("training mode '%s' (contrad and simclr_only are on the MI355X hot path)" % P.mode)
    mod = import_module('.' + P.mode, __package__)
    P.train_fn = {"G": mod.loss_G_fn, "D": mod.loss_D_fn}
    return P
