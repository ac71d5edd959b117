"""losses — L1 / L2 wrappers with the reference's reduction semantics (nerfactor/losses.py:20-46):
keras MeanSquaredError(reduction='none') averages the last axis; `keep_batch=True` then averages
every remaining axis but the first."""
import torch


def _reduce(per_elem, keep_batch):
    loss = per_elem.mean(-1)
    if keep_batch:
        return loss.reshape(loss.shape[0], -1).mean(-1) if loss.ndim > 1 else loss
    return loss.mean()


class L1:
    def __call__(self, gt, pred, weights=None):
        loss = (gt - pred).abs().mean(-1)
        if weights is not None:
            loss = loss * weights
        return loss.mean()


class L2:
    def __call__(self, gt, pred, keep_batch=False, weights=None):
        per = (gt - pred) ** 2
        loss = per.mean(-1)
        if weights is not None:
            loss = loss * weights
        if keep_batch:
            return loss.reshape(loss.shape[0], -1).mean(-1) if loss.ndim > 1 else loss
        return loss.mean()
