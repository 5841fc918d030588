"""torch-CPU restatement of one optimisation step of train.py.  TEST INFRASTRUCTURE.

Restates the body of ``train_epoch`` (``/root/reference/train.py:68-105``):
``mask = model(X)``; ``loss = L1Loss()(mask * X, y)``;
``(loss / accumulation_steps).backward()``; ``optimizer.step()`` with
``torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)``
(``train.py:215-218``); ``model.zero_grad()``.  Gradients come from autograd over
``cascaded_net.forward``; Adam is restated by hand (bias-corrected, no amsgrad).
Parameters whose gradient is None (``aux_out.weight`` -- never used in
``forward``, lib/nets.py:80) are skipped, like torch's optimizer does.
"""
import math

import torch

from . import cascaded_net

PARAM_KINDS = ('conv', 'bn_w', 'bn_b', 'lstm', 'lin_w', 'lin_b')


def param_keys(sd):
    """Keys of trainable tensors (everything except BN buffers)."""
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


def loss_and_grads(sd, X, y, n_fft=2048, dropout=None, accumulation_steps=1, update_running=True):
    """Forward (train mode) + L1 loss + backward.  Returns (loss, {key: grad})."""
    keys = param_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    mask = cascaded_net.forward(X, work, n_fft, training=True, update_running=update_running,
                                dropout=dropout)
    loss = torch.nn.functional.l1_loss(mask * X, y)
    (loss / accumulation_steps).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return float(loss.detach()), grads


class Adam:
    """torch.optim.Adam defaults, restated (train.py:215-218)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.step_no = 0
        self.m, self.v = {}, {}

    def step(self, sd, grads):
        self.step_no += 1
        bc1 = 1 - self.b1 ** self.step_no
        bc2 = 1 - self.b2 ** self.step_no
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(g)
                self.v[k] = torch.zeros_like(g)
            m, v = self.m[k], self.v[k]
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            sd[k] = sd[k] - (self.lr / bc1) * (m / denom)


def synth_batch(B, T=256, n_fft=2048, seed=0):
    """BASELINE.md section 3 training inputs: X ~ U[0,1), y = X * U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    bins = n_fft // 2 + 1
    X = torch.rand((B, 2, bins, T), generator=g)
    y = X * torch.rand((B, 2, bins, T), generator=g)
    return X, y


def dropout_masks(B, seed, nout=32, p=0.1):
    """Injectable Dropout2d keep-masks for the five ASPP modules (lib/layers.py:90)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, c in (('stg1_low_band_net.0', nout // 2), ('stg1_high_band_net', nout // 4),
                    ('stg2_low_band_net.0', nout), ('stg2_high_band_net', nout // 2),
                    ('stg3_full_band_net', nout)):
        keep = (torch.rand((B, 8 * c), generator=g) >= p).float() / (1 - p)
        out[name + '.aspp'] = keep
    return out
