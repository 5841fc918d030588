"""Independent bf16 witnesses for configs[4].  TEST INFRASTRUCTURE (only tests/ may import this).

Question ("next round" item 6): the library's `mfma_mode` 1 -- MFMA operands rounded to bf16, fp32 accumulation and
storage -- gave a gradient whose direction is only weakly correlated with the fp32 gradient on the randomly initialised
CascadedNet (global cosine 0.37 at batch 16).  Is that the arithmetic, or a bug in the bf16 kernels?  The GPU cannot answer that
about itself, so the same train step (oracle/train_step.py == train.py:77-96) is evaluated here on the CPU in two bf16 forms that
share no code with the library:

* ``operands``: every convolution (and the LSTM / Linear matrix products) sees its two operands rounded to bf16 (round to nearest
  even, ``tensor.bfloat16().float()``) in the forward pass AND in both backward products (dz and w for the data gradient, dz and x
  for the weight gradient); products and sums are fp32 -- a bf16 x bf16 product is exact in fp32, so this is the arithmetic of a
  bf16 matrix pipe with fp32 accumulation, with fp32 storage of every tensor.  ``which='s1'`` rounds only the 3x3 stride-1
  convolutions (the layers `mfma_mode` 1 touches, 84 % of the multiply-adds), ``which='all'`` every convolution.
* ``autocast``: ``torch.autocast('cpu', dtype=torch.bfloat16)`` around the forward pass -- torch's own mixed precision: bf16
  operands AND bf16 activations out of every conv / linear, fp32 master weights.

`compare()` gives the global and per-tensor gradient cosines against an fp32 evaluation of the same step.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cascaded_net, train_step


def _r(t):
    return t.bfloat16().float()


class _ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, dil):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil)
        return F.conv2d(_r(x), _r(w), None, stride, pad, dil)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        stride, pad, dil = ctx.cfg
        dzr = _r(dz)
        dx = torch.nn.grad.conv2d_input(x.shape, _r(w), dzr, stride, pad, dil) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(_r(x), w.shape, dzr, stride, pad, dil) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


class _FShim(object):
    """Stands in for torch.nn.functional inside oracle.cascaded_net: conv2d with bf16 operands, everything else untouched."""

    def __init__(self, which):
        self.which = which
        self.rounded = 0

    def __getattr__(self, name):
        return getattr(F, name)

    def conv2d(self, x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        norm = lambda v: (v, v) if isinstance(v, int) else tuple(v)     # noqa: E731
        stride, padding, dilation = norm(stride), norm(padding), norm(dilation)
        s1 = w.shape[2] == 3 and stride == (1, 1) and dilation == (1, 1)
        if groups != 1 or (self.which == 's1' and not s1):
            return F.conv2d(x, w, bias, stride, padding, dilation, groups)
        self.rounded += 1
        out = _ConvBf16Operands.apply(x, w, stride, padding, dilation)
        return out if bias is None else out + bias.view(1, -1, 1, 1)


def loss_and_grads(sd, X, y, n_fft, dropout, form, which='all'):
    """One train step evaluation (no running-statistics update).  form: 'fp32' | 'operands' | 'autocast'."""
    if form == 'fp32':
        return train_step.loss_and_grads(sd, X, y, n_fft=n_fft, dropout=dropout, update_running=False)
    if form == 'autocast':
        with torch.autocast('cpu', dtype=torch.bfloat16):
            keys = train_step.param_keys(sd)
            leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
            work = dict(sd)
            work.update(leaves)
            mask = cascaded_net.forward(X, work, n_fft, training=True, update_running=False, dropout=dropout)
        loss = F.l1_loss(mask.float() * X, y)
        loss.backward()
        return float(loss.detach()), {k: v.grad for k, v in leaves.items() if v.grad is not None}
    assert form == 'operands', form
    shim = _FShim(which)
    saved = cascaded_net.F
    cascaded_net.F = shim
    try:
        out = train_step.loss_and_grads(sd, X, y, n_fft=n_fft, dropout=dropout, update_running=False)
    finally:
        cascaded_net.F = saved
    assert shim.rounded > 0
    return out


def compare(ref, other, min_numel=64):
    """-> (global cosine, |other| / |ref|, per-tensor cosines sorted ascending [(cos, numel, key)]).  Tensors whose exact gradient is
    zero (the dense bias in front of a batch-statistics BatchNorm) are left out, as in tests/test_gpu_b16.py."""
    dot = na = nb = 0.0
    rows = []
    for k, a in ref.items():
        if k.endswith('dense.0.bias') or float(a.norm()) == 0.0 or k not in other:
            continue
        a, b = a.double().flatten(), other[k].double().flatten()
        dot += float(a @ b)
        na += float(a @ a)
        nb += float(b @ b)
        if a.numel() >= min_numel:
            rows.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), a.numel(), k))
    rows.sort()
    return dot / (na ** 0.5 * nb ** 0.5), (nb / na) ** 0.5, rows


def summary(ref, other):
    g, ratio, rows = compare(ref, other)
    return {'global_cosine': g, 'norm_ratio': ratio, 'tensor_cosine_min': rows[0][0],
            'tensor_cosine_median': float(np.median([r[0] for r in rows])), 'worst': rows[:5]}
