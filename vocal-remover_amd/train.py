"""train.py look-alikes (train.py:68-134, 215-218) over the native train step.

    train_epoch(dataloader, model, device, optimizer, accumulation_steps)   train.py:68-105
    validate_epoch(dataloader, model, device)                               train.py:108-134
    Adam(model.parameters(), lr=...)                                        train.py:215-218
    Trainer(model, lr, world_size, rank)       data-parallel step: fwd + L1 + bwd -> all-reduce of the
                                               single flat gradient bucket (RCCL via torch.distributed)
                                               -> fused Adam

The reference's sequence `mask = model(X); loss = crit(mask * X, y); (loss/acc).backward();
optimizer.step(); model.zero_grad()` maps to vr_train_step / vr_adam_step / vr_zero_grad.
"""
import ctypes

import torch

from . import native
from . import spec_utils


class _ParamRef(object):
    """What CascadedNet.parameters() yields: a handle to the native parameter arena."""
    requires_grad = True

    def __init__(self, model):
        self.model = model


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) with the reference's defaults (train.py:215-218), executed by the fused
    vr_adam_step over the library's flat parameter / gradient / moment arenas.  It IS a torch Optimizer (one
    param group, a placeholder tensor), so torch.optim.lr_scheduler.ReduceLROnPlateau (train.py:220-227) drives
    `param_groups[0]['lr']` exactly as in the reference."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError('the reference trains with weight_decay=0 (train.py:215-218)')
        refs = [p for p in params if isinstance(p, _ParamRef)]
        if not refs:
            raise ValueError('pass model.parameters() of a vocal_remover_amd CascadedNet')
        self.model = refs[0].model
        self.grad_scale = 1.0
        self._placeholder = torch.zeros(1, requires_grad=True)
        super().__init__([self._placeholder], dict(lr=lr, betas=betas, eps=eps))
        self.model.set_option('adam_reset', 1)      # a new optimizer starts without moments, like torch.optim.Adam

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        h = self.model._need_handle()
        native.check(native.lib().vr_adam_step(h.h, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                                               float(g['eps']), float(self.grad_scale)))
        self.model._host_stale = True
        return loss

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()


def grad_bucket(model):
    """The flat fp32 gradient arena as a torch CUDA tensor (zero copy) for the all-reduce."""
    h = model._need_handle()
    ptr, n = ctypes.c_void_p(), ctypes.c_int64()
    native.check(native.lib().vr_grad_arena(h.h, ctypes.byref(ptr), ctypes.byref(n)))

    class _Arr(object):
        __cuda_array_interface__ = {'shape': (int(n.value),), 'typestr': '<f4', 'data': (int(ptr.value), False),
                                    'version': 2}
    return torch.as_tensor(_Arr(), device=torch.device('cuda', h.device))


def allreduce_mean_(bucket, world_size):
    """Sum-all-reduce a flat gradient bucket in place; the 1/world is folded into Adam's grad_scale.
    Works on any backend (nccl = RCCL on the GPUs, gloo in the CPU tests)."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


class Trainer(object):

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, world_size=1, rank=0, dropout_seed=None):
        self.model = model
        self.world_size = world_size
        self.rank = rank
        self.opt = Adam(model.parameters(), lr=lr, betas=betas, eps=eps)
        model.train()
        h = model._need_handle()
        if dropout_seed is not None:
            native.check(native.lib().vr_set_dropout(h.h, 1, int(dropout_seed) + rank, None, 0))
        self._bucket = grad_bucket(model) if world_size > 1 else None
        model.zero_grad()

    def step(self, X, y, accumulation_steps=1):
        loss = self.model.train_step(X, y, accumulation_steps)
        if self._bucket is not None:
            self.opt.grad_scale = allreduce_mean_(self._bucket, self.world_size)
            torch.cuda.current_stream().synchronize()
        self.opt.step()
        self.model.zero_grad()
        return loss


def train_epoch(dataloader, model, device, optimizer, accumulation_steps):
    """train.train_epoch (train.py:68-105), same control flow."""
    model.train()
    sum_loss = 0
    itr = -1
    for itr, (X_batch, y_batch) in enumerate(dataloader):
        X_batch = X_batch.to(device)
        y_batch = y_batch.to(device)
        loss = model.train_step(X_batch, y_batch, accumulation_steps)
        if (itr + 1) % accumulation_steps == 0:
            optimizer.step()
            model.zero_grad()
        sum_loss += loss * len(X_batch)
    if itr >= 0 and (itr + 1) % accumulation_steps != 0:
        optimizer.step()
        model.zero_grad()
    return sum_loss / len(dataloader.dataset)


def validate_epoch(dataloader, model, device):
    """train.validate_epoch (train.py:108-134): eval, predict, crop_center(y), L1."""
    model.eval()
    sum_loss = 0
    with torch.no_grad():
        for X_batch, y_batch in dataloader:
            X_batch = X_batch.to(device)
            y_batch = y_batch.to(device)
            y_pred = model.predict(X_batch)
            y_batch = spec_utils.crop_center(y_batch, y_pred)
            loss = torch.nn.functional.l1_loss(y_pred, y_batch)
            sum_loss += loss.item() * len(X_batch)
    return sum_loss / len(dataloader.dataset)


def cpu_baseline_train(sd, B=2):
    """CPU oracle train step (fwd + L1 + bwd + Adam) timed on this box's host cores (bench.py)."""
    import os
    import time
    from oracle import train_step as ots, weights as ow
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(n)
    sd = ow.clone_state_dict(sd)
    X, y = ots.synth_batch(B, T=256, n_fft=2048, seed=0)
    opt = ots.Adam(lr=1e-3)
    t0 = time.perf_counter()
    loss, grads = ots.loss_and_grads(sd, X, y)
    opt.step(sd, grads)
    dt = time.perf_counter() - t0
    return {'value': B * 256 / dt, 'unit': 'spectrogram-frames/sec', 'cores': n, 'kind': 'port',
            'sample': 'one oracle train step (autograd over the restated net + restated Adam) at batch %d x '
                      '[2,1025,256], %.1f s wall' % (B, dt)}
