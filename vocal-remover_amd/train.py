"""train.py look-alikes (train.py:68-134, 215-218) over the native train step.

    train_epoch(dataloader, model, device, optimizer, accumulation_steps)   train.py:68-105
    validate_epoch(dataloader, model, device)                               train.py:108-134
    Adam(model.parameters(), lr=...)                                        train.py:215-218
    Trainer(model, lr, world_size, rank)       data-parallel step: fwd + L1 + bwd -> all-reduce of the
                                               single flat gradient bucket (the library's own RCCL
                                               communicator, vr_allreduce_grads) -> fused Adam

The reference's sequence `mask = model(X); loss = crit(mask * X, y); (loss/acc).backward();
optimizer.step(); model.zero_grad()` maps to vr_train_step / vr_adam_step / vr_zero_grad.
"""
import ctypes

import torch

from . import native


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) with the reference's defaults (train.py:215-218), executed by the fused
    vr_adam_step over the library's flat parameter / gradient / moment arenas.  It IS a torch Optimizer (one
    param group, a placeholder tensor), so torch.optim.lr_scheduler.ReduceLROnPlateau (train.py:220-227) drives
    `param_groups[0]['lr']` exactly as in the reference.  (torch.optim.Adam itself also works on model.parameters():
    the flat parameter is a zero-copy view of the arena -- that is how the reference's train.py runs unmodified.)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError('the reference trains with weight_decay=0 (train.py:215-218)')
        refs = [p for p in params if getattr(p, '_vr_model', None) is not None]
        if not refs:
            raise ValueError('pass model.parameters() of a vocal_remover_amd CascadedNet')
        self.model = refs[0]._vr_model
        self._handle_gen = self.model._handle_gen      # the moments live in THIS native handle
        self.grad_scale = 1.0
        super().__init__(refs[:1], dict(lr=lr, betas=betas, eps=eps))
        self.model.set_option('adam_reset', 1)      # a new optimizer starts without moments, like torch.optim.Adam

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        h = self.model._need_handle()
        if self._handle_gen != self.model._handle_gen:
            raise RuntimeError('this optimizer belongs to a native handle that model.to(...) has closed since: build a new one')
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
    Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the CPU tests)."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


def comm_init(model, rank, world_size, exchange=None):
    """Create the library's own RCCL communicator on the model's handle (vr_comm_init).

    The 128-byte ncclUniqueId travels from rank 0 to the others over a host channel: `exchange(id_bytes_or_None)
    -> id_bytes`; default = torch.distributed.broadcast_object_list on the default process group (any backend)."""
    h = model._need_handle()
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        native.check(native.lib().vr_comm_unique_id(buf))
    if world_size > 1:
        if exchange is None:
            import torch.distributed as dist
            box = [buf.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            raw = box[0]
        else:
            raw = exchange(buf.raw if rank == 0 else None)
        buf = ctypes.create_string_buffer(raw, 128)
    native.check(native.lib().vr_comm_init(h.h, int(rank), int(world_size), buf))


class Trainer(object):
    """Data-parallel train step of train.py:77-96: fwd + L1 + bwd on every rank's shard, ONE sum-all-reduce of the
    flat gradient arena, fused Adam with grad_scale = 1/world.  N ranks == the reference's gradient accumulation
    with accumulation_steps = N (per-replica BatchNorm statistics, rank-local running buffers).

    backend  'rccl'   the library's own RCCL communicator (vr_allreduce_grads, on the handle's stream, no host
                      round trip before Adam); the default on GPUs
             'torch'  torch.distributed.all_reduce over the zero-copy view of the arena (process group = nccl)
             'staged' the bucket is staged through host memory (process group = gloo): several ranks may then share
                      one GPU, which is how the 2-rank path is tested on a 1-GPU box
             'auto'   world 1: none; nccl process group: 'rccl'; gloo process group: 'staged'
    wire     'fp32' (default) or 'bf16' (rccl backend only: the bucket crosses xGMI in bf16)
    broadcast  rank 0's weights, BatchNorm buffers and Adam state replace everyone's at construction."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, world_size=1, rank=0, dropout_seed=None,
                 dropout=True, backend='auto', wire='fp32', broadcast=True):
        self.model = model
        self.world_size = world_size
        self.rank = rank
        self.opt = Adam(model.parameters(), lr=lr, betas=betas, eps=eps)
        model.train()
        h = model._need_handle()
        # Dropout2d(0.1) of the ASPP outputs (lib/layers.py:90) is ON by default (seeded from torch's seed in
        # CascadedNet.to()); every rank draws its own stream.  dropout=False is the explicit opt-out.
        if not dropout:
            native.check(native.lib().vr_set_dropout(h.h, 0, 0, None, 0))
        elif dropout_seed is not None:
            native.check(native.lib().vr_set_dropout(h.h, 1, (int(dropout_seed) + rank) & (2 ** 63 - 1), None, 0))
        elif rank:
            native.check(native.lib().vr_set_dropout(h.h, 1, (torch.initial_seed() + 7919 * rank) & (2 ** 63 - 1), None, 0))
        if backend == 'auto':
            if world_size == 1:
                backend = 'none'
            else:
                import torch.distributed as dist
                backend = 'rccl' if dist.get_backend() == 'nccl' else 'staged'
        if backend not in ('none', 'rccl', 'torch', 'staged'):
            raise ValueError('unknown backend %r' % (backend,))
        if wire not in ('fp32', 'bf16') or (wire == 'bf16' and backend != 'rccl'):
            raise ValueError("wire must be 'fp32', or 'bf16' with backend='rccl'")
        self.backend = backend
        self.wire = 1 if wire == 'bf16' else 0
        self._bucket = grad_bucket(model) if backend in ('torch', 'staged') else None
        if backend == 'rccl':
            comm_init(model, rank, world_size)
            if broadcast:
                native.check(native.lib().vr_broadcast_params(h.h, 0, 1))
                model._host_stale = True
        elif backend in ('torch', 'staged') and broadcast and world_size > 1:
            import torch.distributed as dist
            box = [model.state_dict() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                model.load_state_dict(box[0])
        model.zero_grad()

    def reduce(self):
        """The exchange step: SUM the gradient arena over the ranks in place; Adam's grad_scale becomes 1/world."""
        if self.backend == 'rccl':
            h = self.model._need_handle()
            native.check(native.lib().vr_allreduce_grads(h.h, self.wire))       # on the handle's stream, no host sync
        elif self.backend == 'torch':
            allreduce_mean_(self._bucket, self.world_size)
            torch.cuda.current_stream().synchronize()
        elif self.backend == 'staged':
            host = self._bucket.cpu()
            allreduce_mean_(host, self.world_size)
            self._bucket.copy_(host)
            torch.cuda.current_stream().synchronize()
        self.opt.grad_scale = 1.0 / self.world_size

    def step(self, X, y, accumulation_steps=1):
        loss = self.model.train_step(X, y, accumulation_steps)
        self.reduce()
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
    """train.validate_epoch (train.py:108-134): eval; per batch y_pred = model.predict(X), y = crop_center(y, y_pred),
    L1 -- one vr_validate_step (forward, crop and the reduction on the device), same accumulation as the reference."""
    model.eval()
    sum_loss = 0
    for X_batch, y_batch in dataloader:
        X_batch = X_batch.to(device)
        y_batch = y_batch.to(device)
        loss = model.validate_step(X_batch, y_batch)
        sum_loss += loss * len(X_batch)
    return sum_loss / len(dataloader.dataset)


# ---- the epoch loop of train.py main() (train.py:272-294) + what the reference lacks: a resumable checkpoint ----------
def save_checkpoint(path, model, optimizer, scheduler=None, epoch=0, best_loss=None, log=None):
    """Model state_dict (the reference's 689 keys) + Adam moments / step + scheduler state + loop position."""
    h = model._need_handle()
    import ctypes
    n = model._arena_tensor(native.lib().vr_grad_arena).numel()
    state = {'model': model.state_dict(), 'epoch': int(epoch), 'best_loss': best_loss, 'log': list(log or []),
             'lr': optimizer.param_groups[0]['lr']}
    if isinstance(optimizer, Adam):
        m = torch.empty(n, dtype=torch.float32)
        v = torch.empty(n, dtype=torch.float32)
        step = ctypes.c_int64()
        native.check(native.lib().vr_get_adam_state(h.h, m.data_ptr(), v.data_ptr(), n, ctypes.byref(step)))
        state['adam'] = {'exp_avg': m, 'exp_avg_sq': v, 'step': int(step.value)}
    else:
        state['optimizer'] = optimizer.state_dict()
    if scheduler is not None:
        state['scheduler'] = scheduler.state_dict()
    torch.save(state, path)


def load_checkpoint(path, model, optimizer, scheduler=None):
    """Inverse of save_checkpoint; returns (next_epoch, best_loss, log)."""
    state = torch.load(path, map_location='cpu')
    model.load_state_dict(state['model'])
    h = model._need_handle()
    if 'adam' in state and isinstance(optimizer, Adam):
        a = state['adam']
        m, v = a['exp_avg'].contiguous(), a['exp_avg_sq'].contiguous()
        native.check(native.lib().vr_set_adam_state(h.h, m.data_ptr(), v.data_ptr(), m.numel(), int(a['step'])))
    elif 'optimizer' in state:
        optimizer.load_state_dict(state['optimizer'])
    optimizer.param_groups[0]['lr'] = state.get('lr', optimizer.param_groups[0]['lr'])
    if scheduler is not None and 'scheduler' in state:
        scheduler.load_state_dict(state['scheduler'])
    return state['epoch'] + 1, state['best_loss'], state['log']


def fit(model, device, train_dataloader, val_dataloader, optimizer, scheduler, epochs, accumulation_steps=1,
        model_dir='models', log_path=None, checkpoint_path=None, logger=None, start_epoch=0, best_loss=None, log=None):
    """train.py:272-294: per epoch train_epoch, validate_epoch, scheduler.step(val_loss), save the model on a new best
    validation loss as `models/model_iter{epoch}.pth` (state_dict, loadable by the reference), append [train, val] to
    the loss json.  Additionally writes a resumable checkpoint (optimizer moments included) when asked."""
    import json
    import numpy as np
    log = list(log or [])
    best_loss = np.inf if best_loss is None else best_loss
    for epoch in range(start_epoch, epochs):
        if logger:
            logger.info('# epoch {}'.format(epoch))
        train_loss = train_epoch(train_dataloader, model, device, optimizer, accumulation_steps)
        val_loss = validate_epoch(val_dataloader, model, device)
        if logger:
            logger.info('  * training loss = {:.6f}, validation loss = {:.6f}'.format(train_loss, val_loss))
        scheduler.step(val_loss)
        if val_loss < best_loss:
            best_loss = val_loss
            if logger:
                logger.info('  * best validation loss')
            import os
            os.makedirs(model_dir, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(model_dir, 'model_iter{}.pth'.format(epoch)))
        log.append([train_loss, val_loss])
        if log_path:
            with open(log_path, 'w', encoding='utf8') as f:
                json.dump(log, f, ensure_ascii=False)
        if checkpoint_path:
            save_checkpoint(checkpoint_path, model, optimizer, scheduler, epoch, best_loss, log)
    return log, best_loss
