"""World-size-2 gloo test of the data-parallel step's host logic (the GPU path uses the same code
with backend nccl = RCCL): one SUM all-reduce of the single flat gradient bucket, 1/world folded
into Adam's grad_scale; and the DP == gradient-accumulation identity the design relies on
 checked on the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__
from oracle import train_step, weights


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    vr = __graft_entry__.load_package()
    from vocal_remover_amd import train as vtrain
    g = torch.Generator().manual_seed(100 + rank)
    bucket = torch.randn(1000, generator=g)           # this rank's flat gradient bucket
    mine = bucket.clone()
    scale = vtrain.allreduce_mean_(bucket, world)
    others = [torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = sum(others) / world
    ok = torch.allclose(bucket * scale, want, atol=1e-6) and not torch.equal(bucket, mine)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    __graft_entry__.build()
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_dp_equals_gradient_accumulation_on_the_oracle():
    """N-rank DP with per-replica BatchNorm statistics and gradient averaging == the reference's own
    accumulation_steps=N loop (train.py:91-96) -- the parity definition for the multi-GPU path."""
    sd = weights.make_state_dict(3, n_fft=512, nout=8, nout_lstm=32)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    X, y = train_step.synth_batch(4, T=64, n_fft=512, seed=1)
    X, y = X.double(), y.double()
    # reference semantics: two micro-batches, each loss scaled by 1/2, grads summed
    acc = None
    for i in range(2):
        _, g = train_step.loss_and_grads(dict(sd64), X[2 * i:2 * i + 2], y[2 * i:2 * i + 2], n_fft=512,
                                         accumulation_steps=2, update_running=False)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    # DP semantics: each rank computes the un-scaled gradient of its shard, all-reduce SUM, * 1/world
    shards = [train_step.loss_and_grads(dict(sd64), X[2 * r:2 * r + 2], y[2 * r:2 * r + 2], n_fft=512,
                                        update_running=False)[1] for r in range(2)]
    for k in acc:
        dp = (shards[0][k] + shards[1][k]) * 0.5
        assert float((dp - acc[k]).abs().max()) < 1e-12, k
