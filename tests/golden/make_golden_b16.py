#!/usr/bin/env python
"""Fixture for tests/test_gpu_b16.py::test_b16_small_gradients_vs_fp64_fixture (VERDICT r4 "next round" item 7).

At the benched train configuration (batch 16 x [2,1025,256], the weights / inputs / Dropout2d masks of the `full16` fixture) the GPU step
was only compared with the fp32 CPU oracle, and the ten 1-element BatchNorm gradients of the LSTM squeeze convs only as one vector: two
fp32 evaluations of a cancellation-dominated sum differ by its full size, so a sign error in one of them would have passed.  This
script evaluates the SAME step once in fp64 (oracle/train_step.py -- pinned to the reference's train.py:77-96 in fp64 at <= 1e-9 on every
gradient by tests/test_oracle_vs_reference.py) and once in fp32, and stores, for every trainable tensor: the fp64 gradient norm and the
fp32 CPU oracle's own error against fp64; for every tensor of at most 4096 elements (all BatchNorm weights / biases, the dense biases,
the 1-element tensors) the fp64 gradient itself.

Needs ~85 GB of host memory for the fp64 evaluation (the GPU box's host has 318 GB; the build container has 62 GB), so it is run
once on the GPU box's HOST CPU:  gpurun -- 'python tests/golden/make_golden_b16.py gpurun_out/b16_fp64_small_grads.npz'
and the result is committed as tests/golden/b16_fp64_small_grads.npz.  No GPU and no /root/reference involved."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import train_step, weights  # noqa: E402

SMALL = 4096


def main(out_path, batch=16):
    import bench
    torch.set_num_threads(bench.usable_cores())
    sd = weights.make_state_dict(1234)
    X, y = train_step.synth_batch(batch, T=256, n_fft=2048, seed=3)
    masks = train_step.dropout_masks(batch, seed=5, nout=32)
    t0 = time.time()
    loss32, g32 = train_step.loss_and_grads(weights.clone_state_dict(sd), X, y, dropout=masks, update_running=False)
    t1 = time.time()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    m64 = {k: v.double() for k, v in masks.items()}
    loss64, g64 = train_step.loss_and_grads(sd64, X.double(), y.double(), dropout=m64, update_running=False)
    t2 = time.time()
    keys = sorted(g64)
    out = {'loss64': np.float64(loss64), 'loss32': np.float64(loss32), 'batch': np.int64(batch), 'keys': np.array(keys),
           'norm64': np.array([float(g64[k].norm()) for k in keys]),
           'cpu32_err': np.array([float((g32[k].double() - g64[k]).norm() / max(float(g64[k].norm()), 1e-300)) for k in keys]),
           'numel': np.array([g64[k].numel() for k in keys])}
    for k in keys:
        if g64[k].numel() <= SMALL:
            out['g64/' + k] = g64[k].numpy().astype(np.float64)
    np.savez_compressed(out_path, **out)
    small = [k for k in keys if g64[k].numel() < 16]
    print('batch %d: fp32 %.1f s, fp64 %.1f s; loss fp32 %.9f fp64 %.12f; %d tensors, %d stored in full, %d with < 16 elements'
          % (batch, t1 - t0, t2 - t1, loss32, loss64, len(keys), sum(1 for k in out if k.startswith('g64/')), len(small)))
    for k in small:
        i = keys.index(k)
        print('  %-55s fp64 %+.6e   fp32 CPU oracle %+.6e   (rel err %.2e)' % (k, float(g64[k].flatten()[0]), float(g32[k].flatten()[0]), out['cpu32_err'][i]))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'b16_fp64_small_grads.npz'),
         int(sys.argv[2]) if len(sys.argv) > 2 else 16)
