"""Race probe for the three-stream train executor: the benched step (full net, batch 16) on ONE stream, then N times on the
concurrent executor; every gradient tensor that differs is reported with the number / index range / size of the differences.
Usage: race_probe.py [mode] [runs]   (VR_NO_BWD_FORK=1 / VR_NO_WGRAD_OVERLAP=1 / VR_NO_BAND_FORK=1 switch single forks off)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
from oracle import train_step, weights  # noqa: E402  (seeded inputs only)

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
vr = __graft_entry__.load_package()
sd = weights.make_state_dict(1234)
model = vr.nets.CascadedNet(2048, 1024, 32, 128)
model.load_state_dict(sd)
model.to(torch.device('cuda:0'))
X, y = train_step.synth_batch(16, T=256, n_fft=2048, seed=3)
X, y = X.to('cuda:0'), y.to('cuda:0')
masks = train_step.dropout_masks(16, seed=5, nout=32)


def step(serial):
    model.load_state_dict(sd)
    model.set_option('mfma_mode', mode)
    model.set_option('serial_exec', 1 if serial else 0)
    model.train()
    model.set_dropout_masks(masks)
    model.zero_grad()
    loss = model.train_step(X, y, 1)
    g = model.grads()
    model.set_option('serial_exec', 0)
    return loss, g


loss_s, g_s = step(True)
print('mode %d  forks: bwd %s wgrad %s band %s' % (mode, os.environ.get('VR_NO_BWD_FORK', 'on'), os.environ.get('VR_NO_WGRAD_OVERLAP', 'on'),
                                                 os.environ.get('VR_NO_BAND_FORK', 'on')))
for r in range(runs):
    loss, g = step(False)
    bad = []
    for k in g_s:
        d = (g[k] - g_s[k]).abs()
        if float(d.max()) > 0:
            idx = torch.nonzero(d.flatten() > 0).flatten()
            bad.append('%s shape %s: %d of %d differ, flat index %d..%d, max %.3e of scale %.3e' % (
                k, tuple(g[k].shape), idx.numel(), d.numel(), int(idx.min()), int(idx.max()), float(d.max()), float(g_s[k].abs().max())))
    print('run %d: loss diff %.3e; %d tensors differ' % (r, abs(loss - loss_s), len(bad)))
    for b in bad[:12]:
        print('   ', b)
    for k in g_s:
        d = (g[k] - g_s[k]).abs().flatten()
        if float(d.max()) > 0:
            idx = torch.nonzero(d > 0).flatten()[:20:4]
            print('      sample (index serial concurrent):', ' '.join('%d %.4e %.4e' % (int(i), float(g_s[k].flatten()[i]), float(g[k].flatten()[i])) for i in idx))
