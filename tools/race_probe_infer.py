"""Run-to-run determinism probe of the inference executor (configs[1] / configs[2] workload): one serial_exec=1 run, then N runs of
the concurrent executor; every run that is not bit-equal to the first concurrent run is reported with where (frame range of the
differing samples -> crop) and how much.  Usage: race_probe_infer.py [runs] [tta]   (VR_* environment toggles apply)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
from oracle import separator, weights  # noqa: E402  (seeded inputs only)

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tta = len(sys.argv) > 2 and sys.argv[2] == '1'
vr = __graft_entry__.load_package()
model = vr.nets.CascadedNet(2048, 1024, 32, 128)
model.load_state_dict(weights.make_state_dict(1234))
model.to(torch.device('cuda:0'))
model.eval()
wave = separator.synth_wave(30.0, seed=0)
wd = torch.from_numpy(wave).to('cuda:0')
sp = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=0, cropsize=256)
model.set_option('serial_exec', 1)
ys, vs = [t.cpu().numpy() for t in sp.separate_wave(wd, tta=tta)]
model.set_option('serial_exec', 0)
first = None
bad = 0
print('toggles:', {k: v for k, v in os.environ.items() if k.startswith('VR_')}, 'tta', tta)
for r in range(runs):
    y, v = [t.cpu().numpy() for t in sp.separate_wave(wd, tta=tta)]
    if first is None:
        first = (y, v)
        print('first concurrent run vs serial: max |dy| %.3e' % np.abs(y - ys).max())
        continue
    d = np.abs(y - first[0]).max(axis=0)
    if d.max() > 0:
        bad += 1
        idx = np.nonzero(d)[0]
        print('run %d differs: %d samples, sample %d..%d (frames %d..%d), max %.3e' % (r, idx.size, idx[0], idx[-1], idx[0] // 1024,
                                                                                      idx[-1] // 1024, d.max()))
print('%d of %d runs differ from the first concurrent run' % (bad, runs - 1))
