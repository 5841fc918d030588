"""Probe how the CPU oracle scales with torch threads on this box (cgroup quotas make
os.cpu_count() a bad default).  Prints seconds per 256-frame crop for a few thread counts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cascaded_net, weights  # noqa: E402

sd = weights.make_state_dict(1234)
x = torch.rand(2, 2, 1025, 256)
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch default', torch.get_num_threads())
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        cascaded_net.predict_mask(x[:1], sd)
        t0 = time.perf_counter()
        cascaded_net.predict_mask(x, sd)
        dt = time.perf_counter() - t0
    print('threads %d: %.2f s for 2 crops -> %.1f computed frames/s' % (th, dt, 512 / dt), flush=True)
