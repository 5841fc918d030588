"""VERDICT r5 item 5, measured: a 3x3 stride-2 conv as a stride-1 conv over the space-to-depth input (4 Cin channels at half resolution)
on conv_x3h.  The existing kernel runs the proxy with all nine taps (zero-padded weights would make it exact; a dedicated 2x2-window
variant would issue 6 of the 14 matrix-instruction groups per chunk) -- its time is the UPPER bound of the reformulation, and the time
with the matrix instructions skipped... is not available, so the lower bound is computed from the phase split of profiles/r05_x3h_phase_trace.txt
(multiply phase 1.8 k of 7.5 k cycles per chunk).  Compared with conv_dma_kernel<3, 2, ...> on the real layer.  Per-kernel times from the
library's launch profiler (HIP events on the launch stream)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

vr = __graft_entry__.load_package()
from oracle import weights  # noqa: E402

sd = weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32)
m = vr.nets.CascadedNet(512, 256, 8, 32)
m.load_state_dict(sd)
m.to(torch.device('cuda:0'))
nat, h = vr.native, m._handle.h


def timed(fn, reps=3):
    best = {}
    for _ in range(reps):
        nat.check(nat.lib().vr_profile_begin(h))
        try:
            fn()
        finally:
            a, b, c, d = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
            nat.check(nat.lib().vr_profile_end(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        need = nat.lib().vr_profile_report(h, None, 0)
        buf = ctypes.create_string_buffer(int(need) + 1)
        nat.lib().vr_profile_report(h, buf, need)
        for ln in buf.value.decode().splitlines():
            f = ln.split('\t')
            if 'conv_' in f[0] and 'weights' not in f[0] and 'wscale' not in f[0]:
                ms = float(f[2])
                best[f[0]] = min(best.get(f[0], 1e9), ms)
    return best


def conv(N, Cin, H, W, Cout, stride, flags):
    rng = np.random.default_rng(Cin + Cout)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / (Cin * 9) ** 0.5).astype(np.float32)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = np.empty((N, Cout, Ho, Wo), np.float32)
    epi = np.stack([np.ones(Cout, np.float32), np.zeros(Cout, np.float32)], 1).copy()
    return timed(lambda: nat.check(nat.lib().vr_debug_conv2d(h, nat.np_ptr(x), N, Cin, H, W, nat.np_ptr(w), Cout, 3, stride, 1, 1, flags | 4,
                                                              nat.np_ptr(epi), ctypes.c_float(0.01), None, nat.np_ptr(out), None)))


N = 6          # crops per lane of the S30 inference step
LAYERS = [('stg3.enc2.conv1', 32, 1024, 256, 64), ('stg3.enc3.conv1', 64, 512, 128, 128), ('stg3.enc4.conv1', 128, 256, 64, 192),
          ('stg2l.enc2.conv1', 32, 512, 256, 64)]
m.set_option('mfma_mode', 3)
for name, Cin, H, W, Cout in LAYERS:
    real = conv(N, Cin, H, W, Cout, 2, 0)
    proxy = conv(N, 4 * Cin, H // 2, W // 2, Cout, 1, 2)
    gf = 2.0 * N * (H // 2) * (W // 2) * Cout * Cin * 9 / 1e9
    (kr, tr), = real.items()
    (kp, tp), = proxy.items()
    # the proxy's multiply phase is 14 groups per chunk; a 2x2-window kernel issues 6: at most 8/14 of the multiply share (0.24 of the chunk) goes away
    lo = tp * (1.0 - 0.24 * 8.0 / 14.0)
    print('%-18s N=%d %3d->%3d @%dx%d  %.2f GF | stride-2 fp32 pipe %-52s %.3f ms %5.1f TF | space-to-depth on %-45s %.3f ms (9 taps, upper bound) .. %.3f ms (4-tap estimate) = %5.1f .. %5.1f TF | ratio %.2f .. %.2f'
          % (name, N, Cin, Cout, H // 2, W // 2, gf, kr.replace('vr::', ''), tr, gf / tr, kp.replace('vr::', ''), tp, lo, gf / tp, gf / lo, tp / tr, lo / tr))
