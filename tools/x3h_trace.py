#!/usr/bin/env python
"""Where a chunk of conv_x3h_kernel goes (diagnostics; run on the GPU box with VR_CONV_DBG=64 -- tools/gpu_call2.sh).

The TRACE build of the kernel stamps s_memtime at the phase boundaries of chunks 2..7 in four workgroups from the middle of the grid,
wave by wave (conv_x3h.hip: g_x3h_trace).  This script runs one layer at a time through vr_debug_conv2d and prints, per layer, the
mean cycles between consecutive stamps:
    0 -> 1   multiply phase (14 matrix-instruction groups; operand reads; weight DMA issue and pixel-load issue for later chunks)
    1 -> 2   s_waitcnt for the pixels of chunk k+1
    2 -> 3   chunk maximum (v_max3, DPP, readlane) + low-resolution staging
    3 -> 4   first barrier (every wave done with P(k))
    4 -> 5   shift bookkeeping + split pass (interpolation for upsampled sources, v_fma_mix split, two ds_write_b128 per pixel)
    5 -> 6   s_waitcnt for the weights of chunk k+1 and the LDS stores
    6 -> 7   second barrier (P(k+1) complete)
    7 -> 0'  loop back edge
ticks = s_memtime = shader cycles (MI355X_MICROARCH.md); the script also reports the
kernel's wall time per launch from HIP events for scale."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

LAYERS = [
    # name, N, Cin, H, W (of the array passed in), Cout, up
    ('stg3.dec1-like 96->32 @1024x256, all sources upsampled  <32,16,UP>', 11, 96, 512, 128, 32, 1),
    ('96->32 @1024x256 plain                                 <32,16>', 11, 96, 1024, 256, 32, 0),
    ('stg3.enc2b 64->64 @512x128                              <64,8>', 11, 64, 512, 128, 64, 0),
    ('stg3.dec2-like 192->64 @512x128 upsampled               <64,8,UP>', 11, 192, 256, 64, 64, 1),
    ('stg1_low.enc2b 32->32 @256x128                          <32,8>', 11, 32, 256, 128, 32, 0),
    ('stg3.enc3b 128->128 @256x64                             <64,8>', 11, 128, 256, 64, 128, 0),
]
NAMES = ['multiply phase', 'wait pixels(k+1)', 'chunk max + lowres staging', 'barrier 1', 'shift + split pass', 'wait weights(k+1) + LDS stores',
         'barrier 2', 'back edge']


def main():
    assert int(os.environ.get('VR_CONV_DBG', '0')) & 64, 'run with VR_CONV_DBG=64 (or 96 to add s_setprio)'
    vr = __graft_entry__.load_package()
    nat = vr.native
    model = vr.nets.CascadedNet(2048, 1024, 32, 128)
    model.to(torch.device('cuda:0'))
    model.set_option('mfma_mode', 3)
    h = model._handle
    rng = np.random.default_rng(0)
    for name, N, Cin, H, W, Cout, up in LAYERS:
        x = rng.random((N, Cin, H, W), dtype=np.float32)
        w = ((rng.random((Cout, Cin, 3, 3), dtype=np.float32) - 0.5) / np.sqrt(Cin * 9.0)).astype(np.float32)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        out = np.empty((N, Cout, Ho, Wo), np.float32)
        nat.debug_kernel(h, 'x3h_trace', [0], [], [], [])
        t0 = time.perf_counter()
        nat.check(nat.lib().vr_debug_conv2d(h.h, nat.np_ptr(x), N, Cin, H, W, nat.np_ptr(w), Cout, 3, 1, 1, 1, (1 if up else 0) | 2, None,
                                            ctypes.c_float(1.0), None, nat.np_ptr(out), None))
        t1 = time.perf_counter()
        tr = np.empty(768, np.float32)
        nat.debug_kernel(h, 'x3h_trace', [1], [], [], [tr])
        tr = tr.reshape(4, 4, 6, 8).astype(np.float64)
        ok = (tr >= 0).all(axis=3)
        print('\n== %s  (N=%d; host call %.0f ms incl. copies) ==' % (name, N, (t1 - t0) * 1e3))
        if not ok.any():
            print('   no stamps (fewer than 8 chunks, or another kernel took the launch)')
            continue
        seg = np.diff(tr, axis=3)                                   # [wg][wave][chunk][7]
        back = tr[:, :, 1:, 0] - tr[:, :, :-1, 7]                   # 7 -> next chunk's 0
        per = [seg[..., i][ok].mean() for i in range(7)] + [back[ok[:, :, 1:] & ok[:, :, :-1]].mean()]
        total = sum(per)
        for nm, v in zip(NAMES, per):
            print('   %-34s %9.1f ticks  %5.1f %%' % (nm, v, 100.0 * v / total))
        print('   %-34s %9.1f ticks per chunk (shader cycles; %.2f us at 2.4 GHz)' % ('total', total, total / 2400.0))
        # spread between the waves of a workgroup at the barriers: who arrives last?
        arr = tr[:, :, :, 3]                                        # arrival at barrier 1
        late = (arr - arr.min(axis=1, keepdims=True))[ok]
        print('   arrival spread at barrier 1 across the 4 waves: mean %.1f ticks, max %.1f' % (late.mean(), late.max()))


if __name__ == '__main__':
    main()
