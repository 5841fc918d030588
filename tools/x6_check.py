"""Accuracy of the three multiply modes of the Winograd kernels on single convs (run on the GPU box):
fp64 torch reference vs mfma_mode 0 (v_mfma_f32_32x32x2_f32), 1 (bf16 operands), 2 (six bf16 products of split operands)."""
import ctypes
import importlib
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
vr = importlib.import_module('vocal-remover_amd')
nat = vr.native
from oracle import weights

sd = weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32)
model = vr.nets.CascadedNet(512, 256, 8, 32)
model.load_state_dict(sd)
model.to(torch.device('cuda:0'))
model.eval()
h = model._handle

CASES = [(1, 64, 16, 32, 64), (2, 97, 16, 64, 32), (1, 192, 16, 32, 192), (1, 32, 32, 64, 128), (2, 26, 40, 48, 32),
         (1, 320, 24, 64, 128), (1, 448, 16, 32, 192), (1, 61, 17, 36, 64)]
for (N, Cin, H, W, Cout) in CASES:
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g) * torch.exp(torch.randn(N, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    want = F.conv2d(x.double(), w.double(), None, 1, 1).numpy()
    scale = np.abs(want).max()
    row = []
    for mode in (0, 1, 2):
        model.set_option('mfma_mode', mode)
        got = np.empty(want.shape, np.float32)
        xn, wn = x.numpy(), w.numpy()
        for flags in (2, 0) if mode == 0 else (2,):
            nat.check(nat.lib().vr_debug_conv2d(h.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, 3, 1, 1, 1, flags, None,
                                                ctypes.c_float(1.0), None, nat.np_ptr(got), None))
            e = np.abs(got - want)
            row.append('%s max %.2e rms %.2e' % ('wino%d' % mode if flags else 'direct', e.max() / scale, np.sqrt((e ** 2).mean()) / scale))
    print((N, Cin, H, W, Cout), ' | '.join(row), flush=True)
model.set_option('mfma_mode', 0)
