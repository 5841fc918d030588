"""Neighbour test for DESIGN.md "hardware fact 5" (VERDICT r3 item 9): every kernel of the training path that consumes LDS reads
behind compiler-counted `s_waitcnt lgkmcnt(N > 0)` -- BatchNorm backward (reduce / finalize / apply), the BiLSTM forward, BPTT and
W_hh gradient, the tiled bilinear-upsample transpose, the pooled / rows passes -- runs repeatedly on ONE handle while a second
handle on the same GPU, driven from another host thread, keeps launching conv_x3_kernel<64,8> / <32,8> (mfma_mode 2: the tilings
that disturbed the round-3 lstm_whh_grad kernel in every run) or the fp32 Winograd kernel (mfma_mode 0).  Each victim must return,
bit for bit, what it returned alone.  (The aggressor shares the CUs: both handles use their own non-blocking streams.)"""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def f32(t):
    return np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def _victims():
    g = torch.Generator().manual_seed(7)
    v = {}
    N, C, H, W = 4, 32, 64, 128                                         # BatchNorm backward: vector path, several row chunks
    v['bn_backward'] = ((N, C, H, W), [0.01, 1e-5, 0.1],
                        [f32(torch.randn(N, C, H, W, generator=g)), f32(torch.randn(N, C, H, W, generator=g)), f32(torch.rand(C, generator=g) + 0.5),
                         f32(torch.randn(C, generator=g)), None, f32(torch.zeros(C)), f32(torch.ones(C))],
                        [(N, C, H, W), (C,), (C,), (C, 2), (C,), (C,)])
    N, T, Hh = 16, 256, 64                                              # the benched LSTM shape
    v['lstm'] = ((N, T, Hh), [], [f32(torch.randn(N, 8 * Hh, T, generator=g) * 0.8), f32((torch.rand(4 * Hh, Hh, generator=g) * 2 - 1) / 8),
                                  f32((torch.rand(4 * Hh, Hh, generator=g) * 2 - 1) / 8), f32(torch.randn(N, 2 * Hh, T, generator=g))],
                 [(N, 2 * Hh, T), (N, 8 * Hh, T), (4 * Hh, Hh), (4 * Hh, Hh)])
    N, C, H, W = 4, 16, 64, 64                                          # tiled transpose of the bilinear x2
    v['upsample'] = ((N, C, H, W), [], [f32(torch.randn(N, C, H, W, generator=g)), f32(torch.randn(N, C, 2 * H, 2 * W, generator=g))],
                     [(N, C, 2 * H, 2 * W), (N, C, H, W)])
    N, C, H, W = 4, 64, 32, 16
    v['pool'] = ((N, C, H, W), [], [f32(torch.randn(N, C, H, W, generator=g)), f32(torch.randn(N, C, W, generator=g)), f32(torch.randn(N, C, H, W, generator=g))],
                 [(N, C, W), (N, C, H, W), (N, C, W)])
    N, R, W = 8, 256, 128
    v['rows'] = ((N, R, W), [], [f32(torch.randn(N, R, W, generator=g)), f32(torch.rand(R, 2, generator=g)), f32(torch.randn(N, R, W, generator=g))],
                 [(N, R, W), (R,)])
    return v


def _run(nat, h, name, spec):
    dims, fp, ins, outs = spec
    out = [np.empty(s, np.float32) for s in outs]
    nat.debug_kernel(h, name, list(dims), fp, ins, out)
    return out


@pytest.mark.parametrize('mode', [2, 3, 0], ids=['beside_conv_x3', 'beside_conv_x3h', 'beside_conv_wino'])
def test_lds_consumers_are_bit_stable_beside_the_conv_kernels(vr, mode):
    nat = vr.native
    victim = vr.nets.CascadedNet(512, 256, 8, 32)
    victim.to(torch.device('cuda:0'))
    # aggressor: the full CascadedNet forward on 4 crops -- ~6 ms of back-to-back conv kernels per call, conv_x3_kernel<64,8>, <32,16> and
    # <32,8> among them in mfma_mode 2 (conv_wino / conv_dma in mode 0), on its own streams
    aggr = vr.nets.CascadedNet(2048, 1024, 32, 128)
    aggr.to(torch.device('cuda:0')).eval()
    aggr.set_option('mfma_mode', mode)
    xa = torch.rand(4, 2, 1025, 256, generator=torch.Generator().manual_seed(1)).to('cuda:0')
    aggr.predict_mask(xa)                                               # warm-up: allocations, weight tables
    specs = _victims()
    alone = {k: _run(nat, victim._handle, k, s) for k, s in specs.items()}
    for k, s in specs.items():                                           # (alone it is deterministic to begin with)
        again = _run(nat, victim._handle, k, s)
        assert all(np.array_equal(a, b) for a, b in zip(alone[k], again)), k
    stop, launches, errors = threading.Event(), [0], []

    def hammer():
        try:
            while not stop.is_set():
                aggr.predict_mask(xa)
                launches[0] += 1
        except Exception as e:                                           # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=hammer)
    t.start()
    try:
        bad = []
        for rep in range(12):
            for k, s in specs.items():
                got = _run(nat, victim._handle, k, s)
                for i, (a, b) in enumerate(zip(alone[k], got)):
                    if not np.array_equal(a, b):
                        bad.append('%s output %d, repetition %d: max diff %.3e' % (k, i, rep, float(np.abs(a - b).max())))
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    assert launches[0] >= 4, 'the aggressor did not run beside the victims'
    print('mode %d: %d aggressor forwards (~100 conv launches each) ran beside 12 x %d victim kernels' % (mode, launches[0], len(specs)))
    assert not bad, '\n'.join(bad[:20])
