"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle.

Tolerances (fp32, stated per test):
  * single conv launch vs torch.nn.functional.conv2d:        1e-4 * max|ref| (K up to 2304)
  * predict_mask / forward mask in [0,1]:                    max-abs 1e-4, mean-abs 1e-5
  * STFT vs float64 restatement:                             2e-5 * max|X|;   iSTFT 2e-5 abs
  * Separator y/v spectrograms:                              1e-4 * max|X|
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cascaded_net, separator, stft_np, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def small(vr):
    n_fft, nout, nout_lstm = 512, 8, 32
    sd = weights.make_state_dict(11, n_fft=n_fft, nout=nout, nout_lstm=nout_lstm)
    model = vr.nets.CascadedNet(n_fft, n_fft // 2, nout, nout_lstm)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0'))
    model.eval()
    return model, sd, n_fft


def _conv_case(vr, handle, N, Cin, H, W, Cout, ks, stride, dh, dw, up, use_aff, slope, use_bias, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    aff = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3], 1) if use_aff else None
    bias = torch.randn(Cout, generator=g) if use_bias else None
    xin = x
    if aff is not None:
        xin = xin * aff[:, 0].view(1, -1, 1, 1) + aff[:, 1].view(1, -1, 1, 1)
    xin = torch.where(xin > 0, xin, xin * slope)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode='bilinear', align_corners=True)
    pad = (dh, dw) if ks == 3 else (0, 0)
    want = F.conv2d(xin, w, bias, stride, pad, (dh, dw))
    got = np.empty(tuple(want.shape), np.float32)
    stats = np.empty((Cout, 2), np.float32)
    xn, wn = x.numpy(), w.numpy()
    an = aff.numpy().copy() if aff is not None else None
    bn = bias.numpy() if bias is not None else None
    nat = vr.native
    nat.check(nat.lib().vr_debug_conv2d(
        handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, ks, stride, dh, dw, int(up),
        nat.np_ptr(an) if an is not None else None, ctypes.c_float(slope),
        nat.np_ptr(bn) if bn is not None else None, nat.np_ptr(got), nat.np_ptr(stats)))
    scale = float(want.abs().max())
    err = float(np.abs(got - want.numpy()).max())
    s1 = want.sum(dim=(0, 2, 3)).numpy()
    s2 = (want.double() ** 2).sum(dim=(0, 2, 3)).numpy()
    e1 = float(np.abs(stats[:, 0] - s1).max() / (np.abs(s1).max() + 1.0))
    e2 = float(np.abs(stats[:, 1] - s2).max() / (np.abs(s2).max() + 1.0))
    return err / scale, e1, e2


CONV_CASES = [
    # N, Cin, H,  W,  Cout, ks, stride, dh, dw, up, aff, slope, bias
    (2, 2, 16, 32, 16, 3, 1, 1, 1, 0, 0, 1.0, 0),
    (1, 10, 24, 64, 32, 3, 1, 1, 1, 0, 1, 0.0, 0),
    (2, 26, 40, 48, 32, 3, 1, 1, 1, 0, 1, 0.01, 0),      # W not a multiple of 32
    (1, 64, 16, 32, 64, 3, 1, 1, 1, 0, 1, 0.0, 0),
    (1, 32, 32, 64, 128, 3, 1, 1, 1, 0, 1, 0.0, 0),
    (3, 17, 20, 16, 48, 3, 1, 1, 1, 0, 1, 0.0, 0),       # TW=16 tiles, odd Cin, Cout=48
    (2, 16, 32, 64, 32, 3, 2, 1, 1, 0, 1, 0.01, 0),      # stride 2
    (1, 33, 34, 36, 96, 3, 2, 1, 1, 0, 1, 0.01, 0),
    (2, 8, 32, 16, 8, 3, 2, 1, 1, 0, 0, 1.0, 0),
    (2, 32, 32, 16, 32, 3, 1, 4, 2, 0, 1, 0.0, 0),       # ASPP dilations
    (1, 64, 64, 16, 64, 3, 1, 8, 4, 0, 1, 0.0, 0),
    (2, 16, 32, 16, 16, 3, 1, 12, 6, 0, 1, 0.0, 0),
    (2, 40, 16, 32, 8, 1, 1, 1, 1, 0, 1, 0.0, 0),        # 1x1
    (1, 320, 32, 16, 64, 1, 1, 1, 1, 0, 1, 0.0, 1),      # 1x1 with bias, K=320
    (1, 128, 5, 64, 256, 1, 1, 1, 1, 0, 0, 1.0, 1),
    (2, 12, 8, 16, 32, 3, 1, 1, 1, 1, 1, 0.0, 0),        # fused bilinear x2 upsample
    (1, 24, 16, 24, 16, 3, 1, 1, 1, 1, 1, 0.0, 0),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_kernel_vs_torch(vr, small, case):
    model = small[0]
    rel, e1, e2 = _conv_case(vr, model._handle, *case, seed=hash(case) % 1000)
    assert rel < 1e-4, 'conv max-abs/scale = %.3e' % rel
    assert e1 < 1e-4 and e2 < 1e-4, 'BatchNorm partial sums off: %.3e %.3e' % (e1, e2)


def _plain_conv_case(vr, handle, N, Cin, H, W, Cout, ks, stride, dh, dw, use_epi, slope, use_bias, wino, seed):
    """Plain input (the eval-mode form): LDS-DMA kernel, or the Winograd kernel when `wino`."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    epi = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3], 1) if use_epi else None
    bias = torch.randn(Cout, generator=g) if use_bias else None
    pad = (dh, dw) if ks == 3 else (0, 0)
    want = F.conv2d(x, w, bias, stride, pad, (dh, dw))
    if epi is not None:
        want = want * epi[:, 0].view(1, -1, 1, 1) + epi[:, 1].view(1, -1, 1, 1)
        want = torch.where(want > 0, want, want * slope)
    got = np.empty(tuple(want.shape), np.float32)
    xn, wn = x.numpy(), w.numpy()
    en = epi.numpy().copy() if epi is not None else None
    bn = bias.numpy() if bias is not None else None
    nat = vr.native
    flags = (2 if wino else 0) | (4 if use_epi else 0)
    nat.check(nat.lib().vr_debug_conv2d(
        handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, ks, stride, dh, dw, flags,
        nat.np_ptr(en) if en is not None else None, ctypes.c_float(slope if use_epi else 1.0),
        nat.np_ptr(bn) if bn is not None else None, nat.np_ptr(got), None))
    return float(np.abs(got - want.numpy()).max()) / float(want.abs().max())


PLAIN_CASES = [
    # N, Cin, H,  W,  Cout, ks, stride, dh, dw, epi, slope, bias
    (2, 2, 16, 32, 16, 3, 1, 1, 1, 0, 1.0, 0),           # Cin < one chunk
    (1, 10, 24, 64, 32, 3, 1, 1, 1, 1, 0.0, 0),
    (2, 26, 40, 48, 32, 3, 1, 1, 1, 1, 0.01, 1),         # W not a multiple of 32, bias + epilogue
    (1, 64, 17, 32, 64, 3, 1, 1, 1, 1, 0.0, 0),          # odd H
    (1, 32, 32, 64, 128, 3, 1, 1, 1, 1, 0.01, 0),
    (2, 97, 16, 64, 32, 3, 1, 1, 1, 1, 0.0, 0),          # dec1-like (Cin = 97)
    (1, 192, 16, 32, 192, 3, 1, 1, 1, 0, 1.0, 0),
    (3, 17, 20, 16, 48, 3, 1, 1, 1, 1, 0.0, 0),          # 16-wide tiles
    (2, 16, 32, 64, 32, 3, 2, 1, 1, 1, 0.01, 0),         # stride 2
    (1, 33, 34, 36, 96, 3, 2, 1, 1, 0, 1.0, 0),
    (2, 8, 32, 32, 8, 3, 2, 1, 1, 1, 0.0, 0),            # stride 2 -> 16-wide output
    (2, 32, 32, 16, 32, 3, 1, 4, 2, 1, 0.0, 0),          # ASPP dilations
    (1, 64, 64, 16, 64, 3, 1, 8, 4, 1, 0.0, 0),
    (2, 16, 32, 16, 16, 3, 1, 12, 6, 0, 1.0, 0),
    (2, 40, 16, 32, 8, 1, 1, 1, 1, 1, 0.0, 0),           # 1x1
    (1, 320, 32, 16, 64, 1, 1, 1, 1, 1, 0.0, 1),         # 1x1, 16-wide, K = 320
    (1, 128, 5, 64, 256, 1, 1, 1, 1, 0, 1.0, 1),
    (2, 49, 24, 64, 16, 3, 1, 1, 1, 1, 0.0, 0),          # <= 16 couts: conv_thin.hip (16x16x4 MFMA), dec1-like
    (1, 25, 17, 96, 8, 3, 1, 1, 1, 1, 0.01, 1),          # 8 couts, odd H, bias + epilogue
    (3, 10, 40, 48, 16, 3, 1, 1, 1, 0, 1.0, 0),          # partial 32-column tile, Cin not a multiple of 4
]


@pytest.mark.parametrize('case', PLAIN_CASES, ids=[str(c) for c in PLAIN_CASES])
def test_conv_dma_plain_input_vs_torch(vr, small, case):
    rel = _plain_conv_case(vr, small[0]._handle, *case, wino=False, seed=hash(case) % 1000)
    assert rel < 1e-4, 'conv max-abs/scale = %.3e' % rel


@pytest.mark.parametrize('case', [c for c in PLAIN_CASES if c[5] == 3 and c[6] == 1 and c[7] == 1 and c[3] >= 32],
                         ids=str)
def test_conv_winograd_vs_torch(vr, small, case):
    """F(2x2,3x3) in fp32: the transforms only add and halve; 1e-4 of the output scale is the same bar as
    the direct kernels (measured ~1e-6)."""
    rel = _plain_conv_case(vr, small[0]._handle, *case, wino=True, seed=hash(case) % 1000)
    assert rel < 1e-4, 'winograd conv max-abs/scale = %.3e' % rel
    direct = _plain_conv_case(vr, small[0]._handle, *case, wino=False, seed=hash(case) % 1000)
    assert abs(rel - direct) < 1e-4


UP_CASES = [
    # N, Cin, H (low-res), W (low-res), Cout, epi, slope, bias      conv input = bilinear x2 of the source (lib/layers.py:52)
    (2, 16, 20, 32, 32, 1, 0.0, 0),
    (1, 40, 9, 24, 64, 1, 0.01, 1),          # odd low-res H, 48 output columns (partial 32-column tile), bias + epilogue
    (2, 8, 64, 64, 16, 0, 1.0, 0),           # 16 couts (padded to one 32-cout tile), 128 x 128 outputs
    (1, 97, 16, 16, 32, 1, 0.0, 0),          # dec1-like channel count, 32 x 32 outputs
    (3, 24, 5, 16, 40, 0, 1.0, 1),           # 10 x 32 outputs: one partial row tile per image
]


@pytest.mark.parametrize('case', UP_CASES, ids=str)
def test_conv_x3_fused_upsample_vs_torch(vr, small, case):
    """The decoder's F.interpolate(x2, bilinear, align_corners=True) fused into the split-bf16 direct kernel (conv_x3.hip: low-
    resolution tile in LDS, interpolation inside the split pass) against torch's upsample + conv2d, and against the library's own
    fp32 fused-loader kernel (mfma_mode 0) -- same bar as every other conv: 1e-4 of the output scale."""
    N, Cin, H, W, Cout, use_epi, slope, use_bias = case
    model = small[0]
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    epi = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3], 1) if use_epi else None
    bias = torch.randn(Cout, generator=g) if use_bias else None
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), w, bias, 1, 1)
    if epi is not None:
        want = want * epi[:, 0].view(1, -1, 1, 1) + epi[:, 1].view(1, -1, 1, 1)
        want = torch.where(want > 0, want, want * slope)
    nat = vr.native
    xn, wn = x.numpy(), w.numpy()
    en = epi.numpy().copy() if epi is not None else None
    bn = bias.numpy() if bias is not None else None
    got = {}
    try:
        for mode in (2, 3, 0):
            model.set_option('mfma_mode', mode)
            out = np.empty(tuple(want.shape), np.float32)
            nat.check(nat.lib().vr_debug_conv2d(
                model._handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, 3, 1, 1, 1, 1 | 2 | (4 if use_epi else 0),
                nat.np_ptr(en) if en is not None else None, ctypes.c_float(slope if use_epi else 1.0),
                nat.np_ptr(bn) if bn is not None else None, nat.np_ptr(out), None))
            got[mode] = out
    finally:
        model.set_option('mfma_mode', -1)
    scale = float(want.abs().max())
    e2, e0 = float(np.abs(got[2] - want.numpy()).max()) / scale, float(np.abs(got[0] - want.numpy()).max()) / scale
    e3 = float(np.abs(got[3] - want.numpy()).max()) / scale
    print('fused upsample: split-bf16 direct %.2e, split-fp16 direct %.2e, fp32 fused loader %.2e of the output scale' % (e2, e3, e0))
    assert e2 < 1e-4 and e0 < 1e-4 and e3 < 1e-4
    assert not np.array_equal(got[0], got[2]) and not np.array_equal(got[3], got[2])


@pytest.mark.gpu
def test_conv_x3_fused_upsample_never_reads_past_the_staging_tile(vr, small):
    """the fused bilinear x2 read the +1 row / +1 column neighbours unconditionally (weight 0 at the image edge) -- up to
    (LW + 1) * 32 + 16 bytes past the low-resolution staging tile, i.e. into the epilogue constants that follow it in LDS.  With a
    non-finite constant there, 0 * inf = nan poisoned every output of the tile's last rows / columns.  The steps to the neighbours are
    clamped now: an infinite bias on ONE cout must make that cout infinite and leave every other cout exactly as without it."""
    model = small[0]
    nat = vr.native
    N, Cin, H, W, Cout = 1, 16, 9, 16, 32          # 18 x 32 outputs: the bottom tile rows and the right column interpolate at the image edge
    g = torch.Generator().manual_seed(5)
    xn = torch.randn(N, Cin, H, W, generator=g).numpy()
    wn = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).numpy()
    en = np.stack([np.full(Cout, 1e30, np.float32), np.full(Cout, 3e38, np.float32)], 1).copy()     # huge scale / shift in the LDS constants
    en[:-1] = np.stack([np.ones(Cout - 1, np.float32), np.zeros(Cout - 1, np.float32)], 1)
    outs = []
    try:
        model.set_option('mfma_mode', 3)
        for poison in (False, True):
            bn = np.zeros(Cout, np.float32)
            if poison:
                bn[-1] = np.inf
            out = np.empty((N, Cout, 2 * H, 2 * W), np.float32)
            nat.check(nat.lib().vr_debug_conv2d(model._handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, 3, 1, 1, 1, 1 | 2 | 4,
                                                nat.np_ptr(en), ctypes.c_float(1.0), nat.np_ptr(bn), nat.np_ptr(out), None))
            outs.append(out)
    finally:
        model.set_option('mfma_mode', -1)
    assert np.isfinite(outs[0][:, :-1]).all() and np.isfinite(outs[1][:, :-1]).all()
    assert np.array_equal(outs[0][:, :-1], outs[1][:, :-1])
    assert np.isinf(outs[1][:, -1]).all()


SPLIT_CASES = [(3, 64, 128, 256, 64), (3, 61, 128, 256, 64), (3, 128, 64, 256, 128)]      # big enough for the 64-cout Winograd variant


@pytest.mark.parametrize('case', SPLIT_CASES, ids=str)
def test_conv_winograd_split_bf16_mode_is_fp32_exact(vr, small, case):
    """mfma_mode 2 (fp32 products as six bf16 products of three-way split operands, fp32 accumulation; the DIRECT kernel of
    conv_x3.hip) against an fp64 reference.  "As exact as fp32" is stated against fp32 DIRECT convolutions of the same layer:
    the reference's own arithmetic (torch's fp32 conv on the CPU) and this library's fp32-MFMA direct kernel (conv_dma.hip,
    mode 0 with the plain weights) -- bar: <= 1.5x the larger of the two + 1e-7 of the output scale, in max and in rms.  (The
    Winograd kernel of mode 0 sums 2.25x fewer products and sits ~1.5-2x BELOW every direct form; it is printed, not the bar.)
    Three orders below the bf16-operand mode; and mode 2 must differ from mode 0 in the last bits (= the split kernel ran)."""
    N, Cin, H, W, Cout = case
    model = small[0]
    nat = vr.native
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g) * torch.exp(torch.randn(N, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    want = F.conv2d(x.double(), w.double(), None, 1, 1).numpy()
    cpu32 = F.conv2d(x, w, None, 1, 1).numpy()
    scale = float(np.abs(want).max())
    got = {}
    try:
        for key, mode, flags in (('wino0', 0, 2), ('direct0', 0, 0), ('split', 2, 2), ('half', 3, 2)):
            model.set_option('mfma_mode', mode)
            out = np.empty(want.shape, np.float32)
            nat.check(nat.lib().vr_debug_conv2d(model._handle.h, nat.np_ptr(x.numpy()), N, Cin, H, W, nat.np_ptr(w.numpy()), Cout, 3, 1, 1, 1,
                                                flags, None, ctypes.c_float(1.0), None, nat.np_ptr(out), None))
            got[key] = out
    finally:
        model.set_option('mfma_mode', -1)

    def err(a):
        e = np.abs(a - want)
        return e.max() / scale, np.sqrt((e ** 2).mean()) / scale
    e_split, e_wino, e_direct, e_cpu = err(got['split']), err(got['wino0']), err(got['direct0']), err(cpu32)
    print('max / rms error of the output scale -- split-bf16: %.2e / %.2e   fp32-MFMA direct: %.2e / %.2e   torch CPU fp32: %.2e / %.2e   '
          'fp32-MFMA Winograd: %.2e / %.2e' % (e_split + e_direct + e_cpu + e_wino))
    assert not np.array_equal(got['wino0'], got['split']) and not np.array_equal(got['direct0'], got['split']), 'mode 2 fell back to an fp32 kernel'
    assert e_split[0] <= 1.5 * max(e_direct[0], e_cpu[0]) + 1e-7
    assert e_split[1] <= 1.5 * max(e_direct[1], e_cpu[1]) + 1e-7
    assert e_split[0] < 5e-6
    # mfma_mode 3 (conv_x3h.hip: two fp16 planes per operand, three products): 22 significand bits per operand and one omitted
    # 2^-22 term -- measured 1.3-2x an fp32 direct convolution's rounding error; bar 2.5x + 2e-7, and under 5e-6 absolute
    e_half = err(got['half'])
    print('split-fp16 (three products): %.2e / %.2e' % e_half)
    assert not np.array_equal(got['half'], got['split']) and not np.array_equal(got['half'], got['direct0']), 'mode 3 fell back'
    assert e_half[0] <= 2.5 * max(e_direct[0], e_cpu[0]) + 2e-7 and e_half[1] <= 2.5 * max(e_direct[1], e_cpu[1]) + 2e-7
    assert e_half[0] < 5e-6


def test_forward_taps_small_net(vr, small):
    """Every recorded intermediate of the small net vs the oracle's (localises a broken layer)."""
    model, sd, n_fft = small
    nat = vr.native
    x = torch.rand(2, 2, n_fft // 2 + 1, 160, generator=torch.Generator().manual_seed(0))
    cascaded_net.TAPS = {}
    with torch.no_grad():
        want = cascaded_net.forward(x, sd, n_fft=n_fft)
    taps = cascaded_net.TAPS
    cascaded_net.TAPS = None
    nat.check(nat.lib().vr_debug_record_taps(model._handle.h, 1))
    got = model.forward(x.to('cuda:0')).cpu()
    report, bad = [], []
    for name, ref in taps.items():
        shape = (ctypes.c_int64 * 4)()
        n = nat.lib().vr_debug_get_tap(model._handle.h, name.encode(), None, 0, shape)
        assert n > 0, name
        buf = np.empty(tuple(int(s) for s in shape), np.float32)
        nat.lib().vr_debug_get_tap(model._handle.h, name.encode(), nat.np_ptr(buf), buf.size, shape)
        if tuple(buf.shape) != tuple(ref.shape):
            bad.append('%s shape %s vs %s' % (name, buf.shape, tuple(ref.shape)))
            continue
        err = float(np.abs(buf - ref.numpy()).max())
        scale = float(ref.abs().max()) + 1e-6
        report.append('%-40s err %.3e scale %.3e' % (name, err, scale))
        if err > 2e-4 * scale + 1e-5:
            bad.append(report[-1])
    nat.check(nat.lib().vr_debug_record_taps(model._handle.h, 0))
    print('\n'.join(report))
    assert not bad, '\n'.join(bad)
    assert float((got - want).abs().max()) < 1e-4


def test_predict_variants_small_net(small):
    model, sd, n_fft = small
    x = torch.rand(3, 2, n_fft // 2 + 1, 144, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want_m = cascaded_net.predict_mask(x, sd, n_fft=n_fft)
        want_p = cascaded_net.predict(x, sd, n_fft=n_fft)
        want_f = cascaded_net.forward(x, sd, n_fft=n_fft)
    for dev in ('cuda:0', 'cpu'):
        xin = x.to(dev)
        got_m, got_p, got_f = model.predict_mask(xin), model.predict(xin), model(xin)
        assert got_m.device.type == torch.device(dev).type
        assert got_m.shape == want_m.shape == (3, 2, n_fft // 2 + 1, 16)
        assert float((got_m.cpu() - want_m).abs().max()) < 1e-4
        assert float((got_p.cpu() - want_p).abs().max()) < 1e-4
        assert float((got_f.cpu() - want_f).abs().max()) < 1e-4
    # replicate-pad row (lib/nets.py:111-115)
    assert torch.equal(got_f[:, :, -1], got_f[:, :, -2])


@pytest.mark.parametrize('B,T', [(1, 16), (3, 32), (2, 80), (1, 272), (5, 160)])
def test_forward_shape_sweep_small_net(small, B, T):
    """Every valid frame count (multiples of 16) and batch size: at the small ones the deep levels are 1-8
    columns wide, so the dispatcher leaves the LDS-DMA / Winograd kernels (16-byte pieces need W % 4 == 0) for
    the fused-loader ones and the odd-width upsample -- same results either way."""
    model, sd, n_fft = small
    x = torch.rand(B, 2, n_fft // 2 + 1, T, generator=torch.Generator().manual_seed(100 + T))
    with torch.no_grad():
        want = cascaded_net.forward(x, sd, n_fft=n_fft)
    got = model(x.to('cuda:0')).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-4


def test_reference_error_behaviour(small):
    model, sd, n_fft = small
    with pytest.raises(ValueError):        # crop_center ValueError (frames not a multiple of 16)
        model.predict_mask(torch.rand(1, 2, n_fft // 2 + 1, 152))
    with pytest.raises(AssertionError):    # assert mask.size()[3] > 0
        model.predict_mask(torch.rand(1, 2, n_fft // 2 + 1, 128))
    with pytest.raises(ValueError):
        model.predict_mask(torch.rand(1, 2, 100, 160))


def test_state_dict_roundtrip(small):
    model, sd, _ = small
    model._host_stale = True
    back = model.state_dict()
    assert list(back.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(back[k], sd[k]), k


@pytest.fixture(scope='module')
def full(vr):
    sd = weights.make_state_dict(1234)
    model = vr.nets.CascadedNet(2048, 1024, 32, 128)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0'))
    model.eval()
    return model, sd


def test_predict_mask_full_net(full):
    """The flagship configuration: CascadedNet(2048,1024,32,128), B=2 crops of 256 frames."""
    model, sd = full
    x = torch.rand(2, 2, 1025, 256, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = cascaded_net.predict_mask(x, sd)
    got = model.predict_mask(x.to('cuda:0')).cpu()
    diff = (got - want).abs()
    print('full net: max-abs %.3e mean-abs %.3e mask std %.3f' % (float(diff.max()), float(diff.mean()), float(want.std())))
    assert got.shape == (2, 2, 1025, 128)
    assert float(diff.max()) < 1e-4 and float(diff.mean()) < 1e-5
    assert float(want.std()) > 0.02


@pytest.mark.parametrize('split_mode', [2, 3], ids=['six_bf16_products', 'three_fp16_products'])
def test_predict_mask_full_net_split_bf16_mode(full, split_mode):
    """The same configuration with mfma_mode 2 / 3: the SAME bars against the oracle, and agreement with mode 0 at rounding level."""
    model, sd = full
    x = torch.rand(2, 2, 1025, 256, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = cascaded_net.predict_mask(x, sd)
    model.set_option('mfma_mode', 0)
    ref = model.predict_mask(x.to('cuda:0')).cpu()
    try:
        model.set_option('mfma_mode', split_mode)
        got = model.predict_mask(x.to('cuda:0')).cpu()
    finally:
        model.set_option('mfma_mode', -1)
    diff = (got - want).abs()
    print('full net, split mode: max-abs %.3e mean-abs %.3e; vs mode 0: %.3e' % (float(diff.max()), float(diff.mean()), float((got - ref).abs().max())))
    assert float(diff.max()) < 1e-4 and float(diff.mean()) < 1e-5
    assert not torch.equal(got, ref) and float((got - ref).abs().max()) < 2e-5


def test_batch_independence_full_net(full):
    """Eval-mode crops are independent (inference.py:44-48): batching must not change results."""
    model, _ = full
    x = torch.rand(3, 2, 1025, 256, generator=torch.Generator().manual_seed(3)).to('cuda:0')
    all3 = model.predict_mask(x)
    one = model.predict_mask(x[1:2].contiguous())
    # not bit-identical: the launcher picks tile shapes / channel-chunk sizes from the grid size, which
    # changes the fp32 summation order; anything beyond rounding noise would be cross-crop leakage
    assert float((all3[1:2] - one).abs().max()) < 1e-5


def test_stft_istft_vs_oracle(vr):
    wave = separator.synth_wave(3.0, seed=4)
    want = stft_np.wave_to_spectrogram(wave, 1024, 2048)
    got = vr.spec_utils.wave_to_spectrogram(wave, 1024, 2048)
    assert got.shape == want.shape and got.dtype == np.complex64
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    back_want = stft_np.spectrogram_to_wave(want, 1024)
    back = vr.spec_utils.spectrogram_to_wave(want, hop_length=1024)
    assert back.shape == back_want.shape and back.dtype == np.float32
    assert np.abs(back - back_want).max() < 2e-5
    # round trip property at any size: istft(stft(x)) == x away from the trimmed tail
    assert np.abs(back - wave[:, :back.shape[1]]).max() < 2e-5
    mono = vr.spec_utils.spectrogram_to_wave(want[0], hop_length=1024)
    assert np.abs(mono - back_want[0]).max() < 2e-5


def test_stft_ragged_lengths(vr):
    for L in (1024, 1500, 4096 + 17):
        wave = separator.synth_wave(L / 44100.0, seed=L)[:, :L]
        want = stft_np.wave_to_spectrogram(wave, 1024, 2048)
        got = vr.spec_utils.wave_to_spectrogram(wave, 1024, 2048)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 2e-5 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize('tta', [False, True])
def test_separator_small_net(vr, small, tta):
    model, sd, n_fft = small
    rng = np.random.default_rng(5)
    T = 300
    X = (rng.standard_normal((2, n_fft // 2 + 1, T)) + 1j * rng.standard_normal((2, n_fft // 2 + 1, T))).astype(np.complex64)
    want_y, want_v = separator.separate(X.copy(), sd, tta=tta, n_fft=n_fft, batchsize=2, cropsize=160)
    for bs in (2, 0):
        sp = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=bs, cropsize=160)
        got_y, got_v = (sp.separate_tta if tta else sp.separate)(X.copy())
        assert got_y.shape == want_y.shape and got_y.dtype == np.complex64
        scale = np.abs(X).max()
        assert np.abs(got_y - want_y).max() < 1e-4 * scale
        assert np.abs(got_v - want_v).max() < 1e-4 * scale
        # size-independent property: the two stems sum back to the mixture
        assert np.abs(got_y + got_v - X).max() < 1e-5 * scale


def test_separate_wave_pipeline(vr, small):
    """STFT -> separate -> iSTFT x2 in one device-resident call equals the staged calls."""
    model, sd, n_fft = small
    hop = n_fft // 2
    wave = separator.synth_wave(1.0, seed=6)[:, :hop * 200 + 13]
    sp = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=0, cropsize=160)
    spec = vr.spec_utils.wave_to_spectrogram(wave, hop, n_fft)
    y_spec, v_spec = sp.separate(spec)
    y_want = vr.spec_utils.spectrogram_to_wave(y_spec, hop_length=hop)
    v_want = vr.spec_utils.spectrogram_to_wave(v_spec, hop_length=hop)
    y, v = sp.separate_wave(wave)
    assert y.shape == y_want.shape
    assert np.abs(y - y_want).max() < 1e-5 and np.abs(v - v_want).max() < 1e-5
    yt, vt = sp.separate_wave(torch.from_numpy(wave).to('cuda:0'))
    assert np.abs(yt.cpu().numpy() - y).max() < 1e-6
    # oracle for the whole chain
    spec_o = stft_np.wave_to_spectrogram(wave, hop, n_fft)
    yo, vo = separator.separate(spec_o, sd, n_fft=n_fft, batchsize=4, cropsize=160)
    assert np.abs(y - stft_np.spectrogram_to_wave(yo.astype(np.complex64), hop)).max() < 1e-4


@pytest.mark.parametrize('tta', [False, True])
def test_separator_postprocess_merge_artifacts(vr, small, tta):
    """Separator(postprocess=True): inference.py:27-30 + spec_utils.merge_artifacts, on device."""
    model, sd, n_fft = small
    sd2 = weights.clone_state_dict(sd)
    sd2['out.weight'] = sd['out.weight'] * 0.05          # keeps every frame's mask minimum above the 0.05 threshold
    model.load_state_dict(sd2)
    rng = np.random.default_rng(8)
    T = 300
    X = (rng.standard_normal((2, n_fft // 2 + 1, T)) + 1j * rng.standard_normal((2, n_fft // 2 + 1, T))).astype(np.complex64)
    want_y, want_v = separator.separate(X.copy(), sd2, tta=tta, post=True, n_fft=n_fft, batchsize=2, cropsize=160)
    plain_y, _ = separator.separate(X.copy(), sd2, tta=tta, post=False, n_fft=n_fft, batchsize=2, cropsize=160)
    assert np.abs(want_y - plain_y).max() > 0.05 * np.abs(X).max()          # the flag really changes the result
    sp = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=0, cropsize=160, postprocess=True)
    got_y, got_v = (sp.separate_tta if tta else sp.separate)(X.copy())
    scale = np.abs(X).max()
    assert np.abs(got_y - want_y).max() < 1e-4 * scale
    assert np.abs(got_v - want_v).max() < 1e-4 * scale
    model.load_state_dict(sd)
