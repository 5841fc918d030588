"""conv_x3d.hip (round 6): the 16-column layers -- the dilated 3x3 branches of layers.ASPPModule (lib/layers.py:77-85), its 1x1 branch
(layers.py:74-76) and Encoder.conv2 of enc5 (layers.py:34) -- on the fp16 matrix pipe with conv_x3h.hip's three-product arithmetic, and the
one-launch form of the four ASPP branch convs.  Against torch (fp32 and fp64), against the library's own fp32-MFMA kernel for the same
layer, forward / data gradient / BatchNorm partial sums, and through the whole network with the launch on and off."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cascaded_net, train_step, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def net(vr):
    sd = weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32)
    m = vr.nets.CascadedNet(512, 256, 8, 32)
    m.load_state_dict(sd)
    m.to(torch.device('cuda:0'))
    return m, sd


def _debug_conv(vr, model, x, w, ks, dh, dw, epi, slope, bias, flags, stats=False):
    nat = vr.native
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = np.empty((N, Cout, H, W), np.float32)
    st = np.empty((Cout, 2), np.float32) if stats else None
    xn, wn = np.ascontiguousarray(x.numpy()), np.ascontiguousarray(w.numpy())
    en = np.ascontiguousarray(epi.numpy()) if epi is not None else None
    bn = np.ascontiguousarray(bias.numpy()) if bias is not None else None
    nat.check(nat.lib().vr_debug_conv2d(
        model._handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, ks, 1, dh, dw, flags,
        nat.np_ptr(en) if en is not None else None, ctypes.c_float(slope if epi is not None else 1.0),
        nat.np_ptr(bn) if bn is not None else None, nat.np_ptr(out), nat.np_ptr(st) if stats else None))
    return out, st


X3D_CASES = [
    # N, Cin, H, Cout, ks, dh, dw, epi, slope, bias          (W = 16 always)
    (2, 32, 32, 32, 3, 4, 2, 1, 0.0, 0),
    (1, 64, 64, 64, 3, 8, 4, 1, 0.0, 0),
    (2, 16, 32, 16, 3, 12, 6, 0, 1.0, 0),
    (3, 61, 32, 48, 3, 12, 6, 1, 0.0, 1),          # odd Cin (partial chunk), Cout not a multiple of 32, bias
    (2, 128, 64, 128, 3, 4, 2, 1, 0.0, 0),         # 64-cout tiles
    (2, 128, 32, 128, 3, 8, 4, 1, 0.01, 0),
    (11, 64, 32, 64, 3, 12, 6, 1, 0.0, 0),         # the S30 batch of the 64-channel ASPP (stage 1, high band)
    (1, 24, 20, 32, 3, 12, 6, 1, 0.0, 0),          # H not a multiple of the 16-row tile
    (2, 9, 5, 8, 3, 4, 2, 0, 1.0, 1),              # H smaller than the dilation reach: every tap row but the centre is padding
    (2, 32, 32, 32, 3, 1, 1, 1, 0.01, 0),          # enc5.conv2: dilation 1, LeakyReLU
    (1, 100, 64, 64, 3, 1, 1, 1, 0.01, 0),
    (2, 40, 32, 8, 1, 1, 1, 1, 0.0, 0),            # 1x1 (ASPP conv2)
    (1, 256, 64, 256, 1, 1, 1, 1, 0.0, 1),
    (3, 17, 16, 40, 1, 1, 1, 0, 1.0, 0),
]


@pytest.mark.parametrize('case', X3D_CASES, ids=str)
def test_conv_x3d_vs_torch_and_vs_the_fp32_pipe(vr, net, case):
    """Bar as for conv_x3h (tests/test_gpu_b16.py): 1e-4 of the output scale against torch fp32, and against an fp64 reference no worse
    than 2.5 x the library's fp32-MFMA direct kernel for the same layer + 2e-7 of the scale.  The two kernels must differ in the last
    bits (= conv_x3d ran)."""
    N, Cin, H, Cout, ks, dh, dw, use_epi, slope, use_bias = case
    model = net[0]
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(N, Cin, H, 16, generator=g) * torch.exp(0.5 * torch.randn(N, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    epi = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3], 1) if use_epi else None
    bias = torch.randn(Cout, generator=g) if use_bias else None
    pad = (dh, dw) if ks == 3 else (0, 0)
    want64 = F.conv2d(x.double(), w.double(), bias.double() if bias is not None else None, 1, pad, (dh, dw))
    if epi is not None:
        want64 = want64 * epi[:, 0].double().view(1, -1, 1, 1) + epi[:, 1].double().view(1, -1, 1, 1)
        want64 = torch.where(want64 > 0, want64, want64 * slope)
    scale = float(want64.abs().max())
    flags = 4 if use_epi else 0
    try:
        model.set_option('mfma_mode', 3)
        got, _ = _debug_conv(vr, model, x, w, ks, dh, dw, epi, slope, bias, flags | 2)
        model.set_option('conv_x3d', 0)
        ref, _ = _debug_conv(vr, model, x, w, ks, dh, dw, epi, slope, bias, flags | 2)
    finally:
        model.set_option('conv_x3d', -1)
        model.set_option('mfma_mode', -1)
    e = float(np.abs(got - want64.numpy()).max()) / scale
    e0 = float(np.abs(ref - want64.numpy()).max()) / scale
    print('conv_x3d %.2e, fp32 MFMA kernel %.2e of the output scale' % (e, e0))
    assert e < 1e-4
    assert e <= 2.5 * e0 + 2e-7, (e, e0)
    assert not np.array_equal(got, ref)


def test_conv_x3d_batchnorm_partial_sums(vr, net):
    """Training forward: the per-tile (sum, sum of squares) rows the BatchNorm statistics are finalised from."""
    model = net[0]
    g = torch.Generator().manual_seed(3)
    for (Cin, H, Cout, ks, dh, dw) in ((64, 32, 64, 3, 8, 4), (33, 64, 48, 3, 12, 6), (32, 32, 32, 1, 1, 1), (40, 32, 96, 3, 1, 1)):
        x = torch.randn(4, Cin, H, 16, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
        want = F.conv2d(x.double(), w.double(), None, 1, (dh, dw) if ks == 3 else (0, 0), (dh, dw))
        try:
            model.set_option('mfma_mode', 3)
            got, st = _debug_conv(vr, model, x, w, ks, dh, dw, None, 1.0, None, 2, stats=True)
        finally:
            model.set_option('mfma_mode', -1)
        s1 = want.sum(dim=(0, 2, 3)).numpy()
        s2 = (want ** 2).sum(dim=(0, 2, 3)).numpy()
        assert float(np.abs(got - want.numpy()).max()) / float(want.abs().max()) < 1e-5
        assert float(np.abs(st[:, 0] - s1).max() / (np.abs(s1).max() + 1.0)) < 1e-5
        assert float(np.abs(st[:, 1] - s2).max() / (np.abs(s2).max() + 1.0)) < 1e-5


def test_conv_x3d_special_values(vr, net):
    """conv_x3h's scaling argument holds tile by tile here too: 2^+-100 inputs, fp32 subnormals, a scale that swings by 2^60 from one
    8-channel chunk to the next -- against fp64, relative to the output scale of each image."""
    model = net[0]
    g = torch.Generator().manual_seed(9)
    Cin, Cout, H = 48, 32, 32
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    base = torch.randn(4, Cin, H, 16, generator=g)
    x = base.clone()
    x[0] *= 2.0 ** 100
    x[1] *= 2.0 ** -100
    x[2] *= 2.0 ** -140                                  # subnormal inputs
    sw = torch.ones(Cin)
    sw[8:16] = 2.0 ** 30
    sw[16:24] = 2.0 ** -30
    x[3] *= sw.view(-1, 1, 1)
    try:
        model.set_option('mfma_mode', 3)
        got, _ = _debug_conv(vr, model, x, w, 3, 12, 6, None, 1.0, None, 2)
    finally:
        model.set_option('mfma_mode', -1)
    want = F.conv2d(x.double(), w.double(), None, 1, (12, 6), (12, 6)).numpy()
    for n in range(4):
        sc = np.abs(want[n]).max()
        err = np.abs(got[n] - want[n]).max() / sc
        print('image %d: scale %.3e, error %.2e of it' % (n, sc, err))
        assert err < (3e-3 if n == 2 else 2e-6)          # (image 2: the INPUT is subnormal, 2-3 significant bits)


def test_conv_x3d_data_gradient(vr, net):
    """The data gradient of a dilated conv is the same dilated conv over dz with flipped / transposed weights: vr_debug_conv2d_backward
    takes conv_x3d for it in mfma_mode 3 (and the fp32 kernels with conv_x3d 0); both against torch autograd in fp64."""
    model = net[0]
    nat = vr.native
    g = torch.Generator().manual_seed(21)
    for (Cin, H, Cout, ks, dh, dw) in ((64, 32, 64, 3, 4, 2), (48, 64, 96, 3, 12, 6), (32, 32, 64, 1, 1, 1), (64, 32, 64, 3, 1, 1)):
        x = torch.randn(2, Cin, H, 16, generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(Cout, Cin, ks, ks, generator=g, dtype=torch.float64) / (Cin * ks * ks) ** 0.5).requires_grad_()
        dz = torch.randn(2, Cout, H, 16, generator=g, dtype=torch.float64)
        F.conv2d(x, w, None, 1, (dh, dw) if ks == 3 else (0, 0), (dh, dw)).backward(dz)
        xn, wn, dzn = x.detach().float().numpy(), w.detach().float().numpy(), dz.float().numpy()
        outs = {}
        try:
            model.set_option('mfma_mode', 3)
            for x3d in (2, 0):
                model.set_option('conv_x3d', x3d)
                dxo, dwo = np.empty_like(xn), np.empty_like(wn)
                nat.check(nat.lib().vr_debug_conv2d_backward(
                    model._handle.h, nat.np_ptr(xn), 2, Cin, H, 16, nat.np_ptr(wn), Cout, ks, 1, dh, dw, 0, None, ctypes.c_float(1.0),
                    nat.np_ptr(dzn), nat.np_ptr(dxo), nat.np_ptr(dwo)))
                outs[x3d] = (dxo, dwo)
        finally:
            model.set_option('conv_x3d', -1)
            model.set_option('mfma_mode', -1)
        gx = x.grad.numpy()
        sc = np.abs(gx).max()
        e2, e0 = np.abs(outs[2][0] - gx).max() / sc, np.abs(outs[0][0] - gx).max() / sc
        print('dx: conv_x3d %.2e, fp32 kernel %.2e' % (e2, e0))
        assert e2 < 1e-5 and e2 <= 2.5 * e0 + 2e-7
        assert not np.array_equal(outs[2][0], outs[0][0])
        gw = w.grad.numpy()
        assert np.abs(outs[2][1] - gw).max() / np.abs(gw).max() < 1e-4


def _kernels_of(vr, model, fn):
    nat, h = vr.native, model._handle.h
    nat.check(nat.lib().vr_profile_begin(h))
    try:
        fn()
    finally:
        a, b, c, d = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        nat.check(nat.lib().vr_profile_end(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
    need = nat.lib().vr_profile_report(h, None, 0)
    buf = ctypes.create_string_buffer(int(need) + 1)
    nat.lib().vr_profile_report(h, buf, need)
    return [ln.split('\t')[0] for ln in buf.value.decode().splitlines()]


def test_aspp_branches_in_one_launch_through_the_network(vr, net):
    """predict_mask at the reference's crop size (256 frames -> 16 columns at 1/16 resolution) with the ASPP branch group on (one
    conv_x3d_aspp launch per module), as single conv_x3d launches, and off (round 5's fp32 kernels): each against the CPU oracle at the
    north star's 1e-4, the three within 2e-5 of one another, and the launch profile shows which kernels ran."""
    model, sd = net
    x = torch.rand(3, 2, 257, 256, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = cascaded_net.predict_mask(x, sd, n_fft=512).numpy()
    model.eval()
    xd = x.to('cuda:0')
    got, names = {}, {}
    try:
        model.set_option('mfma_mode', 3)                  # (conv_x3d exists in the fp16-split mode only: also when VR_MFMA_MODE sets another default)
        for mode in (2, 1, 0):
            model.set_option('conv_x3d', mode)
            got[mode] = model.predict_mask(xd).cpu().numpy()
            names[mode] = _kernels_of(vr, model, lambda: model.predict_mask(xd))
    finally:
        model.set_option('conv_x3d', -1)
        model.set_option('mfma_mode', -1)
    for mode in (2, 1, 0):
        err = float(np.abs(got[mode] - want).max())
        print('conv_x3d %d: max-abs error vs the oracle %.2e' % (mode, err))
        assert err < 1e-4
    assert float(np.abs(got[2] - got[0]).max()) < 2e-5 and float(np.abs(got[1] - got[0]).max()) < 2e-5
    assert np.array_equal(got[2], got[1])                 # the same tile arithmetic, grouped or not
    assert not np.array_equal(got[2], got[0])
    n2 = [n for n in names[2] if 'conv_x3d' in n]
    assert sum('conv_x3d_aspp_kernel' in n for n in n2) >= 1 and any('conv_x3d_kernel<9, 1, 1' in n for n in n2), names[2]
    assert not any('conv_x3d_aspp_kernel' in n for n in names[1]) and any('conv_x3d_kernel<9, 12, 6' in n for n in names[1]), names[1]
    assert not any('conv_x3d' in n for n in names[0]), names[0]
    # bit-reproducible from run to run
    try:
        model.set_option('mfma_mode', 3)
        again = model.predict_mask(xd).cpu().numpy()
    finally:
        model.set_option('mfma_mode', -1)
    assert np.array_equal(again, got[2])


def test_train_step_with_conv_x3d_matches_the_fp32_pipe(vr, net):
    """A train step at 256 frames: forward (BatchNorm statistics from conv_x3d's partial sums) and data gradients of the ASPP branches and
    enc5.conv2 on conv_x3d (A) against the same step with conv_x3d off (B).  The loss must agree to 1e-6 relative.  The gradients of two
    fp32-grade evaluations differ by rounding noise carried back through five cascaded nets (ReLU masks, BatchNorm): the yardstick is a
    CONTROL pair of fp32-grade evaluations that never touch conv_x3d -- B against C = conv_x3d off and `train_winograd` 0 (the fused-loader
    kernels of rounds 1-2 everywhere) -- and A must sit as close to B as C does (3 x per tensor class, measured over all tensors)."""
    model, sd = net
    X, y = train_step.synth_batch(2, T=256, n_fft=512, seed=8)
    res = {}
    try:
        for name, x3d, wino in (('A', 2, 1), ('B', 0, 1), ('C', 0, 0)):
            model.load_state_dict(sd)
            model.to(torch.device('cuda:0'))
            model.train()
            model.set_option('mfma_mode', 3)
            model.set_option('conv_x3d', x3d)
            model.set_option('train_winograd', wino)
            model.set_dropout_masks(None)                # (the library's generator is keyed on the number of forwards so far: off for an A / B)
            model.zero_grad()
            loss = model.train_step(X.to('cuda:0'), y.to('cuda:0'), 1)
            res[name] = (float(loss), {k: v.numpy().astype(np.float64) for k, v in model.grads().items()})
    finally:
        model.set_option('conv_x3d', -1)
        model.set_option('mfma_mode', -1)
        model.set_option('train_winograd', 1)
        model.set_dropout_masks(0)
        model.load_state_dict(sd)
        model.to(torch.device('cuda:0'))
        model.eval()
    la, lb, lc = res['A'][0], res['B'][0], res['C'][0]
    print('loss: conv_x3d %.8f, off %.8f, control %.8f' % (la, lb, lc))
    assert abs(la - lb) <= 1e-6 * abs(lb) + 1e-7 and abs(lc - lb) <= 1e-5 * abs(lb) + 1e-7
    gmax = max(float(np.abs(g).max()) for g in res['B'][1].values())

    def spread(p, q):
        """(largest max-abs difference / tensor scale, smallest cosine, rms over tensors of the relative difference)"""
        worst, wcos, acc, n = 0.0, 1.0, 0.0, 0
        for k, gq in res[q][1].items():
            gp = res[p][1][k]
            sc = float(np.abs(gq).max())
            # (a Linear bias in front of BatchNorm1d has an exactly zero gradient: what the kernels leave there is rounding noise)
            if sc < 1e-5 * gmax:
                continue
            d = float(np.abs(gp - gq).max()) / sc
            worst = max(worst, d)
            acc += d * d
            n += 1
            if gq.size >= 16:
                wcos = min(wcos, float((gp * gq).sum() / (np.linalg.norm(gp) * np.linalg.norm(gq) + 1e-300)))
        return worst, wcos, (acc / max(n, 1)) ** 0.5

    wa, ca, ra = spread('A', 'B')
    wc, cc, rc = spread('C', 'B')
    print('conv_x3d vs off:        largest difference %.2e of a tensor scale, smallest cosine %.6f, rms %.2e' % (wa, ca, ra))
    print('control (other fp32 kernels) vs off: largest %.2e, smallest cosine %.6f, rms %.2e' % (wc, cc, rc))
    assert wa <= 3.0 * wc + 1e-4 and ra <= 3.0 * rc + 1e-5 and (1.0 - ca) <= 3.0 * (1.0 - cc) + 1e-6
