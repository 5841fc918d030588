"""conv_x3p.hip (eval path of mfma_mode 2): the 3x3 stride-1 convolution over activations stored as three bf16 planes, in isolation
through vr_debug_kernel('conv_planes') -- to_planes_kernel / upsample2x_planes_kernel -> x3p_weights_kernel (padded channel order)
-> conv_x3p_kernel (LDS-DMA loader) -> fp32 AND plane outputs -- against torch's F.conv2d in fp64 (lib/layers.py:12-20,51-56).

Bars: the split is exact (x = p1 + p2 + p3) and six bf16 products reproduce the fp32 product, so the error against fp64 must be an
fp32 direct convolution's: 2e-6 of the output scale absolute, and <= 3 x torch's own CPU fp32 conv error + 2e-7 (oneDNN sums in
blocks, the MFMA chain sums the Cin x 9 products in one fp32 sequence: measured 7.7e-7 vs 3.1e-7 at Cin = 64, the same ratio the
fp32-MFMA direct kernel of this library has, tests/test_gpu_b16.py).  The plane
output summed back must equal the fp32 output BIT FOR BIT (same epilogue values, exact split).  Special values: subnormals, 2^+-100
scales, bf16 rounding boundaries."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def handle(vr):
    model = vr.nets.CascadedNet(512, 256, 8, 32)
    model.to(torch.device('cuda:0'))
    return vr.native, model._handle


def f32(t):
    return None if t is None else np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def run(handle, xs, w, epi, bias, slope, up0, th, H, W):
    nat, h = handle
    N, Cout = xs[0].shape[0], w.shape[0]
    Cs = [x.shape[1] for x in xs] + [0] * (3 - len(xs))
    out = [np.empty((N, Cout, H, W), np.float32), np.empty((N, Cout, H, W), np.float32)]
    ins = [f32(x) for x in xs] + [None] * (3 - len(xs)) + [f32(w), f32(epi), f32(bias)]
    nat.debug_kernel(h, 'conv_planes', [N, H, W, Cout] + Cs + [int(up0), th], [slope, 1.0 if epi is not None else 0.0], ins, out)
    return out


def reference(xs, w, epi, bias, slope, up0, dtype):
    srcs = [x.to(dtype) for x in xs]
    if up0:
        srcs[0] = F.interpolate(srcs[0], scale_factor=2, mode='bilinear', align_corners=True)
    y = F.conv2d(torch.cat(srcs, 1), w.to(dtype), None if bias is None else bias.to(dtype), 1, 1)
    if epi is not None:
        y = y * epi[:, 0].to(dtype).view(1, -1, 1, 1) + epi[:, 1].to(dtype).view(1, -1, 1, 1)
        y = torch.where(y > 0, y, y * slope)
    return y


# N, (C0, C1, C2), Cout, H, W, up0, th, epilogue, bias
CASES = [
    (2, (64, 0, 0), 64, 32, 64, 0, 0, True, False),        # one source, 64-cout tile
    (2, (64, 0, 0), 64, 32, 64, 0, 8, True, False),        # the same on 8-row tiles
    (1, (16, 8, 0), 32, 48, 96, 0, 16, True, False),       # two sources, 32-cout tile, 16-row tiles
    (2, (2, 8, 16), 32, 40, 72, 0, 0, True, False),        # stage-3 enc1: sources of 2 / 8 / 16 channels (padding group), ragged tile edges
    (1, (16, 1, 8), 8, 24, 100, 1, 8, True, False),        # dec1 of the small band net: upsampled source, 1-channel LSTM branch, 8 couts, W % 32 != 0
    (1, (64, 32, 0), 48, 16, 32, 1, 0, True, True),        # decoder with upsample + bias, Cout not a multiple of 32, one tile column
    (1, (26, 0, 0), 20, 19, 33, 0, 16, False, False),      # no epilogue, odd H / W, Cout % 8 != 0
    (3, (128, 0, 0), 128, 16, 32, 0, 0, True, False),      # two cout tiles
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_planes_vs_torch(handle, case):
    N, Cs, Cout, H, W, up0, th, use_epi, use_bias = case
    g = torch.Generator().manual_seed(sum(Cs) * 7 + Cout + H + W)
    xs = []
    for i, C in enumerate(Cs):
        if C:
            h, w_ = (H // 2, W // 2) if (i == 0 and up0) else (H, W)
            xs.append(torch.randn(N, C, h, w_, generator=g))
    Cin = sum(Cs)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    epi = torch.stack([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3], 1) if use_epi else None
    bias = torch.randn(Cout, generator=g) * 0.2 if use_bias else None
    slope = 0.01
    got, back = run(handle, xs, w, epi, bias, slope, up0, th, H, W)
    want = reference(xs, w, epi, bias, slope, up0, torch.float64)
    cpu32 = reference(xs, w, epi, bias, slope, up0, torch.float32)
    scale = float(want.abs().max())
    e_gpu = float((torch.from_numpy(got).double() - want).abs().max()) / scale
    e_cpu = float((cpu32.double() - want).abs().max()) / scale
    print('conv_planes %s: max error / scale %.2e (torch CPU fp32: %.2e)' % (case, e_gpu, e_cpu))
    assert e_gpu < 2e-6 and e_gpu <= 3 * e_cpu + 2e-7
    assert np.array_equal(back, got), 'plane output summed back differs from the fp32 output'


def test_conv_planes_is_fp32_exact_on_special_values(handle):
    """Subnormals, 2^+-100 scales, values on / next to bf16 rounding boundaries: the three-plane split and the six products must
    still reproduce what an fp32 direct convolution gives (against fp64)."""
    g = torch.Generator().manual_seed(5)
    N, C, Cout, H, W = 1, 32, 32, 16, 64
    base = torch.randn(N, C, H, W, generator=g)
    bits = base.view(torch.int32)
    x = base.clone()
    x[:, 0::4] = (bits[:, 0::4] & ~0xFFFF).view(torch.float32)                    # exactly representable in bf16
    x[:, 1::4] = ((bits[:, 1::4] & ~0xFFFF) | 0x8000).view(torch.float32)         # half way between two bf16 numbers
    x[:, 2::4] = ((bits[:, 2::4] & ~0xFFFF) | 0x7FFF).view(torch.float32)         # one ulp below the half-way point
    # (bf16 shares fp32's exponent range, so the split stays exact down to |x| ~ 2^-110, where the third plane reaches bf16's
    # subnormals; below that an input keeps fewer bits than fp32 would -- subnormal inputs beside normal ones are the last case)
    for scale_x, scale_w in ((1.0, 1.0), (2.0 ** 100, 2.0 ** -100), (2.0 ** -100, 2.0 ** 60), (None, 1.0)):
        if scale_x is None:
            sel = torch.rand(x.shape, generator=g) < 0.3
            xs = [torch.where(sel, torch.randn(x.shape, generator=g) * 2.0 ** -140, x)]
        else:
            xs = [x * scale_x]
        w = torch.randn(Cout, C, 3, 3, generator=g) / (C * 9) ** 0.5 * scale_w
        got, back = run(handle, xs, w, None, None, 1.0, 0, 8, H, W)
        want = reference(xs, w, None, None, 1.0, 0, torch.float64)
        cpu32 = reference(xs, w, None, None, 1.0, 0, torch.float32)
        s = float(want.abs().max())
        e_gpu = float((torch.from_numpy(got).double() - want).abs().max()) / s
        e_cpu = float((cpu32.double() - want).abs().max()) / s
        print('scales %s x %g: %.2e (torch CPU fp32 %.2e)' % (scale_x, scale_w, e_gpu, e_cpu))
        assert np.isfinite(got).all() and e_gpu <= 3 * e_cpu + 2e-7
        assert np.array_equal(back, got)


def test_full_net_on_the_plane_path_vs_oracle_and_default_path(vr):
    """vr_set_option('conv_x3p', 1): the whole eval executor with plane tensors (skip connections written as fp32 + planes by the conv
    epilogue, decoder upsamples written as planes, thin tensors converted) against the CPU oracle under the bars of the default path
    (mask max 1e-4 / mean 1e-5) and against the default path itself."""
    from oracle import cascaded_net, weights
    sd = weights.make_state_dict(1234)
    model = vr.nets.CascadedNet(2048, 1024, 32, 128)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0')).eval()
    x = torch.rand(2, 2, 1025, 256, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = cascaded_net.predict_mask(x, sd)
    base = model.predict_mask(x.to('cuda:0')).cpu()
    try:
        model.set_option('conv_x3p', 1)
        got = model.predict_mask(x.to('cuda:0')).cpu()
        again = model.predict_mask(x.to('cuda:0')).cpu()
    finally:
        model.set_option('conv_x3p', -1)
    d = (got - want).abs()
    print('plane path: mask max %.2e mean %.2e vs oracle; max %.2e vs the default path' % (float(d.max()), float(d.mean()), float((got - base).abs().max())))
    assert float(d.max()) < 1e-4 and float(d.mean()) < 1e-5
    assert float((got - base).abs().max()) < 2e-5
    assert torch.equal(got, again)
    after = model.predict_mask(x.to('cuda:0')).cpu()
    assert torch.equal(after, base)                    # switching back restores the default executor exactly
