"""GPU parity at the BENCHED train configuration (configs[3]: batch 16 x [2,1025,256]) -- VERDICT r2 "what's weak" 1, 2, 5.

Kernel dispatch depends on the grid size (conv_wino.hip: 64-cout variant only from 600 workgroups; conv_dma.hip: cout tile and
8x16 / 16x16 pixel tiles by workgroup count; conv_x3.hip: tile choice by workgroup count), so batch 16 launches variants that the
batch-2 full-net test never reaches, and the train executor runs three streams.  Here:

  * full net, batch 16: the three-stream executor vs serial_exec=1 (every kernel on ONE stream) and vs itself eight times --
    loss, mask and every gradient;
  * single convs (forward, data gradient, weight gradient) at batch-16 grids on both sides of each dispatch threshold,
    vs torch autograd, 2e-4 of the tensor's scale -- in the fp32-MFMA mode and in the split-bf16 mode;
  * bf16-operand mode (configs[4] arithmetic) vs the fp32 mode, GPU vs GPU at batch 16 (no CPU oracle fits this batch);
  * split-bf16 mode ("as exact as fp32") on inputs with subnormals, 2^+-100 scales and values on / next to bf16 rounding
    boundaries, against an fp64 reference.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import train_step, weights

pytestmark = pytest.mark.gpu

HOP, N_FFT = 1024, 2048
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def full16(vr):
    sd = weights.make_state_dict(1234)
    model = vr.nets.CascadedNet(N_FFT, HOP, 32, 128)
    model.load_state_dict(sd)
    model.to(torch.device(DEV))
    X, y = train_step.synth_batch(16, T=256, n_fft=N_FFT, seed=3)
    masks = train_step.dropout_masks(16, seed=5, nout=32)
    return model, sd, X.to(DEV), y.to(DEV), masks


def _step(model, sd, X, y, masks, **options):
    """One train step from the same weights: loss, mask, {key: gradient}."""
    try:
        model.load_state_dict(sd)
        for k, v in options.items():
            model.set_option(k, v)
        model.train()
        model.set_dropout_masks(masks)
        model.zero_grad()
        loss, mask = model.train_step(X, y, 1, return_mask=True)
        return loss, mask.cpu(), model.grads()
    finally:
        model.set_dropout_masks(None)
        for k in options:
            model.set_option(k, -1 if k == 'mfma_mode' else 0)
        model.eval()


def _scale(t):
    return float(t.abs().max()) + 1e-30


def test_b16_train_step_three_streams_vs_one(vr, full16):
    """Forward band fork, side-stream weight gradients and the third backward stream only reorder independent kernels (and the
    order in which several consumers add into one activation gradient): against the same kernels on ONE stream every gradient
    agrees to 1e-6 of its scale, and repeated concurrent runs agree with each other to the same bound."""
    model, sd, X, y, masks = full16
    for mode in (0, 2, 3):
        loss_s, mask_s, g_s = _step(model, sd, X, y, masks, serial_exec=1, mfma_mode=mode)
        # (eight repetitions: the round-3 weight_hh nondeterminism -- DESIGN.md hardware fact 5 -- showed in one run of four to eight)
        runs = [_step(model, sd, X, y, masks, mfma_mode=mode) for _ in range(8)]
        worst, bit_equal = 0.0, True
        for loss, mask, g in runs:
            assert abs(loss - loss_s) <= 1e-6 * abs(loss_s), (mode, loss, loss_s)
            assert float((mask - mask_s).abs().max()) <= 1e-6
            for k in g_s:
                e = float((g[k] - g_s[k]).abs().max()) / _scale(g_s[k])
                worst = max(worst, e)
                assert e <= 1e-6, (mode, k, e)
                bit_equal = bit_equal and torch.equal(g[k], runs[0][2][k])
        print('mfma_mode %d, batch 16: three-stream vs one-stream gradients worst %.2e of scale; run-to-run bit-equal: %s'
              % (mode, worst, bit_equal))


# N, Cin, H, W, Cout, ks, stride, dh, dw   -- workgroup counts on both sides of the dispatch thresholds at batch 16
B16_CONVS = [
    (16, 64, 160, 64, 64, 3, 1, 1, 1),      # 3x3 s1, 64 couts: 16*20*2 = 640 tiles -> 64-cout Winograd / x3<64,8>
    (16, 64, 144, 64, 64, 3, 1, 1, 1),      # 576 tiles -> 32-cout Winograd variant
    (16, 128, 64, 32, 128, 3, 1, 1, 1),     # 16*8*1 tiles x 2 cout tiles = 256 workgroups: halved cout tile
    (16, 32, 128, 128, 32, 3, 1, 1, 1),     # 32 couts, 16-row tiles (>= 1024 workgroups) vs
    (16, 32, 48, 64, 32, 3, 1, 1, 1),       # 8-row tiles
    (16, 97, 64, 64, 32, 3, 1, 1, 1),       # decoder shape, Cin = 97 (partial channel chunk)
    (16, 64, 128, 64, 192, 3, 2, 1, 1),     # stride 2: 16*8*1 tiles x 3 = 384 < 768 -> 32-cout tile; 16-wide? no: Wout = 32
    (16, 64, 256, 64, 128, 3, 2, 1, 1),     # 16*16*1 x 2 = 512 -> 32; x (128/64) ...
    (16, 64, 256, 128, 64, 3, 2, 1, 1),     # 16*16*2 x 1 = 512 -> 32-cout tile, Wout = 64
    (16, 32, 512, 128, 64, 3, 2, 1, 1),     # 16*32*2 = 1024 >= 768 -> 64-cout stride-2 tile
    (16, 128, 32, 16, 128, 3, 1, 1, 1),     # 1/16 resolution, 16 columns: 16*2*1*4 = 128 workgroups -> 8x16 tiles
    (16, 256, 128, 16, 256, 3, 1, 4, 2),    # dilated, 16*8*1*8 = 1024 >= 800 -> 16x16 tiles
    (16, 128, 32, 16, 128, 3, 1, 8, 4),     # dilated, 8x16 tiles
    (16, 640, 32, 16, 128, 1, 1, 1, 1),     # 1x1 bottleneck, 16 columns
    (16, 32, 256, 64, 16, 1, 1, 1, 1),      # 1x1 tail
]


@pytest.mark.parametrize('mode', [0, 2, 3, 1], ids=['fp32_mfma', 'split_bf16', 'split_fp16', 'bf16_operands'])
@pytest.mark.parametrize('case', B16_CONVS, ids=[str(c) for c in B16_CONVS])
def test_b16_conv_dispatch_variants_vs_autograd(vr, full16, case, mode):
    """fp32 modes: 2e-4 of the tensor's scale.  bf16-operand mode (configs[4] arithmetic, kernels that have it): 2e-2 -- operands
    rounded to 8 significant bits, fp32 accumulation over >= 288 products."""
    N, Cin, H, W, Cout, ks, stride, dh, dw = case
    model = full16[0]
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).requires_grad_(True)
    a = x.clone().requires_grad_(True)
    pad = (dh, dw) if ks == 3 else (0, 0)
    torch.set_num_threads(16)
    out = F.conv2d(a, w, None, stride, pad, (dh, dw))
    dz = torch.randn(out.shape, generator=g)
    out.backward(dz)
    nat = vr.native
    xn, wn, dzn = x.numpy(), w.detach().numpy(), dz.numpy()
    got = np.empty(tuple(out.shape), np.float32)
    dx = np.empty(tuple(x.shape), np.float32)
    dwt = np.empty(tuple(w.shape), np.float32)
    try:
        model.set_option('mfma_mode', mode)
        flags = 2 if (ks == 3 and stride == 1 and dh == 1) else 0          # transformed / split weights for the 3x3 stride-1 kernels
        nat.check(nat.lib().vr_debug_conv2d(model._handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, ks, stride, dh, dw,
                                            flags, None, ctypes.c_float(1.0), None, nat.np_ptr(got), None))
        nat.check(nat.lib().vr_debug_conv2d_backward(model._handle.h, nat.np_ptr(xn), N, Cin, H, W, nat.np_ptr(wn), Cout, ks, stride,
                                                     dh, dw, 0, None, ctypes.c_float(1.0), nat.np_ptr(dzn), nat.np_ptr(dx),
                                                     nat.np_ptr(dwt)))
    finally:
        model.set_option('mfma_mode', -1)
    ef = float(np.abs(got - out.detach().numpy()).max() / out.detach().abs().max())
    ex = float(np.abs(dx - a.grad.numpy()).max() / a.grad.abs().max())
    ew = float(np.abs(dwt - w.grad.numpy()).max() / w.grad.abs().max())
    tol = 2e-2 if mode == 1 else 2e-4
    assert ef < tol and ex < tol and ew < tol, 'forward %.2e dgrad %.2e wgrad %.2e' % (ef, ex, ew)


def test_b16_bf16_mode_vs_fp32_mode(vr, full16):
    """configs[4] arithmetic (bf16 MFMA operands in the Winograd / GEMM kernels, fp32 everything else) against the fp32 modes on
    the SAME GPU at the benched batch, and -- as the yardstick -- the two fp32 modes (0: fp32 MFMA / Winograd, 2: split-bf16
    direct) against each other: those differ by fp32 rounding only.

    Measured (printed): the two fp32 modes give the same loss to 8 digits and gradients with cosine 0.9998.  The bf16-operand mode
    does NOT track fp32 at this depth, at batch 16 either: loss within 7e-6, gradient norm within 0.3 %, but mask mean-abs 4e-2
    and gradient cosine 0.37 globally (0.45 median per tensor, also for the >= 64K-element conv weights).  Every bf16 KERNEL is
    pinned at 2e-2 of its output scale in the dispatch test above, so this is the arithmetic (8 significant bits per operand
    through ~100 convolutions and train-mode BatchNorms of a randomly initialised net), not a defect of one kernel."""
    model, sd, X, y, masks = full16
    loss2, mask2, g2 = _step(model, sd, X, y, masks, mfma_mode=2)
    loss0, mask0, g0 = _step(model, sd, X, y, masks, mfma_mode=0)
    loss1, mask1, g1 = _step(model, sd, X, y, masks, mfma_bf16=1)

    def compare(ga, gb):
        rows, dot, na, nb = [], 0.0, 0.0, 0.0
        for k in ga:
            if k.endswith('dense.0.bias') or float(ga[k].norm()) == 0.0:        # exact gradient 0: rounding noise only
                continue
            a, b = ga[k].double().flatten(), gb[k].double().flatten()
            dot += float(a @ b); na += float(a @ a); nb += float(b @ b)
            if a.numel() >= 64:
                rows.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), float(b.norm() / a.norm()), a.numel(), k))
        rows.sort()
        return rows, dot / (na ** 0.5 * nb ** 0.5), (nb / na) ** 0.5

    rows1, gcos1, gratio1 = compare(g2, g1)
    rows0, gcos0, gratio0 = compare(g2, g0)
    print('\n'.join('%-60s cos %.5f  |g| ratio %.4f  n=%d' % (k, c, r, n) for c, r, n, k in rows1[:8]))
    big1 = [r for r in rows1 if r[2] >= 65536]
    print('fp32 mode 0 vs mode 2, batch 16: loss %.8f vs %.8f; gradient: global cosine %.6f, per-tensor cosine min %.4f median %.5f'
          % (loss0, loss2, gcos0, rows0[0][0], float(np.median([r[0] for r in rows0]))))
    print('bf16 mode vs fp32 (mode 2), batch 16: loss %.8f vs %.8f; mask mean-abs %.2e max-abs %.2e; gradient: global cosine %.4f, '
          '|g| ratio %.4f; per-tensor cosine min %.4f median %.4f; tensors >= 64K elements: min %.4f median %.4f'
          % (loss1, loss2, float((mask1 - mask2).abs().mean()), float((mask1 - mask2).abs().max()), gcos1, gratio1, rows1[0][0],
             float(np.median([r[0] for r in rows1])), min(r[0] for r in big1), float(np.median([r[0] for r in big1]))))
    # the two fp32 modes agree to rounding (amplified by ~100 train-mode BatchNorms: measured mask 1.4e-4, global cosine 0.99976)
    assert abs(loss0 - loss2) <= 1e-6 * abs(loss2) and float((mask0 - mask2).abs().max()) <= 5e-4
    assert gcos0 >= 0.999 and rows0[0][0] >= 0.99
    # bf16 operands: the loss agrees (measured 6.5e-6 relative) and the gradient NORM is right (ratio 1.003), but 2^-9 operand
    # rounding through ~100 convolutions moves the mask by 4e-2 on average at random initialisation and leaves the gradient
    # DIRECTION only weakly correlated with fp32's (global cosine 0.37, per-tensor median 0.45) -- measured, not a kernel defect
    # (every bf16 kernel is within 2e-2 of its output scale above).  The bars are sanity bounds around that measurement.
    assert abs(loss1 - loss2) <= 1e-3 * abs(loss2)
    assert float((mask1 - mask2).abs().mean()) <= 0.1
    assert gcos1 >= 0.2 and 0.95 <= gratio1 <= 1.05


SPECIAL = ['subnormal', 'scale_2^-100', 'scale_2^+100', 'bf16_boundaries', 'chunk_scales', 'descending_chunks']


@pytest.mark.parametrize('kind', SPECIAL)
@pytest.mark.parametrize('shape', [(2, 40, 32, 64, 64), (2, 24, 40, 64, 32)], ids=['64couts', '32couts'])
def test_split_bf16_mode_is_fp32_exact_on_special_values(vr, full16, kind, shape):
    """x = bf16(x) + bf16(x - x1) + (x - x1 - x2) must stay exact where rounding to bf16 is delicate: fp32 subnormals (the third
    plane underflows bf16's range only below 2^-133), huge / tiny scales, values exactly on a bf16 grid point, half way between
    two, and one fp32 ulp to either side.  Error vs fp64 at most 1.5x an fp32 DIRECT convolution's (the larger of torch's CPU
    conv and this library's fp32-MFMA direct kernel) + 1e-7 of the output scale."""
    N, Cin, H, W, Cout = shape
    model = full16[0]
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((N, Cin, H, W)) * np.exp(rng.standard_normal((N, Cin, H, W)))).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9.0)).astype(np.float32)
    if kind == 'subnormal':
        sel = rng.random(x.shape) < 0.3
        x[sel] = (rng.standard_normal(int(sel.sum())) * 2.0 ** -140).astype(np.float32)
    elif kind == 'scale_2^-100':
        x = (x * np.float32(2.0 ** -100)).astype(np.float32)
    elif kind == 'scale_2^+100':
        x = (x * np.float32(2.0 ** 100)).astype(np.float32)
    elif kind == 'chunk_scales':
        # every 8-channel chunk and every 8-row band at its own power of two (2^-40 .. 2^+40), weights per cout likewise: the running
        # shift of conv_x3h.hip has to move up AND down inside one workgroup, and the per-cout weight scale covers 80 binades
        x = x * (2.0 ** rng.integers(-40, 41, size=(1, (Cin + 7) // 8, (H + 7) // 8, 1))).repeat(8, 1)[:, :Cin].repeat(8, 2)[:, :, :H].astype(np.float32)
        w = (w * (2.0 ** rng.integers(-40, 41, size=(Cout, 1, 1, 1)))).astype(np.float32)
    elif kind == 'descending_chunks':
        # chunk after chunk 2^50 smaller (2^40 ... 2^-120 and below): the running shift of conv_x3h.hip wants to follow, but the
        # accumulators still hold the first chunk's sums -- it may not run more than 2^64 ahead of the largest chunk (overflow)
        x = (x * (2.0 ** np.clip(40 - 50 * np.arange((Cin + 7) // 8), -125, 40)).reshape(1, -1, 1, 1).repeat(8, 1)[:, :Cin]).astype(np.float32)
    else:
        u = x.view(np.uint32).copy()
        r = rng.integers(0, 5, size=x.shape)
        u = np.where(r == 0, u & 0xffff0000, u)                       # exact bf16 value
        u = np.where(r == 1, (u & 0xffff0000) | 0x8000, u)            # half way between two bf16 values (ties-to-even case)
        u = np.where(r == 2, (u & 0xffff0000) | 0x7fff, u)            # one ulp below the tie
        u = np.where(r == 3, (u & 0xffff0000) | 0x8001, u)            # one ulp above the tie
        x = u.astype(np.uint32).view(np.float32)
        wu = w.view(np.uint32).copy()
        w = np.where(rng.random(w.shape) < 0.5, (wu & 0xffffff00) | 0x80, wu).astype(np.uint32).view(np.float32)
    want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, 1, 1).numpy()
    cpu32 = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 1, 1).numpy()
    # (chunk_scales: one scale per output channel, the weights of different couts differ by up to 2^80)
    scale = np.abs(want).max(axis=(0, 2, 3), keepdims=True) if kind == 'chunk_scales' else float(np.abs(want).max())
    nat = vr.native
    errs = {'torch cpu fp32': float((np.abs(cpu32.astype(np.float64) - want) / scale).max())}
    # fp32-MFMA direct kernel (mode 0, plain weights), fp32-MFMA Winograd (mode 0, transformed weights), split-bf16 direct (mode 2)
    for key, mode, flags in (('fp32 MFMA direct', 0, 0), ('fp32 MFMA Winograd', 0, 2), ('split-bf16', 2, 2), ('split-fp16', 3, 2)):
        got = np.empty(want.shape, np.float32)
        try:
            model.set_option('mfma_mode', mode)
            nat.check(nat.lib().vr_debug_conv2d(model._handle.h, nat.np_ptr(x), N, Cin, H, W, nat.np_ptr(w), Cout, 3, 1, 1, 1, flags, None,
                                                ctypes.c_float(1.0), None, nat.np_ptr(got), None))
        finally:
            model.set_option('mfma_mode', -1)
        assert np.isfinite(got).all()
        errs[key] = float((np.abs(got.astype(np.float64) - want) / scale).max())
    print('%s %s: max error / scale  ' % (kind, shape) + '  '.join('%s %.3e' % kv for kv in errs.items()))
    # as exact as an fp32 DIRECT convolution (the Winograd form sums 2.25x fewer products and sits below all of them)
    assert errs['split-bf16'] <= 1.5 * max(errs['fp32 MFMA direct'], errs['torch cpu fp32']) + 1e-7
    # mfma_mode 3 (conv_x3h.hip): operands scaled by exact powers of two per tile and chunk, two fp16 planes, three products
    assert errs['split-fp16'] <= 2.5 * max(errs['fp32 MFMA direct'], errs['torch cpu fp32']) + 2e-7


def test_b16_train_step_vs_cpu_oracle(vr, full16):
    """The BENCHED train configuration against the oracle ("missing" 4): full net, batch 16 x [2,1025,256], one body of
    train.py:77-96 -- loss, train-mode mask, every BatchNorm running statistic, every gradient -- vs the fp32 CPU oracle at the
    SAME batch (~40 GB of host memory, ~30-60 s; the fp64 oracle at batch 16 would need ~95 GB and minutes, it calibrates the
    batch-2 test in test_gpu_configs.py instead).  Both sides round in fp32 in different orders, so the gradient bars are those
    of two fp32 evaluations against each other: per tensor rel-L2 <= 6e-2 (>= 16 elements), median <= 3e-2, the vector of
    per-tensor gradient norms within 2 %, global cosine >= 0.999 (measured: median 2.6e-2, norms within 1.1 %, cosine 0.99964); loss 2e-6 (measured: equal to 8 digits); mask 5e-4 (the GPU's own two fp32 modes differ by
    1.4e-4 at this batch, test_b16_fp32_modes_agree); running statistics 1e-4 of scale.  Both multiply modes, same bars."""
    from oracle import cascaded_net
    model, sd, X, y, masks = full16
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    Xc, yc = X.cpu(), y.cpu()
    sd32 = weights.clone_state_dict(sd)
    loss_c, g_c = train_step.loss_and_grads(sd32, Xc, yc, n_fft=N_FFT, dropout=masks)          # updates sd32's running stats
    with torch.no_grad():
        mask_c = cascaded_net.forward(Xc, weights.clone_state_dict(sd), N_FFT, training=True, update_running=False, dropout=masks)
    for mode in (0, 2, 3):
        try:
            model.load_state_dict(sd)
            model.set_option('mfma_mode', mode)
            model.train()
            model.set_dropout_masks(masks)
            model.zero_grad()
            loss, mask = model.train_step(X, y, 1, return_mask=True)
            grads = model.grads()
            state = model.state_dict()
        finally:
            model.set_dropout_masks(None)
            model.set_option('mfma_mode', -1)
            model.load_state_dict(sd)
            model.eval()
        assert abs(loss - loss_c) < 2e-6, (mode, loss, loss_c)
        e_mask = float((mask.cpu() - mask_c).abs().max())
        rel, bad, dot, n_g, n_c, norm_dev = [], [], 0.0, 0.0, 0.0, 0.0
        small_a, small_b = [], []
        for k in g_c:
            if k.endswith('dense.0.bias'):
                assert float(grads[k].abs().max()) < 1e-6, k           # exact gradient 0: a BatchNorm follows the bias
                continue
            a, b = grads[k].double(), g_c[k].double()
            e = float((a - b).norm() / (b.norm() + 1e-30))
            rel.append((e, k))
            dot += float((a * b).sum()); n_g += float((a * a).sum()); n_c += float((b * b).sum())
            if b.numel() >= 16:
                norm_dev = max(norm_dev, abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0))
                if e > 6e-2:
                    bad.append('%s %.3e' % (k, e))
            else:
                # the 1-element BatchNorm weight / bias of an LSTM squeeze conv: a sum of 0.5 M products of either sign that cancels
                # to a few per cent of its terms, so the 2-3 % disagreement of two fp32 evaluations upstream shows as tens of per
                # cent -- or, where the sum nearly vanishes, as a multiple -- of the value itself (measured 0.60 / 2.3; batch 2 vs
                # fp64: CPU fp32 0.06, GPU 0.15-0.35).  A relative error of a near-zero scalar says nothing: these tensors are
                # compared as ONE concatenated vector; the reduction itself is pinned at 1e-4 against fp64 in test_gpu_kernels.py
                small_a.append(a.reshape(-1)); small_b.append(b.reshape(-1))
        rel.sort(reverse=True)
        med = float(np.median([r[0] for r in rel]))
        cos = dot / (n_g * n_c) ** 0.5
        print('mfma_mode %d, batch 16 vs fp32 CPU oracle: loss %.8f / %.8f; mask max-abs %.2e; gradient rel-L2 median %.2e, worst %s '
              '%.2e; per-tensor norm deviation max %.2e; global cosine %.6f' % (mode, loss, loss_c, e_mask, med, rel[0][1], rel[0][0],
                                                                                norm_dev, cos))
        sa, sb = torch.cat(small_a), torch.cat(small_b)
        e_small = float((sa - sb).norm() / sb.norm())
        print('            the %d tensors with < 16 elements as one vector: rel-L2 %.2e' % (len(small_a), e_small))
        assert not bad, '\n'.join(bad)
        assert med < 3e-2 and cos > 0.999 and norm_dev < 2e-2 and e_mask < 5e-4 and e_small < 0.25
        for k in sd32:
            if k.endswith('running_mean') or k.endswith('running_var'):
                scale = float(sd32[k].abs().max()) + 1e-6
                assert float((state[k] - sd32[k]).abs().max()) < 1e-4 * scale, k
            elif k.endswith('num_batches_tracked'):
                assert int(state[k]) == int(sd32[k]), k


B16_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'b16_fp64_small_grads.npz')


def test_b16_small_gradients_vs_fp64_fixture(vr, full16):
    """Where the fp32-vs-fp32 comparison above is blind (VERDICT r4 "weak" 2): the tensors with fewer than 16 elements -- the ten 1-element
    BatchNorm weights / biases of the LSTM squeeze convs among them -- were only compared as one concatenated vector against another
    fp32 evaluation.  Here every tensor of at most 4096 elements (all BatchNorm weights / biases) is compared, one by one, with the
    **fp64** oracle gradient of the SAME batch-16 step (tests/golden/b16_fp64_small_grads.npz, generated by
    tests/golden/make_golden_b16.py on the GPU box's host: ~85 GB of memory), and the bar of each tensor is calibrated by what the fp32
    CPU oracle itself loses against fp64 on that tensor (also in the fixture):
        rel-L2 error vs fp64  <=  3 x max(the fp32 CPU oracle's error on this tensor, the upper quartile of that error over its size class)
    and, for every 1-element tensor the fp32 CPU oracle gets to better than 50 %, the SIGN must be right.  Norms of ALL 367 gradients
    vs fp64: within 3 x the CPU oracle's own deviation + 1 %.  Modes 3 (default) and 0."""
    assert os.path.exists(B16_FIXTURE), 'run tests/golden/make_golden_b16.py (see its docstring)'
    fx = np.load(B16_FIXTURE)
    keys = [str(k) for k in fx['keys']]
    cpu_err = dict(zip(keys, fx['cpu32_err'])); norm64 = dict(zip(keys, fx['norm64'])); numel = dict(zip(keys, fx['numel']))
    small = [k for k in keys if 'g64/' + k in fx.files and not k.endswith('dense.0.bias')]
    tiny = [k for k in small if numel[k] < 16]
    assert len(tiny) >= 10 and len(small) >= 200
    q_tiny = float(np.quantile([cpu_err[k] for k in tiny], 0.75))
    q_rest = float(np.quantile([cpu_err[k] for k in small if numel[k] >= 16], 0.75))
    model, sd, X, y, masks = full16
    for mode in (3, 0):
        loss, _, grads = _step(model, sd, X, y, masks, mfma_mode=mode)
        assert abs(loss - float(fx['loss64'])) < 2e-6, (mode, loss, float(fx['loss64']))
        bad, rows = [], []
        for k in small:
            g64 = torch.from_numpy(fx['g64/' + k]).reshape(grads[k].shape)
            e = float((grads[k].double().cpu() - g64).norm() / max(float(g64.norm()), 1e-300))
            bar = 3.0 * max(float(cpu_err[k]), q_tiny if numel[k] < 16 else q_rest)
            rows.append((e / bar, e, float(cpu_err[k]), k))
            if e > bar:
                bad.append('%s: %.3e > %.3e (fp32 CPU oracle: %.3e)' % (k, e, bar, cpu_err[k]))
            if numel[k] == 1 and cpu_err[k] < 0.5:
                assert float(grads[k].flatten()[0]) * float(g64.flatten()[0]) > 0, 'sign of %s: %r vs fp64 %r' % (k, float(grads[k].flatten()[0]), float(g64.flatten()[0]))
        rows.sort(reverse=True)
        print('mfma_mode %d vs the fp64 fixture at batch 16: %d tensors <= 4096 elements; worst (error / bar): %s' % (
            mode, len(small), '; '.join('%s %.2e (cpu fp32 %.2e)' % (k.replace('.conv.1.', '.bn.'), e, c) for _, e, c, k in rows[:4])))
        for k in tiny:
            g64 = float(fx['g64/' + k].flatten()[0]) if numel[k] == 1 else None
            if g64 is not None:
                print('    %-52s fp64 %+.5e  gpu %+.5e  (gpu err %.2e, fp32 CPU oracle err %.2e)' % (k, g64, float(grads[k].flatten()[0]),
                      abs(float(grads[k].flatten()[0]) - g64) / max(abs(g64), 1e-300), cpu_err[k]))
        assert not bad, '\n'.join(bad)
        for k in keys:
            if k.endswith('dense.0.bias') or norm64[k] == 0.0:
                continue
            dev = abs(float(grads[k].double().norm()) / norm64[k] - 1.0)
            assert dev <= 3.0 * float(cpu_err[k]) + 1e-2, (k, dev, float(cpu_err[k]))


def test_b16_configs4_slice_vs_fp32(vr, full16):
    """configs[4] as this library runs it (bench.py `train_bf16`): the data-parallel step with bf16 where it is exact or harmless -- the
    3x3 stride-1 convolutions on the 16-bit matrix pipe with split products (mfma_mode 3, the default) and the gradient bucket rounded to bf16,
    all-reduced by RCCL in bf16 and widened back (here world 1: the same kernels and the same rounding, one rank) -- against the fp32
    step (mfma_mode 0, fp32 bucket) at the benched batch.  Bars: loss 1e-4 relative, global gradient cosine
    >= 0.99, per-tensor cosine median >= 0.98."""
    from vocal_remover_amd import train as vtrain
    model, sd, X, y, masks = full16

    def step(mode, wire):
        try:
            model.load_state_dict(sd)
            model.set_option('mfma_mode', mode)
            trainer = vtrain.Trainer(model, lr=1e-3, world_size=1, rank=0, backend='rccl', wire=wire, dropout=True, broadcast=False)
            model.set_dropout_masks(masks)
            model.zero_grad()
            loss = model.train_step(X, y, 1)
            trainer.reduce()                                         # vr_allreduce_grads in the wire format
            torch.cuda.synchronize()
            return loss, model.grads()
        finally:
            model.set_dropout_masks(None)
            model.set_option('mfma_mode', -1)
            model.eval()

    try:
        loss_a, g_a = step(0, 'fp32')
        loss_b, g_b = step(3, 'bf16')
    except vr.native.VRError as e:
        pytest.skip('RCCL not usable on this box: %s' % e)
    dot = na = nb = 0.0
    cos = []
    for k in g_a:
        if k.endswith('dense.0.bias') or float(g_a[k].norm()) == 0.0:
            continue
        a, b = g_a[k].double().flatten(), g_b[k].double().flatten()
        dot += float(a @ b); na += float(a @ a); nb += float(b @ b)
        if a.numel() >= 64:
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-300)))
    gcos, med = dot / (na * nb) ** 0.5, float(np.median(cos))
    print('configs[4] slice vs fp32 at batch 16: loss %.8f vs %.8f; gradient global cosine %.6f, per-tensor median %.5f, min %.4f'
          % (loss_b, loss_a, gcos, med, min(cos)))
    assert abs(loss_b - loss_a) <= 1e-4 * abs(loss_a)
    assert gcos >= 0.99 and med >= 0.98
