"""Isolated GPU parity of every non-conv kernel of the training path (through vr_debug_kernel, C ABI) against torch
autograd / torch ops of the same operation in fp64: BatchNorm train forward statistics + backward
(bn_finalize / bn_bwd_reduce / finalize / apply), BiLSTM forward + BPTT + W_hh gradient, bilinear x2 upsample and its
transpose, the frequency average pool / broadcast backward, the thin (Cout 1 or 2) 1x1 conv gradients, the
sigmoid . X L1 head, Linear's BatchNorm1d+ReLU rows pass and bias sums, and the fused Adam.

Tolerance: 1e-4 of the reference tensor's max-abs (fp32 kernels vs an fp64 reference); the conv kernels have their
own tests (test_gpu_parity.py, test_gpu_train.py).  Reference semantics: autograd of lib/layers.py:8-133 and
train.py:81-96.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope='module')
def handle(vr):
    model = vr.nets.CascadedNet(512, 256, 8, 32)
    model.to(torch.device('cuda:0'))
    return vr.native, model._handle


def f32(t):
    return np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def close(got, want, what, tol=TOL):
    want = want.detach().double().numpy() if torch.is_tensor(want) else np.asarray(want, np.float64)
    scale = float(np.abs(want).max()) + 1e-30
    err = float(np.abs(got.astype(np.float64) - want).max()) / scale
    assert err < tol, '%s: max-abs/scale = %.3e' % (what, err)
    return err


def act(v, slope):
    return torch.where(v > 0, v, v * slope)


@pytest.mark.parametrize('shape,slope,use_post', [((2, 16, 32, 64), 0.0, False), ((3, 5, 7, 10), 0.01, True),
                                                  ((4, 1, 64, 32), 0.0, False), ((2, 40, 16, 16), 0.01, True),
                                                  # more than 16384 elements per channel: several reduce blocks per channel, the last one finalises
                                                  ((4, 6, 128, 64), 0.01, True), ((16, 3, 64, 256), 0.0, False)])
def test_batchnorm_train_forward_stats_and_backward(handle, shape, slope, use_post):
    nat, h = handle
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + C)
    z = (torch.randn(shape, generator=g) * 1.5 + torch.randn(1, C, 1, 1, generator=g)).float()
    G = torch.randn(shape, generator=g).float()
    gamma = (torch.rand(C, generator=g) + 0.5).float()
    beta = (torch.randn(C, generator=g) * 0.3).float()
    rm0 = (torch.randn(C, generator=g) * 0.1).float()
    rv0 = (torch.rand(C, generator=g) + 0.5).float()
    post = ((torch.rand(N, C, generator=g) > 0.2).float() / 0.9) if use_post else None
    # reference: autograd through F.batch_norm (training) -> activation -> Dropout2d keep-mask, in fp64
    zd = z.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    y = F.batch_norm(zd, rm, rv, gd, bd, True, 0.1, 1e-5)
    v = act(y, slope)
    if post is not None:
        v = v * post.double()[:, :, None, None]
    v.backward(G.double())
    mean = z.double().mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.double().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    aff_want = torch.stack([gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd], 1)
    out = [np.empty(shape, np.float32), np.empty(C, np.float32), np.empty(C, np.float32), np.empty((C, 2), np.float32),
           np.empty(C, np.float32), np.empty(C, np.float32)]
    nat.debug_kernel(h, 'bn_backward', shape, [slope, 1e-5, 0.1],
                     [f32(z), f32(G), f32(gamma), f32(beta), f32(post) if post is not None else None, f32(rm0), f32(rv0)], out)
    close(out[0], zd.grad, 'dz')
    close(out[1], gd.grad, 'dgamma')
    close(out[2], bd.grad, 'dbeta')
    close(out[3], aff_want, 'affine (scale, shift)', 1e-5)
    close(out[4], rm, 'running_mean', 1e-5)
    close(out[5], rv, 'running_var (unbiased)', 1e-5)


# (11, 72, 64): eleven samples over the eight sample slices of lstm_whh_grad_kernel, a ragged last frame block; (16, 256, 32): the
# benched batch
@pytest.mark.parametrize('N,T,H', [(2, 128, 64), (3, 128, 32), (1, 40, 16), (11, 72, 64), (16, 256, 32)])
def test_bilstm_forward_bptt_and_whh_gradient(handle, N, T, H):
    nat, h = handle
    g = torch.Generator().manual_seed(T + H)
    G4 = 4 * H
    gx = (torch.randn(N, 2 * G4, T, generator=g) * 0.8).float()
    whh = [(torch.rand(G4, H, generator=g) * 2 - 1).float() / H ** 0.5 for _ in range(2)]
    dh = torch.randn(N, 2 * H, T, generator=g).float()
    gxd = gx.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in whh]
    outs = [None] * 2
    for d in range(2):                      # torch.nn.LSTM cell, gate order i, f, g, o (lib/layers.py:113-117)
        hcur = torch.zeros(N, H, dtype=torch.float64)
        c = torch.zeros(N, H, dtype=torch.float64)
        seq = [None] * T
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            gates = gxd[:, d * G4:(d + 1) * G4, t] + hcur @ wd[d].t()
            i, f, gg, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            hcur = torch.sigmoid(o) * torch.tanh(c)
            seq[t] = hcur
        outs[d] = torch.stack(seq, dim=2)
    hout = torch.cat(outs, dim=1)
    hout.backward(dh.double())
    out = [np.empty((N, 2 * H, T), np.float32), np.empty((N, 2 * G4, T), np.float32), np.empty((G4, H), np.float32),
           np.empty((G4, H), np.float32)]
    nat.debug_kernel(h, 'lstm', [N, T, H], [], [f32(gx), f32(whh[0]), f32(whh[1]), f32(dh)], out)
    close(out[0], hout, 'h')
    close(out[1], gxd.grad, 'dgx (BPTT)')
    close(out[2], wd[0].grad, 'dW_hh forward')
    close(out[3], wd[1].grad, 'dW_hh reverse')


@pytest.mark.parametrize('shape', [(2, 3, 8, 16), (1, 2, 5, 7), (2, 4, 16, 1), (1, 6, 1, 12), (1, 2, 33, 20)])
def test_bilinear_upsample_and_its_transpose(handle, shape):
    nat, h = handle
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(shape, generator=g).float()
    dhi = torch.randn(N, C, 2 * H, 2 * W, generator=g).float()
    xd = x.double().requires_grad_(True)
    up = F.interpolate(xd, scale_factor=2, mode='bilinear', align_corners=True)      # lib/layers.py:52
    up.backward(dhi.double())
    out = [np.empty((N, C, 2 * H, 2 * W), np.float32), np.empty(shape, np.float32)]
    nat.debug_kernel(h, 'upsample', shape, [], [f32(x), f32(dhi)], out)
    close(out[0], up, 'upsample x2', 1e-5)
    close(out[1], xd.grad, 'upsample backward')


@pytest.mark.parametrize('shape', [(2, 8, 32, 16), (1, 5, 3, 9)])
def test_frequency_avgpool_forward_backward_and_broadcast_backward(handle, shape):
    nat, h = handle
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.randn(shape, generator=g).float()
    gp = torch.randn(N, C, W, generator=g).float()
    d = torch.randn(shape, generator=g).float()
    xd = x.double().requires_grad_(True)
    pooled = F.adaptive_avg_pool2d(xd, (1, None))[:, :, 0]                           # lib/layers.py:72
    pooled.backward(gp.double())
    out = [np.empty((N, C, W), np.float32), np.empty(shape, np.float32), np.empty((N, C, W), np.float32)]
    nat.debug_kernel(h, 'pool', shape, [], [f32(x), f32(gp), f32(d)], out)
    close(out[0], pooled, 'avgpool over frequency', 1e-5)
    close(out[1], xd.grad, 'avgpool backward', 1e-5)
    close(out[2], d.double().sum(dim=2), 'backward of the broadcast along frequency', 1e-5)


@pytest.mark.parametrize('shape,CO,slope,use_aff', [((2, 16, 32, 64), 1, 0.0, True), ((1, 12, 9, 8), 1, 0.01, False),
                                                    ((2, 32, 64, 32), 2, 0.0, True), ((3, 5, 4, 12), 2, 1.0, False)])
def test_thin_conv_forward_dgrad_wgrad(handle, shape, CO, slope, use_aff):
    nat, h = handle
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C + CO)
    x = torch.randn(shape, generator=g).float()
    aff = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3], 1).float() if use_aff else None
    w = (torch.randn(CO, C, generator=g) / C ** 0.5).float()
    dz = torch.randn(N, CO, H, W, generator=g).float()
    v = x.double()
    if aff is not None:
        v = v * aff[:, 0].double().view(1, -1, 1, 1) + aff[:, 1].double().view(1, -1, 1, 1)
    a = act(v, slope).detach().requires_grad_(True)          # the value consumers see is the leaf
    wd = w.double().requires_grad_(True)
    z = F.conv2d(a, wd.view(CO, C, 1, 1))
    z.backward(dz.double())
    out = [np.empty(shape, np.float32), np.empty((CO, C), np.float32), np.empty((N, H, W), np.float32)]
    nat.debug_kernel(h, 'thin', list(shape) + [CO], [slope], [f32(x), f32(aff) if aff is not None else None, f32(w), f32(dz)], out)
    close(out[0] * 0.5, a.grad, 'thin dgrad')               # (the hook stores the gradient, then accumulates it once more)
    close(out[1], wd.grad, 'thin wgrad')
    if CO == 1:
        close(out[2], z[:, 0], 'squeeze conv forward')


@pytest.mark.parametrize('N,C,H,W', [(2, 8, 16, 32), (1, 32, 64, 16)])
def test_sigmoid_mask_l1_head(handle, N, C, H, W):
    """train.py:81,89: mask = sigmoid(out(h)) replicate-padded by one bin (lib/nets.py:109-115); L1(mask * X, y)."""
    nat, h = handle
    bins = H + 1
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, H, W, generator=g).float()
    aff = torch.stack([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3], 1).float()
    w = (torch.randn(2, C, generator=g) / C ** 0.5).float()
    X = torch.rand(N, 2, bins, W, generator=g).float()
    Y = (X * torch.rand(N, 2, bins, W, generator=g)).float()
    gscale = 1.0 / (X.numel() * 3)
    a = act(x.double() * aff[:, 0].double().view(1, -1, 1, 1) + aff[:, 1].double().view(1, -1, 1, 1), 0.0)
    logits = F.conv2d(a, w.double().view(2, C, 1, 1)).requires_grad_(True)
    mask = F.pad(torch.sigmoid(logits), (0, 0, 0, 1), mode='replicate')
    loss_sum = (mask * X.double() - Y.double()).abs().sum()
    (loss_sum * gscale).backward()
    out = [np.empty((N, 2, H, W), np.float32), np.empty((N, 2, bins, W), np.float32), np.empty(1, np.float32)]
    nat.debug_kernel(h, 'head_loss', [N, C, H, W, bins], [0.0, gscale], [f32(x), f32(aff), f32(w), f32(X), f32(Y)], out)
    close(out[0], logits.grad, 'dLoss/dlogit')
    close(out[1], mask, 'mask', 1e-5)
    assert abs(float(out[2][0]) - float(loss_sum) / X.numel()) < 2e-6


def test_linear_batchnorm1d_rows_pass_and_bias_sums(handle):
    nat, h = handle
    N, R, W = 3, 48, 20
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, R, W, generator=g).float()
    aff = torch.stack([torch.rand(R, generator=g) + 0.5, torch.randn(R, generator=g) * 0.3], 1).float()
    d = torch.randn(N, R, W, generator=g).float()
    out = [np.empty((N, R, W), np.float32), np.empty(R, np.float32)]
    nat.debug_kernel(h, 'rows', [N, R, W], [], [f32(x), f32(aff), f32(d)], out)
    close(out[0], torch.relu(x.double() * aff[:, 0].double().view(1, -1, 1) + aff[:, 1].double().view(1, -1, 1)), 'BN1d+ReLU', 1e-5)
    close(out[1], d.double().sum(dim=(0, 2)), 'bias gradient (channel sums)', 1e-5)


@pytest.mark.parametrize('step', [1, 7])
def test_fused_adam_vs_torch_optim(handle, step):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) (train.py:215-218): `step` steps of torch's optimizer on the
    same gradients, the library's kernel doing the last one from torch's own state."""
    nat, h = handle
    n = 5000
    g = torch.Generator().manual_seed(step)
    p = torch.randn(n, generator=g).double().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3)
    grads = [torch.randn(n, generator=g).double() * 0.1 for _ in range(step)]
    for gi in grads[:-1]:
        p.grad = gi.clone()
        opt.step()
    p_before = p.detach().clone()
    st = opt.state[p] if step > 1 else None
    m0 = st['exp_avg'].clone() if st else torch.zeros(n, dtype=torch.float64)
    v0 = st['exp_avg_sq'].clone() if st else torch.zeros(n, dtype=torch.float64)
    p.grad = grads[-1].clone()
    opt.step()
    gscale = 0.5                                   # the library multiplies every gradient first (1/world)
    out = [np.empty(n, np.float32) for _ in range(3)]
    nat.debug_kernel(h, 'adam', [n], [1e-3, 0.9, 0.999, 1e-8, gscale, float(step)],
                     [f32(p_before), f32(grads[-1] / gscale), f32(m0), f32(v0)], out)
    assert float(np.abs(out[0] - p.detach().numpy()).max()) < 2e-6
    close(out[1], opt.state[p]['exp_avg'], 'exp_avg', 1e-5)
    close(out[2], opt.state[p]['exp_avg_sq'], 'exp_avg_sq', 1e-5)
