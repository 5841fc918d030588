"""torch-CPU fp32 restatement of the reference's CascadedNet.  TEST INFRASTRUCTURE.

Functional form over a plain ``state_dict`` (reference key names), so it
travels to the GPU box where ``/root/reference`` does not exist.  Each function
cites the reference lines it restates; ``tests/test_oracle_vs_reference.py``
pins it against the reference's own modules executed on CPU.

BatchNorm handling: ``training=False`` uses running statistics; with
``training=True`` batch statistics are used and, when ``update_running`` is
true, the running buffers in ``sd`` are updated in place exactly like
``nn.BatchNorm2d(momentum=0.1)`` (unbiased variance for the running update).
Dropout2d (``lib/layers.py:90,102-103``) is replaced by injectable per-(sample,
channel) keep-masks ``dropout[prefix]`` of shape [N, C] holding 0 or 1/0.9.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn(x, sd, p, training, update_running):
    """BatchNorm{1,2}d of lib/layers.py:21,120 (eps 1e-5, momentum 0.1, affine)."""
    w, b = sd[p + '.weight'], sd[p + '.bias']
    if not training:
        return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], w, b,
                            False, BN_MOMENTUM, BN_EPS)
    dims = [d for d in range(x.dim()) if d != 1]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    if update_running:
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            sd[p + '.running_mean'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            sd[p + '.running_var'].mul_(1 - BN_MOMENTUM).add_(
                BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            sd[p + '.num_batches_tracked'] += 1
    shape = [1, -1] + [1] * (x.dim() - 2)
    xh = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    return xh * w.view(shape) + b.view(shape)


def conv_bn_act(x, sd, p, stride=1, pad=1, dil=1, act='relu', training=False, update_running=True):
    """layers.Conv2DBNActiv, lib/layers.py:8-26 (conv bias=False -> BN -> activ)."""
    h = F.conv2d(x, sd[p + '.conv.0.weight'], None, stride, pad, dil)
    h = _bn(h, sd, p + '.conv.1', training, update_running)
    if act == 'relu':
        return F.relu(h)
    if act == 'leaky':
        return F.leaky_relu(h, 0.01)
    raise ValueError(act)


def encoder(x, sd, p, stride, **kw):
    """layers.Encoder, lib/layers.py:29-40: conv(stride) -> conv(1), LeakyReLU."""
    h = conv_bn_act(x, sd, p + '.conv1', stride, 1, 1, 'leaky', **kw)
    return conv_bn_act(h, sd, p + '.conv2', 1, 1, 1, 'leaky', **kw)


def crop_center(h1, h2):
    """spec_utils.crop_center, lib/spec_utils.py:8-23 (time axis only)."""
    if h1.shape[3] == h2.shape[3]:
        return h1
    if h1.shape[3] < h2.shape[3]:
        raise ValueError('h1_shape[3] must be greater than h2_shape[3]')
    s = (h1.shape[3] - h2.shape[3]) // 2
    return h1[:, :, :, s:s + h2.shape[3]]


def decoder(x, skip, sd, p, **kw):
    """layers.Decoder, lib/layers.py:43-64 (bilinear x2 align_corners, cat skip, one conv)."""
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    if skip is not None:
        x = torch.cat([x, crop_center(skip, x)], dim=1)
    return conv_bn_act(x, sd, p + '.conv1', 1, 1, 1, 'relu', **kw)


def aspp(x, sd, p, dilations, dropout=None, **kw):
    """layers.ASPPModule, lib/layers.py:67-105."""
    _, _, h, w = x.shape
    pooled = x.mean(dim=2, keepdim=True)                      # AdaptiveAvgPool2d((1, None))
    f1 = conv_bn_act(pooled, sd, p + '.conv1.1', 1, 0, 1, 'relu', **kw)
    f1 = F.interpolate(f1, size=(h, w), mode='bilinear', align_corners=True)
    f2 = conv_bn_act(x, sd, p + '.conv2', 1, 0, 1, 'relu', **kw)
    f3 = conv_bn_act(x, sd, p + '.conv3', 1, dilations[0], dilations[0], 'relu', **kw)
    f4 = conv_bn_act(x, sd, p + '.conv4', 1, dilations[1], dilations[1], 'relu', **kw)
    f5 = conv_bn_act(x, sd, p + '.conv5', 1, dilations[2], dilations[2], 'relu', **kw)
    out = torch.cat((f1, f2, f3, f4, f5), dim=1)
    out = conv_bn_act(out, sd, p + '.bottleneck', 1, 0, 1, 'relu', **kw)
    if dropout is not None:
        out = out * dropout[:, :, None, None]
    return out


def bilstm(x, sd, p):
    """nn.LSTM(bidirectional=True) of lib/layers.py:113-117; x [T, N, I] -> [T, N, 2H].

    Gate order i, f, g, o; gates = x W_ih^T + b_ih + h W_hh^T + b_hh.
    """
    T, N, _ = x.shape
    outs = []
    for sfx in ('', '_reverse'):
        w_ih, w_hh = sd[p + '.weight_ih_l0' + sfx], sd[p + '.weight_hh_l0' + sfx]
        bias = sd[p + '.bias_ih_l0' + sfx] + sd[p + '.bias_hh_l0' + sfx]
        H = w_hh.shape[1]
        gx = x @ w_ih.t() + bias                               # [T, N, 4H]
        h = x.new_zeros(N, H)
        c = x.new_zeros(N, H)
        hs = [None] * T
        order = range(T) if sfx == '' else range(T - 1, -1, -1)
        for t in order:
            g = gx[t] + h @ w_hh.t()
            i, f, gg, o = g.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs, 0))
    return torch.cat(outs, dim=2)


def lstm_module(x, sd, p, **kw):
    """layers.LSTMModule, lib/layers.py:108-133."""
    N, _, nbins, nframes = x.shape
    h = conv_bn_act(x, sd, p + '.conv', 1, 0, 1, 'relu', **kw)[:, 0]     # N, nbins, nframes
    h = h.permute(2, 0, 1)                                                # nframes, N, nbins
    h = bilstm(h, sd, p + '.lstm')
    h = h.reshape(-1, h.shape[-1]) @ sd[p + '.dense.0.weight'].t() + sd[p + '.dense.0.bias']
    h = F.relu(_bn(h, sd, p + '.dense.1', kw.get('training', False), kw.get('update_running', True)))
    h = h.reshape(nframes, N, 1, nbins)
    return h.permute(1, 2, 3, 0)


DILATIONS = ((4, 2), (8, 4), (12, 6))

# Set to a dict to record intermediate activations (test diagnostics): keys '<net prefix>.e1' ...
TAPS = None


def _tap(name, t):
    if TAPS is not None:
        TAPS[name] = t.detach().clone()


def base_net(x, sd, p, dropout=None, **kw):
    """nets.BaseNet, lib/nets.py:8-41."""
    e1 = conv_bn_act(x, sd, p + '.enc1', 1, 1, 1, 'relu', **kw)
    e2 = encoder(e1, sd, p + '.enc2', 2, **kw)
    e3 = encoder(e2, sd, p + '.enc3', 2, **kw)
    e4 = encoder(e3, sd, p + '.enc4', 2, **kw)
    e5 = encoder(e4, sd, p + '.enc5', 2, **kw)
    for i, e in enumerate((e1, e2, e3, e4, e5)):
        _tap('%s.e%d' % (p, i + 1), e)
    h = aspp(e5, sd, p + '.aspp', DILATIONS, dropout=None if dropout is None else dropout.get(p + '.aspp'), **kw)
    _tap(p + '.aspp', h)
    h = decoder(h, e4, sd, p + '.dec4', **kw)
    _tap(p + '.dec4', h)
    h = decoder(h, e3, sd, p + '.dec3', **kw)
    _tap(p + '.dec3', h)
    h = decoder(h, e2, sd, p + '.dec2', **kw)
    _tap(p + '.dec2', h)
    lo = lstm_module(h, sd, p + '.lstm_dec2', **kw)
    _tap(p + '.lstm', lo)
    h = torch.cat([h, lo], dim=1)
    h = decoder(h, e1, sd, p + '.dec1', **kw)
    _tap(p + '.dec1', h)
    return h


BASE_NETS = ('stg1_low_band_net.0', 'stg1_high_band_net', 'stg2_low_band_net.0',
             'stg2_high_band_net', 'stg3_full_band_net')


def forward(x, sd, n_fft=2048, training=False, update_running=True, dropout=None):
    """CascadedNet.forward, lib/nets.py:82-117 (is_complex=False). x [B,2,n_fft/2+1,T]."""
    kw = dict(training=training, update_running=update_running)
    max_bin = n_fft // 2
    output_bin = n_fft // 2 + 1
    x = x[:, :, :max_bin]
    bandw = x.shape[2] // 2
    l1_in, h1_in = x[:, :, :bandw], x[:, :, bandw:]
    l1 = base_net(l1_in, sd, 'stg1_low_band_net.0', dropout, **kw)
    l1 = conv_bn_act(l1, sd, 'stg1_low_band_net.1', 1, 0, 1, 'relu', **kw)
    h1 = base_net(h1_in, sd, 'stg1_high_band_net', dropout, **kw)
    aux1 = torch.cat([l1, h1], dim=2)
    l2 = base_net(torch.cat([l1_in, l1], dim=1), sd, 'stg2_low_band_net.0', dropout, **kw)
    l2 = conv_bn_act(l2, sd, 'stg2_low_band_net.1', 1, 0, 1, 'relu', **kw)
    h2 = base_net(torch.cat([h1_in, h1], dim=1), sd, 'stg2_high_band_net', dropout, **kw)
    aux2 = torch.cat([l2, h2], dim=2)
    f3 = base_net(torch.cat([x, aux1, aux2], dim=1), sd, 'stg3_full_band_net', dropout, **kw)
    mask = torch.sigmoid(F.conv2d(f3, sd['out.weight']))
    return F.pad(mask, (0, 0, 0, output_bin - mask.shape[2]), mode='replicate')


def predict_mask(x, sd, n_fft=2048, offset=64):
    """CascadedNet.predict_mask, lib/nets.py:124-131."""
    mask = forward(x, sd, n_fft)
    if offset > 0:
        mask = mask[:, :, :, offset:-offset]
        assert mask.shape[3] > 0
    return mask


def predict(x, sd, n_fft=2048, offset=64):
    """CascadedNet.predict, lib/nets.py:133-141."""
    pred = x * forward(x, sd, n_fft)
    if offset > 0:
        pred = pred[:, :, :, offset:-offset]
        assert pred.shape[3] > 0
    return pred
