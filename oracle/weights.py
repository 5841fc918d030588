"""Seeded weight factory for CascadedNet.  TEST INFRASTRUCTURE.

``models/baseline.pth`` is not shipped with the reference (``models/`` holds
only ``.gitkeep``), so every parity and timing run uses seeded random weights
with non-trivial BatchNorm running statistics.  The key -> shape map restates
the module tree of ``lib/nets.py:46-80`` / ``lib/layers.py`` (689 entries for
the default ``CascadedNet(2048, 1024, 32, 128)``) and is pinned against the
reference's own ``state_dict()`` in ``tests/test_oracle_vs_reference.py``.
"""
import math

import torch


def _cba(spec, p, nin, nout, k):
    spec.append((p + '.conv.0.weight', (nout, nin, k, k), 'conv'))
    spec.append((p + '.conv.1.weight', (nout,), 'bn_w'))
    spec.append((p + '.conv.1.bias', (nout,), 'bn_b'))
    spec.append((p + '.conv.1.running_mean', (nout,), 'bn_rm'))
    spec.append((p + '.conv.1.running_var', (nout,), 'bn_rv'))
    spec.append((p + '.conv.1.num_batches_tracked', (), 'nbt'))


def _base_net(spec, p, nin, c, nin_lstm, nout_lstm):
    _cba(spec, p + '.enc1', nin, c, 3)
    for name, a, b in (('enc2', c, 2 * c), ('enc3', 2 * c, 4 * c), ('enc4', 4 * c, 6 * c), ('enc5', 6 * c, 8 * c)):
        _cba(spec, p + '.' + name + '.conv1', a, b, 3)
        _cba(spec, p + '.' + name + '.conv2', b, b, 3)
    _cba(spec, p + '.aspp.conv1.1', 8 * c, 8 * c, 1)
    _cba(spec, p + '.aspp.conv2', 8 * c, 8 * c, 1)
    for name in ('conv3', 'conv4', 'conv5'):
        _cba(spec, p + '.aspp.' + name, 8 * c, 8 * c, 3)
    _cba(spec, p + '.aspp.bottleneck', 40 * c, 8 * c, 1)
    _cba(spec, p + '.dec4.conv1', 14 * c, 6 * c, 3)
    _cba(spec, p + '.dec3.conv1', 10 * c, 4 * c, 3)
    _cba(spec, p + '.dec2.conv1', 6 * c, 2 * c, 3)
    q = p + '.lstm_dec2'
    _cba(spec, q + '.conv', 2 * c, 1, 1)
    hid = nout_lstm // 2
    for sfx in ('', '_reverse'):
        spec.append((q + '.lstm.weight_ih_l0' + sfx, (4 * hid, nin_lstm), 'lstm'))
        spec.append((q + '.lstm.weight_hh_l0' + sfx, (4 * hid, hid), 'lstm'))
        spec.append((q + '.lstm.bias_ih_l0' + sfx, (4 * hid,), 'lstm'))
        spec.append((q + '.lstm.bias_hh_l0' + sfx, (4 * hid,), 'lstm'))
    spec.append((q + '.dense.0.weight', (nin_lstm, nout_lstm), 'lin_w'))
    spec.append((q + '.dense.0.bias', (nin_lstm,), 'lin_b'))
    spec.append((q + '.dense.1.weight', (nin_lstm,), 'bn_w'))
    spec.append((q + '.dense.1.bias', (nin_lstm,), 'bn_b'))
    spec.append((q + '.dense.1.running_mean', (nin_lstm,), 'bn_rm'))
    spec.append((q + '.dense.1.running_var', (nin_lstm,), 'bn_rv'))
    spec.append((q + '.dense.1.num_batches_tracked', (), 'nbt'))
    _cba(spec, p + '.dec1.conv1', 3 * c + 1, c, 3)


def state_dict_spec(n_fft=2048, nout=32, nout_lstm=128):
    """Ordered (key, shape, kind) list in the reference's registration order."""
    nin = 2
    nin_lstm = (n_fft // 2) // 2
    spec = []
    _base_net(spec, 'stg1_low_band_net.0', nin, nout // 2, nin_lstm // 2, nout_lstm)
    _cba(spec, 'stg1_low_band_net.1', nout // 2, nout // 4, 1)
    _base_net(spec, 'stg1_high_band_net', nin, nout // 4, nin_lstm // 2, nout_lstm // 2)
    _base_net(spec, 'stg2_low_band_net.0', nout // 4 + nin, nout, nin_lstm // 2, nout_lstm)
    _cba(spec, 'stg2_low_band_net.1', nout, nout // 2, 1)
    _base_net(spec, 'stg2_high_band_net', nout // 4 + nin, nout // 2, nin_lstm // 2, nout_lstm // 2)
    _base_net(spec, 'stg3_full_band_net', 3 * nout // 4 + nin, nout, nin_lstm, nout_lstm)
    spec.append(('out.weight', (nin, nout, 1, 1), 'conv'))
    spec.append(('aux_out.weight', (nin, 3 * nout // 4, 1, 1), 'conv'))
    return spec


def make_state_dict(seed=1234, n_fft=2048, nout=32, nout_lstm=128):
    """Seeded random weights; BN affine and running stats perturbed so BN is not a no-op."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def uni(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    for key, shape, kind in state_dict_spec(n_fft, nout, nout_lstm):
        if kind == 'conv':
            fan_in = shape[1] * shape[2] * shape[3]
            b = math.sqrt(3.0 / fan_in) * 1.4
            sd[key] = uni(shape, -b, b)
        elif kind == 'lstm':
            hid = shape[0] // 4
            b = 1.0 / math.sqrt(hid)
            sd[key] = uni(shape, -b, b)
        elif kind == 'lin_w':
            b = 1.0 / math.sqrt(shape[1])
            sd[key] = uni(shape, -b, b)
        elif kind == 'lin_b':
            sd[key] = uni(shape, -0.1, 0.1)
        elif kind == 'bn_w':
            sd[key] = uni(shape, 0.7, 1.3)
        elif kind == 'bn_b':
            sd[key] = uni(shape, -0.15, 0.15)
        elif kind == 'bn_rm':
            sd[key] = uni(shape, -0.1, 0.1)
        elif kind == 'bn_rv':
            sd[key] = uni(shape, 0.6, 1.4)
        elif kind == 'nbt':
            sd[key] = torch.zeros((), dtype=torch.int64)
        else:
            raise AssertionError(kind)
    return sd


def clone_state_dict(sd):
    return {k: v.clone() for k, v in sd.items()}
