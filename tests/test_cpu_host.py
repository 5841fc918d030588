"""CPU-only checks: the C-ABI library builds/loads and exports what include/vr_mi355.h declares,
host-side logic matches the oracle, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import __graft_entry__
from oracle import separator, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    __graft_entry__.build()
    return __graft_entry__.load_package()


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'vr_mi355.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.native.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), 'missing export: ' + name
    assert sorted(built.native.exported_symbols()) == declared, 'ctypes binding and header disagree'


def test_no_cpu_fallback(built):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    net = built.nets.CascadedNet(512, 256, 8, 32)
    with pytest.raises(RuntimeError):
        net.predict_mask(torch.rand(1, 2, 257, 160))         # no handle -> loud failure
    with pytest.raises(built.native.VRError):
        net.to(torch.device('cuda:0'))                        # vr_create: no HIP device
    assert built.native.lib().vr_last_error().decode().startswith('no HIP device')


def test_null_handle_is_an_error_not_a_crash(built):
    L = built.native.lib()
    assert L.vr_set_mode(None, 0) == -2
    assert L.vr_num_params(None) == -2
    assert L.vr_destroy(None) == -2


def test_state_spec_matches_oracle_spec(built):
    for cfg in ((2048, 32, 128), (512, 8, 32)):
        mine = built.nets.state_spec(*cfg)
        ref = weights.state_dict_spec(*cfg)
        assert [(k, tuple(s)) for k, s, _ in mine] == [(k, tuple(s)) for k, s, _ in ref]
    net = built.nets.CascadedNet(2048, 1024, 32, 128)
    sd = net.state_dict()
    assert len(sd) == 689
    assert sum(v.numel() for k, v in sd.items() if v.dtype == torch.float32 and 'running' not in k) == 14740882
    assert net.offset == 64 and net.max_bin == 1024 and net.output_bin == 1025


def test_load_state_dict_semantics(built):
    net = built.nets.CascadedNet(512, 256, 8, 32)
    sd = weights.make_state_dict(3, n_fft=512, nout=8, nout_lstm=32)
    net.load_state_dict(sd)
    back = net.state_dict()
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    bad = dict(sd)
    bad.pop('out.weight')
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)
    bad = dict(sd)
    bad['out.weight'] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)


def test_make_padding_and_crop_center(built):
    for width in (1, 127, 128, 129, 1292, 1280, 5000):
        for crop, off in ((256, 64), (160, 64), (512, 64)):
            assert built.dataset.make_padding(width, crop, off) == separator.make_padding(width, crop, off)
    assert built.dataset.make_padding(1292, 256, 64) == (64, 180, 128)
    a = torch.arange(2 * 3 * 4 * 10.).reshape(2, 3, 4, 10)
    b = torch.zeros(2, 3, 4, 6)
    assert torch.equal(built.spec_utils.crop_center(a, b), a[:, :, :, 2:8])
    assert built.spec_utils.crop_center(a, a) is a
    with pytest.raises(ValueError):
        built.spec_utils.crop_center(b, a)


def test_dropin_shadow_modules_resolve(built):
    import importlib.util
    p = os.path.join(ROOT, 'vocal-remover_amd', 'dropin', 'lib', 'nets.py')
    spec = importlib.util.spec_from_file_location('dropin_nets_probe', p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.CascadedNet is built.nets.CascadedNet
