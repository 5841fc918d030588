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


def _ref_merge(reference_lib):
    import importlib
    return importlib.import_module('lib.spec_utils').merge_artifacts


def _mask_from_frame_min(fmin):
    m = np.full((2, 5, len(fmin)), 0.9, np.float32)
    m[1, 3] = fmin
    return m


MERGE_CASES = {
    'one_long_run': lambda T: np.where((np.arange(T) > 40) & (np.arange(T) < 300), 0.5, 0.01),
    'starts_at_zero': lambda T: np.where(np.arange(T) < 200, 0.4, 0.0),
    'two_runs_close': lambda T: np.where(((np.arange(T) > 10) & (np.arange(T) < 120)) | ((np.arange(T) > 130) & (np.arange(T) < 330)), 0.3, 0.02),
    'short_runs_only': lambda T: np.where((np.arange(T) % 50) < 30, 0.6, 0.01),
    'runs_to_the_end': lambda T: np.where(np.arange(T) > 250, 0.2, 0.04),
    'all_above': lambda T: np.full(T, 0.5),
}


@pytest.mark.parametrize('name', sorted(MERGE_CASES))
def test_merge_artifacts_host_logic_matches_reference(built, reference_lib, name):
    """--postprocess: the library's host half and the oracle restatement vs lib/spec_utils.merge_artifacts."""
    ref = _ref_merge(reference_lib)
    T = 400
    fmin = MERGE_CASES[name](T).astype(np.float32)
    mask = _mask_from_frame_min(fmin)
    want = ref(mask.copy())
    got_oracle = separator.merge_artifacts(mask.copy())
    assert np.abs(got_oracle - want).max() < 1e-7
    w = np.empty(T, np.float32)
    L = built.native.lib()
    frame_min = np.ascontiguousarray(mask.min(axis=(0, 1)))      # keep alive: np_ptr does not hold a reference
    built.native.check(L.vr_debug_merge_artifacts_weight(built.native.np_ptr(frame_min), T, 0.05, 64, 32,
                                                         built.native.np_ptr(w)))
    got = mask + w[None, None, :] * (1 - mask)
    assert np.abs(got - want).max() < 1e-6


def test_merge_artifacts_error_behaviour(built, reference_lib):
    ref = _ref_merge(reference_lib)
    T = 300
    mask = _mask_from_frame_min(np.zeros(T, np.float32))
    with pytest.raises(IndexError):
        ref(mask.copy())
    with pytest.raises(IndexError):
        separator.merge_artifacts(mask.copy())
    w = np.empty(T, np.float32)
    L = built.native.lib()
    zeros_f, ones_f = np.zeros(T, np.float32), np.ones(T, np.float32)
    with pytest.raises(IndexError):
        built.native.check(L.vr_debug_merge_artifacts_weight(built.native.np_ptr(zeros_f), T, 0.05, 64, 32,
                                                             built.native.np_ptr(w)))
    with pytest.raises(ValueError):      # min_range < 2 * fade_size (lib/spec_utils.py:61-62)
        built.native.check(L.vr_debug_merge_artifacts_weight(built.native.np_ptr(ones_f), T, 0.05, 32, 32,
                                                             built.native.np_ptr(w)))
