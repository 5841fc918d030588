"""spec_utils look-alikes (lib/spec_utils.py:8-31,96-165) backed by the HIP STFT / iSTFT / audio kernels."""
import os

import numpy as np

from . import audio, native

_handles = {}


def _signal_handle(n_fft, hop_length):
    """A handle used only for its FFT plan; one per (n_fft, hop) on the process's GPU."""
    key = (int(n_fft), int(hop_length))
    if key not in _handles:
        device = int(os.environ.get('VR_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        _handles[key] = native.Handle(device, n_fft, hop_length, 4, 4)
    return _handles[key]


def crop_center(h1, h2):
    """lib/spec_utils.py:8-23: centre-crop h1 on the time axis to h2's width (views only)."""
    h1_shape = h1.size()
    h2_shape = h2.size()
    if h1_shape[3] == h2_shape[3]:
        return h1
    elif h1_shape[3] < h2_shape[3]:
        raise ValueError('h1_shape[3] must be greater than h2_shape[3]')
    s_time = (h1_shape[3] - h2_shape[3]) // 2
    e_time = s_time + h2_shape[3]
    return h1[:, :, :, s_time:e_time]


def wave_to_spectrogram(wave, hop_length, n_fft):
    """lib/spec_utils.py:26-31: [2, L] float32 -> [2, n_fft/2+1, 1 + L//hop] complex64."""
    wave = np.ascontiguousarray(np.asarray(wave, dtype=np.float32))
    if wave.ndim != 2 or wave.shape[0] != 2:
        raise ValueError('wave must be [2, L]')
    L = wave.shape[1]
    T = 1 + L // hop_length
    spec = np.empty((2, n_fft // 2 + 1, T), dtype=np.complex64)
    h = _signal_handle(n_fft, hop_length)
    native.check(native.lib().vr_stft(h.h, native.np_ptr(wave), 0, L, native.np_ptr(spec), 0))
    return spec


def spectrogram_to_wave(spec, hop_length=1024):
    """lib/spec_utils.py:157-165: [2, bins, T] (or [bins, T]) complex64 -> float32 wave."""
    spec = np.asarray(spec)
    mono = spec.ndim == 2
    if mono:
        spec = np.asarray([spec, spec])
    spec = np.ascontiguousarray(spec.astype(np.complex64))
    bins, T = spec.shape[1], spec.shape[2]
    n_fft = 2 * (bins - 1)
    wave = np.empty((2, hop_length * (T - 1)), dtype=np.float32)
    h = _signal_handle(n_fft, hop_length)
    native.check(native.lib().vr_istft(h.h, native.np_ptr(spec), 0, T, native.np_ptr(wave), 0))
    return wave[0] if mono else wave


def align_wave_head_and_tail(a, b, sr):
    """lib/spec_utils.py:96-119: trim both, cross-correlate the first 4 s of the mono sums, shift by the best lag,
    cut to the common length.  The O(N^2) `np.correlate(..., 'full')` of the reference runs on the GPU (vr_xcorr_argmax)."""
    import ctypes
    a, _ = audio.trim(a)
    b, _ = audio.trim(b)
    a_mono = a[:, :sr * 4].sum(axis=0)
    b_mono = b[:, :sr * 4].sum(axis=0)
    a_mono = np.ascontiguousarray(a_mono - a_mono.mean(), dtype=np.float32)
    b_mono = np.ascontiguousarray(b_mono - b_mono.mean(), dtype=np.float32)
    offset = len(a_mono) - 1
    best = ctypes.c_int64()
    device = int(os.environ.get('VR_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    native.check(native.lib().vr_xcorr_argmax(device, native.np_ptr(a_mono), len(a_mono), native.np_ptr(b_mono), len(b_mono),
                                              ctypes.byref(best)))
    # np.correlate(a, b, 'full')[k] pairs a[n + k - (len(b) - 1)] with b[n]; the reference subtracts len(a) - 1
    delay = int(best.value) - offset
    if delay > 0:
        a = a[:, delay:]
    else:
        b = b[:, np.abs(delay):]
    if a.shape[1] < b.shape[1]:
        b = b[:, :a.shape[1]]
    else:
        a = a[:, :b.shape[1]]
    return a, b


def cache_or_load(mix_path, inst_path, sr, hop_length, n_fft):
    """lib/spec_utils.py:122-154: same cache directories, file names and on-disk layout ([T, 2, bins] complex64 .npy)."""
    mix_basename = os.path.splitext(os.path.basename(mix_path))[0]
    inst_basename = os.path.splitext(os.path.basename(inst_path))[0]
    cache_dir = 'sr{}_hl{}_nf{}'.format(sr, hop_length, n_fft)
    mix_cache_dir = os.path.join(os.path.dirname(mix_path), cache_dir)
    inst_cache_dir = os.path.join(os.path.dirname(inst_path), cache_dir)
    os.makedirs(mix_cache_dir, exist_ok=True)
    os.makedirs(inst_cache_dir, exist_ok=True)
    mix_cache_path = os.path.join(mix_cache_dir, mix_basename + '.npy')
    inst_cache_path = os.path.join(inst_cache_dir, inst_basename + '.npy')
    if os.path.exists(mix_cache_path) and os.path.exists(inst_cache_path):
        X = np.load(mix_cache_path).transpose(1, 2, 0)
        y = np.load(inst_cache_path).transpose(1, 2, 0)
    else:
        X, _ = audio.load(mix_path, sr=sr, mono=False, dtype=np.float32, res_type='kaiser_fast')
        y, _ = audio.load(inst_path, sr=sr, mono=False, dtype=np.float32, res_type='kaiser_fast')
        X, y = align_wave_head_and_tail(X, y, sr)
        X = wave_to_spectrogram(X, hop_length, n_fft)
        y = wave_to_spectrogram(y, hop_length, n_fft)
        np.save(mix_cache_path, X.transpose(2, 0, 1))
        np.save(inst_cache_path, y.transpose(2, 0, 1))
    assert X.shape == y.shape
    return X, y, mix_cache_path, inst_cache_path
