"""spec_utils look-alikes (lib/spec_utils.py:8-31,157-165) backed by the HIP STFT / iSTFT kernels."""
import os

import numpy as np

from . import native

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
