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


def merge_artifacts(y_mask, thres=0.05, min_range=64, fade_size=32):
    """lib/spec_utils.py:60-93 with numpy in / numpy out, for callers that keep the reference's own `Separator._postprocess`
    (inference.py:26-30).  The O(T) run logic (which frames get blended, the linear fades, the IndexError on a mask with no
    frame above `thres`, the ValueError on min_range < 2 * fade_size) is the library's host half of --postprocess -- the same
    code `vr_separate(..., postprocess)` runs; the blend `y_mask += weight * (1 - y_mask)` is applied in place, like there.
    (`Separator(postprocess=True)` of this package does all of it on the device instead.)"""
    y_mask = np.asarray(y_mask)
    T = y_mask.shape[2]
    frame_min = np.ascontiguousarray(y_mask.min(axis=(0, 1)), dtype=np.float32)
    weight = np.empty(T, dtype=np.float32)
    native.check(native.lib().vr_debug_merge_artifacts_weight(native.np_ptr(frame_min), T, float(thres), int(min_range),
                                                              int(fade_size), native.np_ptr(weight)))
    y_mask += weight.astype(y_mask.dtype)[None, None, :] * (1 - y_mask)
    return y_mask


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


def _head_lag(a, b, sr, seconds=4):
    """Lag (samples, positive = `a` starts late) at which the first `seconds` of the two stereo waves, summed to mono with the
    mean removed, correlate best -- argmax of the full cross-correlation, evaluated on the GPU (vr_xcorr_argmax)."""
    import ctypes
    heads = []
    for w in (a, b):
        m = w[:, :sr * seconds].sum(axis=0)
        heads.append(np.ascontiguousarray(m - m.mean(), dtype=np.float32))
    device = int(os.environ.get('VR_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    k = ctypes.c_int64()
    native.check(native.lib().vr_xcorr_argmax(device, native.np_ptr(heads[0]), len(heads[0]), native.np_ptr(heads[1]), len(heads[1]),
                                              ctypes.byref(k)))
    return int(k.value) - (len(heads[0]) - 1)        # index of the 'full' correlation -> lag (the reference's own zero point)


def align_wave_head_and_tail(a, b, sr):
    """What lib/spec_utils.py:96-119 does to a (mixture, instrumental) pair before the STFT: strip leading / trailing silence
    of each (librosa.effects.trim), drop the head of whichever starts late by the cross-correlation lag, cut both to the
    shorter length."""
    a, b = audio.trim(a)[0], audio.trim(b)[0]
    lag = _head_lag(a, b, sr)
    if lag > 0:
        a = a[:, lag:]
    else:
        b = b[:, -lag:]
    n = min(a.shape[1], b.shape[1])
    return a[:, :n], b[:, :n]


class SpectrogramCache(object):
    """The reference's spectrogram cache (lib/spec_utils.py:122-154) as an on-disk contract: the STFT of <dir>/<song>.<ext>
    lives in <dir>/sr{sr}_hl{hop}_nf{n_fft}/<song>.npy as [T, 2, bins] complex64 (time-major, so that the training set can
    seek-read cropsize rows, lib/dataset.py:15-46), and is computed from the ALIGNED pair, never from one file alone."""

    def __init__(self, sr, hop_length, n_fft):
        self.sr, self.hop_length, self.n_fft = sr, hop_length, n_fft
        self.folder = 'sr{}_hl{}_nf{}'.format(sr, hop_length, n_fft)

    def npy_path(self, audio_path):
        folder = os.path.join(os.path.dirname(audio_path), self.folder)
        os.makedirs(folder, exist_ok=True)
        return os.path.join(folder, os.path.splitext(os.path.basename(audio_path))[0] + '.npy')

    def pair(self, mix_path, inst_path):
        """-> X, y [2, bins, T] complex64 and their .npy paths; decodes + aligns + transforms only on a cache miss."""
        paths = [self.npy_path(mix_path), self.npy_path(inst_path)]
        if all(os.path.exists(p) for p in paths):
            specs = [np.load(p).transpose(1, 2, 0) for p in paths]
        else:
            waves = [audio.load(p, sr=self.sr, mono=False, dtype=np.float32, res_type='kaiser_fast')[0] for p in (mix_path, inst_path)]
            specs = [wave_to_spectrogram(w, self.hop_length, self.n_fft) for w in align_wave_head_and_tail(waves[0], waves[1], self.sr)]
            for p, spec in zip(paths, specs):
                np.save(p, spec.transpose(2, 0, 1))
        assert specs[0].shape == specs[1].shape
        return specs[0], specs[1], paths[0], paths[1]


def cache_or_load(mix_path, inst_path, sr, hop_length, n_fft):
    """The reference's entry point to the cache (lib/spec_utils.py:122): X, y, X_cache_path, y_cache_path."""
    return SpectrogramCache(sr, hop_length, n_fft).pair(mix_path, inst_path)
