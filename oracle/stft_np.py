"""numpy restatement of the STFT / iSTFT the reference obtains from librosa.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: librosa
(``librosa~=0.10.0``, reference ``requirements.txt:4``) is a third-party
dependency that is absent here, and the reference has no golden vectors for
this boundary.  What is restated is librosa-0.10's documented default
behaviour at the reference's call sites:

* ``lib/spec_utils.py:26-31``  ``wave_to_spectrogram`` ->
  ``librosa.stft(wave[c], n_fft=n_fft, hop_length=hop_length)`` per channel:
  ``win_length = n_fft``, periodic Hann window
  (``scipy.signal.get_window('hann', n_fft, fftbins=True)``), ``center=True``
  with ``pad_mode='constant'`` (zeros, n_fft//2 each side),
  ``n_frames = 1 + len(y)//hop``, output complex64 ``[1 + n_fft//2, n_frames]``.
* ``lib/spec_utils.py:157-165`` ``spectrogram_to_wave`` ->
  ``librosa.istft(spec[c], hop_length=hop_length)``: ``n_fft = 2*(bins-1)``,
  same window, ``irfft`` * window, overlap-add into ``n_fft + hop*(T-1)``
  samples, divide by the window sum-of-squares where it exceeds
  ``finfo(float32).tiny``, trim ``n_fft//2`` on both sides
  -> ``hop*(T-1)`` float32 samples.
"""
import numpy as np


def hann_periodic(n_fft):
    """scipy.signal.get_window('hann', n_fft, fftbins=True) in closed form."""
    n = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)


def stft(y, n_fft=2048, hop_length=1024):
    """One channel.  y: float32 [L] -> complex64 [n_fft//2+1, 1 + L//hop]."""
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.concatenate([np.zeros(pad, np.float32), y, np.zeros(pad, np.float32)])
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    win = hann_periodic(n_fft)
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = yp[idx].astype(np.float64) * win[None, :]
    spec = np.fft.rfft(frames, n=n_fft, axis=1)            # [T, bins]
    return np.ascontiguousarray(spec.T).astype(np.complex64)


def istft(spec, hop_length=1024):
    """One channel.  spec: complex64 [bins, T] -> float32 [hop*(T-1)]."""
    spec = np.asarray(spec)
    bins, n_frames = spec.shape
    n_fft = 2 * (bins - 1)
    win = hann_periodic(n_fft)
    frames = np.fft.irfft(spec.T.astype(np.complex128), n=n_fft, axis=1) * win[None, :]
    total = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(total, np.float64)
    wss = np.zeros(total, np.float64)
    w2 = win * win
    for t in range(n_frames):
        s = t * hop_length
        y[s:s + n_fft] += frames[t]
        wss[s:s + n_fft] += w2
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2: total - n_fft // 2].astype(np.float32)


def wave_to_spectrogram(wave, hop_length, n_fft):
    """Restates lib/spec_utils.py:26-31: [2, L] float32 -> [2, bins, T] complex64."""
    return np.asarray([stft(wave[0], n_fft, hop_length), stft(wave[1], n_fft, hop_length)])


def spectrogram_to_wave(spec, hop_length=1024):
    """Restates lib/spec_utils.py:157-165."""
    if spec.ndim == 2:
        return istft(spec, hop_length)
    return np.asarray([istft(spec[0], hop_length), istft(spec[1], hop_length)])
