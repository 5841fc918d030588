"""numpy restatement of the audio front end the reference gets from librosa / resampy.  TEST INFRASTRUCTURE.

`resample_kaiser_fast` restates resampy 0.4's band-limited interpolation (resampy/interpn.py `_resample_loop`,
resampy/filters.py `sinc_window`) with the published 'kaiser_fast' parameters -- 16 zero crossings, precision 9
(512 table samples per crossing), roll-off 0.85, Kaiser beta 8.555504641634386 -- as called by
`librosa.load(path, sr=44100, mono=False, dtype=np.float32, res_type='kaiser_fast')`
(/root/reference/inference.py:136-138, lib/spec_utils.py:139-142), including librosa.resample's
`fix_length(ceil(n * ratio))`.  resampy and librosa are third-party dependencies pinned in requirements.txt
(librosa~=0.10.0, resampy~=0.4.0) and absent from /root/reference and from this image: PARITY UNPINNED; the tests anchor
the restatement on its defining properties instead (unit DC gain, a tone keeps frequency and amplitude, the stop band is
attenuated).
"""
import numpy as np
import scipy.signal


def kaiser_fast_table():
    num_zeros, precision, rolloff, beta = 16, 9, 0.85, 8.555504641634386
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample_kaiser_fast(x, sr_orig, sr_new):
    """x [..., n] float32 -> [..., ceil(n * sr_new / sr_orig)] float32."""
    x = np.asarray(x, dtype=np.float32)
    ratio = float(sr_new) / sr_orig
    n_in = x.shape[-1]
    n_core = int(n_in * ratio)
    n_out = int(np.ceil(n_in * ratio))
    win, precision = kaiser_fast_table()
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * precision)
    nwin = win.shape[0]
    xs = x.reshape(-1, n_in)
    y = np.zeros((xs.shape[0], n_out), dtype=np.float32)
    for t in range(n_core):
        time_register = t * time_increment
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * precision
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        acc = np.zeros(xs.shape[0], dtype=np.float32)
        for i in range(i_max):
            k = offset + i * index_step
            acc = (acc.astype(np.float64) + (win[k] + eta * delta[k]) * xs[:, n - i].astype(np.float64)).astype(np.float32)
        frac = scale - frac
        index_frac = frac * precision
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_in - n - 1, (nwin - offset) // index_step)
        for k2 in range(k_max):
            k = offset + k2 * index_step
            acc = (acc.astype(np.float64) + (win[k] + eta * delta[k]) * xs[:, n + k2 + 1].astype(np.float64)).astype(np.float32)
        y[:, t] = acc
    return y.reshape(x.shape[:-1] + (n_out,))


def align_head_and_tail_delay(a_mono, b_mono):
    """The lag lib/spec_utils.py:107-108 computes: argmax(np.correlate(a, b, 'full')) - (len(a) - 1)."""
    return int(np.argmax(np.correlate(a_mono, b_mono, 'full'))) - (len(a_mono) - 1)
