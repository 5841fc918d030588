"""numpy/torch-CPU restatement of inference.Separator.  TEST INFRASTRUCTURE.

Restates ``/root/reference/inference.py:16-102`` and
``lib/dataset.py:198-205`` (make_padding).  Pinned against the reference's
own ``Separator`` in ``tests/test_oracle_vs_reference.py``.
"""
import numpy as np
import torch

from . import cascaded_net


def make_padding(width, cropsize, offset):
    """dataset.make_padding, lib/dataset.py:198-205 (adds a full roi when width % roi == 0)."""
    left = offset
    roi_size = cropsize - offset * 2
    if roi_size == 0:
        roi_size = cropsize
    right = roi_size - (width % roi_size) + left
    return left, right, roi_size


def _separate(X_spec_pad, roi_size, sd, n_fft, batchsize, cropsize, offset):
    """Separator._separate, inference.py:42-68: overlapping crops -> predict_mask -> stitch."""
    patches = (X_spec_pad.shape[2] - 2 * offset) // roi_size
    crops = np.asarray([X_spec_pad[:, :, i * roi_size:i * roi_size + cropsize] for i in range(patches)])
    out = []
    with torch.no_grad():
        for i in range(0, patches, batchsize):
            xb = torch.from_numpy(crops[i:i + batchsize])
            m = cascaded_net.predict_mask(torch.abs(xb), sd, n_fft, offset).numpy()
            out.append(np.concatenate(m, axis=2))
    return np.concatenate(out, axis=2)


def postprocess(X_spec, mask):
    """Separator._postprocess, inference.py:26-40 with postprocess=False."""
    X_mag = np.abs(X_spec)
    X_phase = np.angle(X_spec)
    y_spec = mask * X_mag * np.exp(1.j * X_phase)
    v_spec = (1 - mask) * X_mag * np.exp(1.j * X_phase)
    return y_spec, v_spec


def merge_artifacts(y_mask, thres=0.05, min_range=64, fade_size=32):
    """spec_utils.merge_artifacts, lib/spec_utils.py:60-93 (--postprocess), restated.

    Frames whose mask minimum over (channel, bin) stays above `thres` for more than `min_range`
    consecutive frames are blended towards 1 (y += w * (1 - y)) with linear fades of `fade_size`.
    numpy slice semantics (negative starts, broadcast errors) and the IndexError on an empty
    above-threshold set are the reference's own behaviour and are kept.
    """
    if min_range < fade_size * 2:
        raise ValueError('min_range must be >= fade_size * 2')
    T = y_mask.shape[2]
    above = np.flatnonzero(y_mask.min(axis=(0, 1)) > thres)
    first = above[0]                                   # IndexError when nothing is above the threshold
    breaks = np.flatnonzero(np.diff(above) != 1)
    starts = np.concatenate([[first], above[breaks + 1]])
    ends = np.concatenate([above[breaks], [above[-1]]])
    w = np.zeros(T, dtype=y_mask.dtype)
    prev_end = None
    for s, e in zip(starts, ends):
        if not (e - s > min_range):
            continue
        s, e = int(s), int(e)
        if prev_end is not None and s - prev_end < fade_size:
            s = prev_end - fade_size * 2
        if s != 0:
            w[s:s + fade_size] = np.linspace(0, 1, fade_size)
        else:
            s -= fade_size
        if e != T:
            w[e - fade_size:e] = np.linspace(1, 0, fade_size)
        else:
            e += fade_size
        w[s + fade_size:e - fade_size] = 1
        prev_end = e
    return y_mask + w[None, None, :] * (1 - y_mask)


def separate_mask(X_spec, sd, n_fft=2048, batchsize=4, cropsize=256, offset=64):
    """Mask half of Separator.separate, inference.py:70-77."""
    n_frame = X_spec.shape[2]
    pad_l, pad_r, roi = make_padding(n_frame, cropsize, offset)
    X_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
    X_pad = X_pad / np.abs(X_spec).max()
    mask = _separate(X_pad, roi, sd, n_fft, batchsize, cropsize, offset)
    return mask[:, :, :n_frame]


def separate_tta_mask(X_spec, sd, n_fft=2048, batchsize=4, cropsize=256, offset=64):
    """Mask half of Separator.separate_tta, inference.py:83-98.

    The normaliser is ``X_spec_pad.max()`` on a COMPLEX array (inference.py:87,94):
    numpy orders complex numbers lexicographically (real, then imag).
    """
    n_frame = X_spec.shape[2]
    pad_l, pad_r, roi = make_padding(n_frame, cropsize, offset)
    X_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
    X_pad = X_pad / X_pad.max()
    mask = _separate(X_pad, roi, sd, n_fft, batchsize, cropsize, offset)
    pad_l += roi // 2
    pad_r += roi // 2
    X_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
    X_pad = X_pad / X_pad.max()
    mask_tta = _separate(X_pad, roi, sd, n_fft, batchsize, cropsize, offset)
    mask_tta = mask_tta[:, :, roi // 2:]
    return (mask[:, :, :n_frame] + mask_tta[:, :, :n_frame]) * 0.5


def separate(X_spec, sd, tta=False, post=False, **kw):
    """Separator.separate / separate_tta -> (y_spec, v_spec), inference.py:70-102
    (post=True: Separator(postprocess=True), inference.py:27-30)."""
    mask = (separate_tta_mask if tta else separate_mask)(X_spec, sd, **kw)
    if post:
        mask = merge_artifacts(np.abs(mask))
    return postprocess(X_spec, mask)


def synth_wave(seconds=30.0, sr=44100, seed=0):
    """Seeded synthetic stereo audio: noise + three sines."""
    rng = np.random.default_rng(seed)
    L = int(round(seconds * sr))
    t = np.arange(L, dtype=np.float64) / sr
    wave = 0.1 * rng.standard_normal((2, L))
    for f in (220.0, 440.0, 3520.0):
        ph = rng.uniform(0, 2 * np.pi, size=(2, 1))
        wave += 0.2 * np.sin(2 * np.pi * f * t[None, :] + ph)
    return wave.astype(np.float32)
