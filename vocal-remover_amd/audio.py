"""Audio front / back end of the reference's scripts without librosa / soundfile.

    load(path, sr, mono, dtype, res_type)   <- librosa.load(...)        inference.py:136-138, lib/spec_utils.py:139-142
    write(path, data, sr)                   <- soundfile.write(...)     inference.py:173,178
    trim(y, top_db)                         <- librosa.effects.trim(y)  lib/spec_utils.py:97-98

Decoding covers RIFF/WAVE (PCM 8/16/24/32 bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE); the reference also accepts
.m4a/.mp3/.mp4/.flac through audioread/ffmpeg, which is outside this package: those raise.  Resampling runs on the GPU
(vr_resample: resampy's 'kaiser_fast' band-limited interpolation restated -- resampy is not vendored in the reference,
parity unpinned).  `write` produces 16-bit PCM, soundfile's default subtype for .wav.
"""
import os
import struct

import numpy as np

from . import native

_PCM, _FLOAT, _EXT = 1, 3, 0xFFFE


def read_wav(path):
    """-> (float32 array [channels, samples] in [-1, 1), sample rate); soundfile.read(..., dtype='float32').T"""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError('%s: not a RIFF/WAVE file (only .wav is decoded here; the reference reads other containers '
                         'through audioread/ffmpeg)' % path)
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack('<I', data[pos + 4:pos + 8])[0]
        chunk = data[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            tag, ch, rate, _, align, bits = struct.unpack('<HHIIHH', chunk[:16])
            if tag == _EXT and len(chunk) >= 26:
                tag = struct.unpack('<H', chunk[24:26])[0]
            fmt = (tag, ch, rate, align, bits)
        elif cid == b'data':
            body = chunk
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError('%s: missing fmt or data chunk' % path)
    tag, ch, rate, align, bits = fmt
    n = len(body) // align
    body = body[:n * align]
    if tag == _PCM and bits == 16:
        x = np.frombuffer(body, '<i2').astype(np.float32) / 32768.0
    elif tag == _PCM and bits == 8:
        x = (np.frombuffer(body, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == _PCM and bits == 24:
        b = np.frombuffer(body, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / float(1 << 23)
    elif tag == _PCM and bits == 32:
        x = (np.frombuffer(body, '<i4').astype(np.float64) / float(1 << 31)).astype(np.float32)
    elif tag == _FLOAT and bits == 32:
        x = np.frombuffer(body, '<f4').astype(np.float32)
    elif tag == _FLOAT and bits == 64:
        x = np.frombuffer(body, '<f8').astype(np.float32)
    else:
        raise ValueError('%s: unsupported WAV encoding (format tag %d, %d bits)' % (path, tag, bits))
    return np.ascontiguousarray(x.reshape(n, ch).T), int(rate)


def write(path, data, sr):
    """soundfile.write(path, data [samples, channels] (or [samples]), sr): 16-bit PCM.  libsndfile's default float -> PCM_16
    path (normalisation on, clipping off: f2s_array) scales by 0x7FFF and rounds to nearest; the clip is only a guard."""
    data = np.asarray(data, dtype=np.float32)
    if data.ndim == 1:
        data = data[:, None]
    n, ch = data.shape
    pcm = np.clip(np.rint(data * 32767.0), -32768, 32767).astype('<i2')
    body = pcm.tobytes()
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<IHHIIHH', 16, _PCM, ch, int(sr), int(sr) * ch * 2, ch * 2, 16))
        f.write(b'data' + struct.pack('<I', len(body)) + body)


def _device():
    return int(os.environ.get('VR_DEVICE', os.environ.get('LOCAL_RANK', '0')))


def resample(y, orig_sr, target_sr, res_type='kaiser_fast'):
    """librosa.resample(y, orig_sr=..., target_sr=..., res_type='kaiser_fast') along the last axis, on the GPU."""
    if res_type != 'kaiser_fast':
        raise NotImplementedError("only res_type='kaiser_fast' (every call site of the reference) is implemented")
    y = np.asarray(y, dtype=np.float32)
    if orig_sr == target_sr:
        return y
    mono = y.ndim == 1
    x = np.ascontiguousarray(y[None] if mono else y)
    n_out = int(np.ceil(x.shape[-1] * float(target_sr) / orig_sr))
    out = np.empty((x.shape[0], n_out), dtype=np.float32)
    native.check(native.lib().vr_resample(_device(), native.np_ptr(x), x.shape[0], x.shape[1], int(orig_sr), int(target_sr),
                                          native.np_ptr(out), n_out))
    return out[0] if mono else out


def load(path, sr=22050, mono=True, dtype=np.float32, res_type='kaiser_fast'):
    """librosa.load: decode to float32, optional down-mix, resample to `sr` (None keeps the file's rate)."""
    y, sr_native = read_wav(path)
    if mono:
        y = y.mean(axis=0)
    elif y.shape[0] == 1:
        y = y[0]                                  # librosa returns 1-D for mono files even with mono=False
    if sr is not None and sr != sr_native:
        y = resample(y, sr_native, sr, res_type=res_type)
    else:
        sr = sr_native
    return np.ascontiguousarray(y.astype(dtype)), sr


def trim(y, top_db=60, frame_length=2048, hop_length=512):
    """librosa.effects.trim (0.10): frames whose RMS is within top_db of the loudest one are signal; a frame counts if
    ANY channel is non-silent.  Returns (y[..., start:end], (start, end)).  O(L) host work on 2 x L floats."""
    y = np.asarray(y)
    pad = frame_length // 2
    yp = np.pad(y, [(0, 0)] * (y.ndim - 1) + [(pad, pad)], mode='constant')          # feature.rms: center=True, zero padding
    n_frames = 1 + (yp.shape[-1] - frame_length) // hop_length
    sq = yp.astype(np.float64) ** 2
    csum = np.concatenate([np.zeros(sq.shape[:-1] + (1,)), np.cumsum(sq, axis=-1)], axis=-1)
    starts = np.arange(n_frames) * hop_length
    mse = (csum[..., starts + frame_length] - csum[..., starts]) / frame_length
    rms = np.sqrt(np.maximum(mse, 0.0)).astype(np.float32)
    ref = rms.max()
    amin = 1e-5
    db = 20.0 * np.log10(np.maximum(amin, rms)) - 20.0 * np.log10(np.maximum(amin, ref))   # amplitude_to_db(ref=np.max, top_db=None)
    non_silent = db > -top_db
    if non_silent.ndim > 1:
        non_silent = non_silent.reshape(-1, non_silent.shape[-1]).any(axis=0)
    nz = np.flatnonzero(non_silent)
    if nz.size > 0:
        start = int(nz[0] * hop_length)
        end = min(y.shape[-1], int((nz[-1] + 1) * hop_length))
    else:
        start, end = 0, 0
    return y[..., start:end], np.asarray([start, end])
