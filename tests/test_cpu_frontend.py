"""CPU tests of the host-side front end: WAV codec, librosa.effects.trim restatement,
dataset preparation pinned to the REFERENCE's own lib/dataset.py where it is importable (build container)."""
import os
import random
import struct

import numpy as np
import pytest

from oracle import audio_np


def test_wav_codec_round_trip_and_encodings(vr, tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1000, 2)) * 0.3).astype(np.float32)
    x[0] = [1.5, -1.5]                                       # clipped like libsndfile does
    p = str(tmp_path / 'a.wav')
    vr.audio.write(p, x, 44100)
    y, sr = vr.audio.read_wav(p)
    assert sr == 44100 and y.shape == (2, 1000)
    # written as rint(x * 0x7FFF) (libsndfile's default float -> PCM_16 scale), read back as q / 0x8000
    assert np.array_equal(y.T, np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.float32) / 32768)
    # scipy writes the other encodings; the reader must agree with scipy's reader
    import scipy.io.wavfile as wf
    for dt, tol in ((np.int32, 1e-9), (np.float32, 0.0), (np.uint8, 1e-9), (np.float64, 1e-7)):
        q = str(tmp_path / ('b_%s.wav' % np.dtype(dt).name))
        if dt == np.int32:
            d = (x * (2 ** 31 - 1)).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
        elif dt == np.uint8:
            d = ((x.clip(-1, 1) * 127) + 128).astype(np.uint8)
        else:
            d = x.astype(dt)
        wf.write(q, 22050, d)
        got, sr2 = vr.audio.read_wav(q)
        rate, ref = wf.read(q)
        ref = ref.astype(np.float64)
        if dt == np.int32:
            ref = ref / 2 ** 31
        elif dt == np.uint8:
            ref = (ref - 128) / 128
        assert sr2 == rate == 22050 and np.abs(got.T - ref).max() <= tol + 1e-7
    # 24-bit PCM by hand
    v = np.array([[0, 1], [-1, 8388607], [-8388608, 5]], np.int32)
    body = b''.join(struct.pack('<i', int(s))[:3] for s in v.reshape(-1))
    r = str(tmp_path / 'c.wav')
    with open(r, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 2, 8000, 48000, 6, 24)
                + b'data' + struct.pack('<I', len(body)) + body)
    got, sr3 = vr.audio.read_wav(r)
    assert sr3 == 8000 and np.abs(got.T - v / 8388608.0).max() < 1e-9
    with pytest.raises(ValueError):
        open(str(tmp_path / 'd.mp3'), 'wb').write(b'ID3')
        vr.audio.read_wav(str(tmp_path / 'd.mp3'))


def test_trim_removes_leading_and_trailing_silence(vr):
    sr = 8000
    rng = np.random.default_rng(1)
    sig = (rng.standard_normal((2, 3 * sr)) * 0.2).astype(np.float32)
    y = np.concatenate([np.zeros((2, 1700), np.float32), sig, 1e-6 * np.ones((2, 2500), np.float32)], axis=1)
    t, (s, e) = vr.audio.trim(y)
    assert s <= 1700 and 1700 - s <= 2048 and e >= 1700 + 3 * sr and e - (1700 + 3 * sr) <= 2048
    assert t.shape[1] == e - s and np.array_equal(t, y[:, s:e])
    t0, (s0, e0) = vr.audio.trim(np.zeros((2, 5000), np.float32))       # all silent: db == 0 everywhere > -60 -> nothing trimmed
    assert (s0, e0) == (0, 5000)


def test_resample_restatement_is_a_unit_gain_low_pass():
    """Pins oracle/audio_np.py itself (resampy is absent): DC gain 1, pass-band tone kept, stop band attenuated."""
    sr_in, sr_out, n = 48000, 44100, 6000
    t = np.arange(n) / sr_in
    core = slice(100, int(n * sr_out / sr_in) - 100)
    dc = audio_np.resample_kaiser_fast(np.ones(n, np.float32), sr_in, sr_out)
    assert len(dc) == int(np.ceil(n * sr_out / sr_in)) and np.abs(dc[core] - 1).max() < 1e-3
    tone = audio_np.resample_kaiser_fast(np.sin(2 * np.pi * 1000 * t).astype(np.float32), sr_in, sr_out)
    ref = np.sin(2 * np.pi * 1000 * np.arange(len(tone)) / sr_out)
    assert np.abs(tone[core] - ref[core]).max() < 2e-3
    hi = audio_np.resample_kaiser_fast(np.sin(2 * np.pi * 23500 * t).astype(np.float32), sr_in, sr_out)
    assert np.abs(hi[core]).max() < 0.05


def _fake_cache(tmp_path, sr, hop, n_fft, names=('s0', 's1', 's2'), lengths=(300, 437, 512)):
    """Song pairs whose spectrogram cache already exists: cache_or_load then never decodes audio (no librosa needed)."""
    rng = np.random.RandomState(4)
    bins = n_fft // 2 + 1
    cache = 'sr{}_hl{}_nf{}'.format(sr, hop, n_fft)
    for sub in ('mixtures', 'instruments'):
        os.makedirs(str(tmp_path / 'data' / sub / cache), exist_ok=True)
    for name, T in zip(names, lengths):
        y = (rng.randn(T, 2, bins) + 1j * rng.randn(T, 2, bins)).astype(np.complex64)
        X = (y + 0.3 * (rng.randn(T, 2, bins) + 1j * rng.randn(T, 2, bins))).astype(np.complex64)
        for sub, arr in (('mixtures', X), ('instruments', y)):
            open(str(tmp_path / 'data' / sub / (name + '.wav')), 'wb').write(b'')          # the pair list is built from file names
            np.save(str(tmp_path / 'data' / sub / cache / (name + '.npy')), arr)
    return str(tmp_path / 'data')


def test_dataset_preparation_matches_the_reference(vr, reference_lib, tmp_path, monkeypatch):
    """make_pair / train_val_split / make_training_set / make_validation_set against lib/dataset.py on the same cache."""
    import lib.dataset as ref_ds          # reference (librosa stubbed by the fixture; the cache exists, so it is never called)
    sr, hop, n_fft = 44100, 256, 512
    root = _fake_cache(tmp_path, sr, hop, n_fft)
    monkeypatch.chdir(tmp_path)
    assert vr.dataset.make_pair(root + '/mixtures', root + '/instruments') == ref_ds.make_pair(root + '/mixtures', root + '/instruments')
    random.seed(3)
    want_split = ref_ds.train_val_split(root, 'random', 0.34, [])
    random.seed(3)
    got_split = vr.dataset.train_val_split(root, 'random', 0.34, [])
    assert got_split == want_split
    want_ts = ref_ds.make_training_set(want_split[0], sr, hop, n_fft)
    got_ts = vr.dataset.make_training_set(got_split[0], sr, hop, n_fft)
    assert [(a, b) for a, b, _ in got_ts] == [(a, b) for a, b, _ in want_ts]
    assert all(abs(float(g[2]) - float(w[2])) == 0.0 for g, w in zip(got_ts, want_ts))
    os.makedirs('ref'); os.makedirs('got')
    monkeypatch.chdir(tmp_path / 'ref')
    want_p = ref_ds.make_validation_set(want_split[1] + want_split[0][:1], 160, sr, hop, n_fft, 64)
    monkeypatch.chdir(tmp_path / 'got')
    got_p = vr.dataset.make_validation_set(got_split[1] + got_split[0][:1], 160, sr, hop, n_fft, 64)
    assert got_p == want_p and len(got_p) >= 3
    for p in got_p:
        a, b = np.load(str(tmp_path / 'got' / p)), np.load(str(tmp_path / 'ref' / p))
        assert np.array_equal(a['X'], b['X']) and np.array_equal(a['y'], b['y'])
    with pytest.raises(ValueError):
        vr.dataset.train_val_split(root, 'subdirs', 0.2, [['a', 'b']])
