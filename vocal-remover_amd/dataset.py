"""Host side of the training input pipeline (mirror of the reference's lib/dataset.py for the hot path).

`make_padding` (lib/dataset.py:198-205) is shared with inference.  `VocalRemoverTrainingSet` keeps the
reference's constructor and its on-disk format (cached `[T, 2, bins]` complex64 .npy + coef), draws the
numpy random numbers in the reference's order and seek-reads the cropsize rows; everything numeric
(/coef, aggressively_remove_vocal, channel swap, inst-only, mixup, abs, transpose) runs on the GPU as one
kernel per batch (csrc/augment.hip through vr_augment_batch).  There is no CPU fallback.

Drop-in at train.py:235-247: build the set with the same arguments plus the model, then iterate a
`DeviceLoader(dataset, batch_size, shuffle=True)` instead of torch.utils.data.DataLoader -- it yields
(X_batch, y_batch) already on the model's device, which is what train_epoch consumes (train.py:71-79).
"""
import ctypes
import os
import random

import numpy as np
import torch

from . import native


def make_padding(width, cropsize, offset):
    left = offset
    roi_size = cropsize - offset * 2
    if roi_size == 0:
        roi_size = cropsize
    right = roi_size - (width % roi_size) + left
    return left, right, roi_size


class _Aug(ctypes.Structure):               # include/vr_mi355.h: vr_aug
    _fields_ = [('coef', ctypes.c_float), ('coef_mix', ctypes.c_float), ('lam', ctypes.c_float), ('flags', ctypes.c_int)]


def _npy_header(path):
    with open(path, 'rb') as f:
        np.lib.format.read_magic(f)
        shape, fortran, dtype = np.lib.format.read_array_header_1_0(f)
        if fortran:
            raise AssertionError('Fortran order arrays are not supported')       # lib/dataset.py:38
        return shape, dtype, f.tell()


def _read_rows(path, start_row, n_rows, out):
    """Rows [start_row, start_row+n_rows) of the cached spectrogram straight into `out` ([T, 2, bins] complex64)."""
    shape, dtype, data_off = _npy_header(path)
    if dtype != np.complex64 or tuple(shape[1:]) != tuple(out.shape[1:]):
        raise ValueError('%s: expected complex64 rows of shape %s, found %s %s' % (path, out.shape[1:], dtype, shape[1:]))
    row_bytes = int(np.prod(shape[1:])) * dtype.itemsize
    with open(path, 'rb') as f:
        f.seek(data_off + start_row * row_bytes)
        got = f.readinto(out.reshape(-1).view(np.uint8))
    if got != n_rows * row_bytes:
        raise ValueError('%s: short read (%d of %d bytes)' % (path, got, n_rows * row_bytes))


class VocalRemoverTrainingSet(object):
    """lib/dataset.py:15-120 with the numeric part on the GPU.  `model` is the vocal_remover_amd CascadedNet
    whose device (and HIP stream) the batches are produced on."""

    def __init__(self, training_set, cropsize, reduction_rate, reduction_weight, mixup_rate, mixup_alpha, model=None):
        self.training_set = training_set
        self.cropsize = cropsize
        self.reduction_rate = reduction_rate
        self.reduction_weight = None if reduction_weight is None else \
            np.ascontiguousarray(np.asarray(reduction_weight, np.float32).reshape(-1))
        self.mixup_rate = mixup_rate
        self.mixup_alpha = mixup_alpha
        self.model = model

    def __len__(self):
        return len(self.training_set)

    def read_npy_shape(self, path):
        return _npy_header(path)[0]

    # ---- random decisions, in the reference's order (lib/dataset.py:58-120) -------------------------
    def _draw_aug(self):
        flags = 0
        if np.random.uniform() < self.reduction_rate:
            flags |= 1
        if np.random.uniform() < 0.5:
            flags |= 2
        if np.random.uniform() < 0.01:
            flags |= 4
        return flags

    def plan(self, idx):
        X_path, y_path, coef = self.training_set[idx]
        n_rows = self.read_npy_shape(X_path)[0]
        p = {'paths': (X_path, y_path), 'coef': float(coef), 'start': int(np.random.randint(0, n_rows - self.cropsize))}
        p['flags'] = self._draw_aug()
        p['mix'] = None
        if np.random.uniform() < self.mixup_rate:
            j = int(np.random.randint(0, len(self)))
            Xj, yj, coef_j = self.training_set[j]
            nj = self.read_npy_shape(Xj)[0]
            m = {'paths': (Xj, yj), 'coef': float(coef_j), 'start': int(np.random.randint(0, nj - self.cropsize))}
            m['flags'] = self._draw_aug()
            m['lam'] = float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
            p['mix'] = m
        return p

    # ---- batch assembly + the device kernel ---------------------------------------------------------------
    def batch(self, indices):
        if self.model is None:
            raise RuntimeError('VocalRemoverTrainingSet needs the model (its device runs the augmentation); no CPU fallback')
        h = self.model._need_handle()
        plans = [self.plan(i) for i in indices]
        B, T = len(plans), self.cropsize
        bins = int(self.read_npy_shape(plans[0]['paths'][0])[2])
        X = np.empty((B, T, 2, bins), np.complex64)
        y = np.empty((B, T, 2, bins), np.complex64)
        any_mix = any(p['mix'] is not None for p in plans)
        Xm = np.zeros((B, T, 2, bins), np.complex64) if any_mix else None
        ym = np.zeros((B, T, 2, bins), np.complex64) if any_mix else None
        desc = (_Aug * B)()
        need_rw = False
        for b, p in enumerate(plans):
            _read_rows(p['paths'][0], p['start'], T, X[b])
            _read_rows(p['paths'][1], p['start'], T, y[b])
            flags, coef_mix, lam = p['flags'], 1.0, 1.0
            if p['mix'] is not None:
                m = p['mix']
                _read_rows(m['paths'][0], m['start'], T, Xm[b])
                _read_rows(m['paths'][1], m['start'], T, ym[b])
                flags |= 8 | (m['flags'] << 4)
                coef_mix, lam = m['coef'], m['lam']
            need_rw = need_rw or bool(flags & (1 | 16))
            desc[b] = _Aug(p['coef'], coef_mix, lam, flags)
        if need_rw and self.reduction_weight is None:
            raise ValueError('reduction_rate > 0 needs reduction_weight (train.py:197-205)')
        dev = torch.device('cuda', h.device)
        X_mag = torch.empty((B, 2, bins, T), dtype=torch.float32, device=dev)
        y_mag = torch.empty((B, 2, bins, T), dtype=torch.float32, device=dev)
        rw = self.reduction_weight
        native.check(native.lib().vr_augment_batch(
            h.h, native.np_ptr(X), native.np_ptr(y), native.np_ptr(Xm) if any_mix else None,
            native.np_ptr(ym) if any_mix else None, ctypes.cast(desc, ctypes.c_void_p),
            native.np_ptr(rw) if rw is not None else None, B, T, bins, 0,
            ctypes.c_void_p(X_mag.data_ptr()), ctypes.c_void_p(y_mag.data_ptr()), 1))
        return X_mag, y_mag

    def __getitem__(self, idx):
        X_mag, y_mag = self.batch([idx])
        return X_mag[0], y_mag[0]


class VocalRemoverValidationSet(object):
    """lib/dataset.py:123-140: validation patches saved by make_validation_set as .npz with complex X, y of shape
    [2, bins, cropsize]; `abs` runs on the model's GPU (vr_augment_batch with no augmentation flags)."""

    def __init__(self, patch_list, model=None):
        self.patch_list = patch_list
        self.model = model

    def __len__(self):
        return len(self.patch_list)

    def batch(self, indices):
        if self.model is None:
            raise RuntimeError('VocalRemoverValidationSet needs the model (its device computes the magnitudes); no CPU fallback')
        h = self.model._need_handle()
        Xs, ys = [], []
        for i in indices:
            data = np.load(self.patch_list[i])
            Xs.append(np.ascontiguousarray(data['X'].astype(np.complex64, copy=False).transpose(2, 0, 1)))   # -> [T, 2, bins]
            ys.append(np.ascontiguousarray(data['y'].astype(np.complex64, copy=False).transpose(2, 0, 1)))
        X, y = np.stack(Xs), np.stack(ys)
        B, T, _, bins = X.shape
        desc = (_Aug * B)(*[_Aug(1.0, 1.0, 1.0, 0) for _ in range(B)])
        dev = torch.device('cuda', h.device)
        X_mag = torch.empty((B, 2, bins, T), dtype=torch.float32, device=dev)
        y_mag = torch.empty((B, 2, bins, T), dtype=torch.float32, device=dev)
        native.check(native.lib().vr_augment_batch(h.h, native.np_ptr(X), native.np_ptr(y), None, None,
                                                   ctypes.cast(desc, ctypes.c_void_p), None, B, T, bins, 0,
                                                   ctypes.c_void_p(X_mag.data_ptr()), ctypes.c_void_p(y_mag.data_ptr()), 1))
        return X_mag, y_mag

    def __getitem__(self, idx):
        X_mag, y_mag = self.batch([idx])
        return X_mag[0], y_mag[0]


class DeviceLoader(object):
    """Iterable stand-in for torch.utils.data.DataLoader(dataset, batch_size, shuffle) (train.py:242-247):
    yields (X_batch, y_batch) device tensors produced by one vr_augment_batch call per batch."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, generator=None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.drop_last = drop_last
        self.generator = generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                break
            yield self.dataset.batch(idx)


# ---- dataset preparation: the on-disk contract of lib/dataset.py:143-248 ------------------------------------------------
# What has to be identical to the reference is what lands on disk and what the lists contain: songs pair up by sorted file
# name, the spectrogram cache is <audio dir>/sr{}_hl{}_nf{}/<song>.npy ([T, 2, bins] complex64, spec_utils.SpectrogramCache),
# validation patches are cs{}_sr{}_hl{}_nf{}_of{}/<song>_p<j>.npz with keys X and y.  tests/test_cpu_frontend.py holds
# these functions bit-equal to the reference's on a shared cache.
AUDIO_SUFFIXES = frozenset(('.wav', '.m4a', '.mp3', '.mp4', '.flac'))


def _audio_files(folder):
    names = (n for n in os.listdir(folder) if os.path.splitext(n)[1] in AUDIO_SUFFIXES)
    return sorted(os.path.join(folder, n) for n in names)


def make_pair(mix_dir, inst_dir):
    """Song i of the mixtures folder belongs to song i of the instruments folder, both in sorted order."""
    return list(zip(_audio_files(mix_dir), _audio_files(inst_dir)))


_SUBDIR_LAYOUT = {'train': ('training/mixtures', 'training/instruments'), 'val': ('validation/mixtures', 'validation/instruments')}


def train_val_split(dataset_dir, split_mode, val_rate, val_filelist):
    """(train pairs, validation pairs).  'random': ONE random.shuffle of the pair list (the caller seeds `random`, train.py:172),
    the last int(n * val_rate) pairs validate -- or, with a given validation list, everything not on it trains.
    'subdirs': the training/ and validation/ folders are the split."""
    def under(rel):
        return os.path.join(dataset_dir, rel)

    if split_mode == 'subdirs':
        if val_filelist:
            raise ValueError('`val_filelist` option is not available with `subdirs` mode')
        return make_pair(*map(under, _SUBDIR_LAYOUT['train'])), make_pair(*map(under, _SUBDIR_LAYOUT['val']))
    if split_mode != 'random':
        raise UnboundLocalError('unknown split_mode %r' % (split_mode,))      # (what the reference ends in for any other mode)
    pairs = make_pair(under('mixtures'), under('instruments'))
    random.shuffle(pairs)
    if val_filelist:
        held_out = [list(p) for p in val_filelist]
        return [p for p in pairs if list(p) not in held_out], val_filelist
    n_val = int(len(pairs) * val_rate)
    cut = len(pairs) - n_val if n_val else 0        # (val_rate too small: slicing by -0 leaves the training list empty)
    return pairs[:cut], pairs[cut:]


def _song_peak(X, y):
    return np.max([np.abs(X).max(), np.abs(y).max()])


def make_training_set(filelist, sr, hop_length, n_fft):
    """One [mixture cache path, instrumental cache path, peak magnitude of the pair] row per song (VocalRemoverTrainingSet input)."""
    from .spec_utils import SpectrogramCache
    cache = SpectrogramCache(sr, hop_length, n_fft)
    rows = []
    for mix_path, inst_path in filelist:
        X, y, mix_npy, inst_npy = cache.pair(mix_path, inst_path)
        rows.append([mix_npy, inst_npy, _song_peak(X, y)])
    return rows


def make_validation_set(filelist, cropsize, sr, hop_length, n_fft, offset):
    """Cuts every validation song into ceil(T / roi) windows of `cropsize` frames, `roi` apart, of the peak-normalised,
    make_padding-padded spectrogram pair and stores each once as cs{}_sr{}_hl{}_nf{}_of{}/<song>_p<j>.npz; returns the paths."""
    from .spec_utils import SpectrogramCache
    cache = SpectrogramCache(sr, hop_length, n_fft)
    out_dir = 'cs{}_sr{}_hl{}_nf{}_of{}'.format(cropsize, sr, hop_length, n_fft, offset)
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for mix_path, inst_path in filelist:
        song = os.path.splitext(os.path.basename(mix_path))[0]
        X, y, _, _ = cache.pair(mix_path, inst_path)
        peak = _song_peak(X, y)
        frames = X.shape[2]
        left, right, roi = make_padding(frames, cropsize, offset)
        widths = ((0, 0), (0, 0), (left, right))
        padded = {'X': np.pad(X / peak, widths, mode='constant'), 'y': np.pad(y / peak, widths, mode='constant')}
        for j in range(-(-frames // roi)):
            path = os.path.join(out_dir, '{}_p{}.npz'.format(song, j))
            if not os.path.exists(path):
                np.savez(path, **{k: v[:, :, j * roi:j * roi + cropsize] for k, v in padded.items()})
            written.append(path)
    return written
