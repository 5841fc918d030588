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


# ---- dataset preparation (lib/dataset.py:143-248): same file lists, cache layout and patch files as the reference ----
def make_pair(mix_dir, inst_dir):
    """lib/dataset.py:143-159."""
    input_exts = ['.wav', '.m4a', '.mp3', '.mp4', '.flac']
    X_list = sorted([os.path.join(mix_dir, fname) for fname in os.listdir(mix_dir) if os.path.splitext(fname)[1] in input_exts])
    y_list = sorted([os.path.join(inst_dir, fname) for fname in os.listdir(inst_dir) if os.path.splitext(fname)[1] in input_exts])
    return list(zip(X_list, y_list))


def train_val_split(dataset_dir, split_mode, val_rate, val_filelist):
    """lib/dataset.py:162-195 (uses the `random` module's stream exactly like the reference)."""
    if split_mode == 'random':
        filelist = make_pair(os.path.join(dataset_dir, 'mixtures'), os.path.join(dataset_dir, 'instruments'))
        random.shuffle(filelist)
        if len(val_filelist) == 0:
            val_size = int(len(filelist) * val_rate)
            train_filelist = filelist[:-val_size]
            val_filelist = filelist[-val_size:]
        else:
            train_filelist = [pair for pair in filelist if list(pair) not in val_filelist]
    elif split_mode == 'subdirs':
        if len(val_filelist) != 0:
            raise ValueError('`val_filelist` option is not available with `subdirs` mode')
        train_filelist = make_pair(os.path.join(dataset_dir, 'training/mixtures'), os.path.join(dataset_dir, 'training/instruments'))
        val_filelist = make_pair(os.path.join(dataset_dir, 'validation/mixtures'), os.path.join(dataset_dir, 'validation/instruments'))
    return train_filelist, val_filelist


def make_training_set(filelist, sr, hop_length, n_fft):
    """lib/dataset.py:208-217: [[X_cache_path, y_cache_path, coef], ...]."""
    from . import spec_utils
    ret = []
    for X_path, y_path in filelist:
        X, y, X_cache_path, y_cache_path = spec_utils.cache_or_load(X_path, y_path, sr, hop_length, n_fft)
        coef = np.max([np.abs(X).max(), np.abs(y).max()])
        ret.append([X_cache_path, y_cache_path, coef])
    return ret


def make_validation_set(filelist, cropsize, sr, hop_length, n_fft, offset):
    """lib/dataset.py:220-248: normalised, padded, overlapping cropsize-frame patches as .npz (keys X, y) in the
    reference's directory `cs{}_sr{}_hl{}_nf{}_of{}` with the reference's file names."""
    from . import spec_utils
    patch_list = []
    patch_dir = 'cs{}_sr{}_hl{}_nf{}_of{}'.format(cropsize, sr, hop_length, n_fft, offset)
    os.makedirs(patch_dir, exist_ok=True)
    for X_path, y_path in filelist:
        basename = os.path.splitext(os.path.basename(X_path))[0]
        X, y, _, _ = spec_utils.cache_or_load(X_path, y_path, sr, hop_length, n_fft)
        coef = np.max([np.abs(X).max(), np.abs(y).max()])
        X, y = X / coef, y / coef
        l, r, roi_size = make_padding(X.shape[2], cropsize, offset)
        X_pad = np.pad(X, ((0, 0), (0, 0), (l, r)), mode='constant')
        y_pad = np.pad(y, ((0, 0), (0, 0), (l, r)), mode='constant')
        len_dataset = int(np.ceil(X.shape[2] / roi_size))
        for j in range(len_dataset):
            outpath = os.path.join(patch_dir, '{}_p{}.npz'.format(basename, j))
            start = j * roi_size
            if not os.path.exists(outpath):
                np.savez(outpath, X=X_pad[:, :, start:start + cropsize], y=y_pad[:, :, start:start + cropsize])
            patch_list.append(outpath)
    return patch_list
