"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's training sample pipeline,
lib/dataset.py:15-120 (VocalRemoverTrainingSet).  Only tests/ may import this module.

Pinned in tests/test_oracle_vs_reference.py::test_training_sample_pipeline_matches_reference against the
reference class itself, on synthetic cached .npy files and a range of numpy seeds (same global numpy RNG
stream => identical random decisions).

The random numbers are drawn from numpy's GLOBAL generator in exactly the reference's order:
  crop start (randint)                                   dataset.py:60
  reduce? swap? inst? (three uniform())                  dataset.py:69,72,77
  mixup? (uniform())                                     dataset.py:113
  if mixup: partner index (randint), its crop start, its three aug draws, lam (beta)   dataset.py:86-97
"""
import numpy as np


def npy_rows(path, first_row, n_rows):
    """Rows [first_row, first_row + n_rows) of a C-ordered .npy without loading the file (dataset.py:33-46)."""
    with open(path, 'rb') as f:
        np.lib.format.read_magic(f)
        shape, fortran, dtype = np.lib.format.read_array_header_1_0(f)
        assert not fortran
        per_row = int(np.prod(shape[1:]))
        f.seek(first_row * per_row * dtype.itemsize, 1)
        flat = np.fromfile(f, count=per_row * n_rows, dtype=dtype)
    return flat.reshape((-1,) + tuple(shape[1:]))


def npy_shape(path):
    with open(path, 'rb') as f:                                  # dataset.py:27-31
        np.lib.format.read_magic(f)
        return np.lib.format.read_array_header_1_0(f)[0]


def remove_vocal(X, y, reduction_weight):
    """dataset.py:48-56: push the instrumental magnitude down where the vocal dominates, keep y's phase."""
    xm, ym = np.abs(X), np.abs(y)
    v = xm - ym
    v = v * (v > ym)
    ym = np.clip(ym - v * reduction_weight, 0, np.inf)
    return ym * np.exp(1.j * np.angle(y))


def _crop(pair, cropsize):
    X_path, y_path = pair
    n = npy_shape(X_path)[0]
    start = np.random.randint(0, n - cropsize)                   # dataset.py:60
    X = npy_rows(X_path, start, cropsize).transpose(1, 2, 0)     # [T,2,bins] -> [2,bins,T]   (dataset.py:63-64)
    y = npy_rows(y_path, start, cropsize).transpose(1, 2, 0)
    return X, y


def _augment(X, y, reduction_rate, reduction_weight):
    if np.random.uniform() < reduction_rate:                     # dataset.py:69-70
        y = remove_vocal(X, y, reduction_weight)
    if np.random.uniform() < 0.5:                                # dataset.py:72-75
        X, y = X[::-1].copy(), y[::-1].copy()
    if np.random.uniform() < 0.01:                               # dataset.py:77-79
        X = y.copy()
    return X, y


def training_sample(training_set, idx, cropsize, reduction_rate, reduction_weight, mixup_rate, mixup_alpha):
    """VocalRemoverTrainingSet.__getitem__ (dataset.py:105-120) -> (X_mag, y_mag), each [2, bins, cropsize]."""
    X_path, y_path, coef = training_set[idx]
    X, y = _crop((X_path, y_path), cropsize)
    X = X / coef
    y = y / coef
    X, y = _augment(X, y, reduction_rate, reduction_weight)
    if np.random.uniform() < mixup_rate:                         # dataset.py:113-114
        j = np.random.randint(0, len(training_set))              # dataset.py:86
        Xj_path, yj_path, coef_j = training_set[j]
        Xj, yj = _crop((Xj_path, yj_path), cropsize)
        Xj, yj = _augment(Xj / coef_j, yj / coef_j, reduction_rate, reduction_weight)
        lam = np.random.beta(mixup_alpha, mixup_alpha)           # dataset.py:97
        X = lam * X + (1 - lam) * Xj
        y = lam * y + (1 - lam) * yj
    return np.abs(X), np.abs(y)
