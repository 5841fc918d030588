"""Generate tests/golden/dataset_pipeline.npz by running the REFERENCE's own lib/dataset.py on CPU.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_dataset.py
Holds three small cached spectrograms (the reference's on-disk format [T, 2, bins] complex64), their coef,
a reduction_weight, and for a range of numpy seeds the (X_mag, y_mag) VocalRemoverTrainingSet.__getitem__
returned -- the anchor for the device training-input pipeline on the GPU box, where the reference is absent.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for name in ('librosa', 'soundfile', 'cv2'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['cv2'].IMREAD_COLOR = 1
sys.path.insert(0, '/root/reference')

from lib import dataset as ref_dataset      # noqa: E402  reference

BINS, CROP, LENGTHS, NSEEDS = 33, 32, (90, 140, 75), 12
PARAMS = dict(reduction_rate=0.5, mixup_rate=0.5, mixup_alpha=0.4)


def main():
    rng = np.random.RandomState(2024)
    out = {}
    tmp = tempfile.mkdtemp()
    ts = []
    for i, T in enumerate(LENGTHS):
        arrs = []
        for tag in ('X', 'y'):
            a = (rng.randn(T, 2, BINS) + 1j * rng.randn(T, 2, BINS)).astype(np.complex64) * (0.3 + i)
            a[rng.rand(T, 2, BINS) < 0.02] = 0
            path = os.path.join(tmp, 'song%d_%s.npy' % (i, tag))
            np.save(path, a)
            out['song%d_%s' % (i, tag)] = a
            arrs.append((path, a))
        coef = np.max([np.abs(arrs[0][1]).max(), np.abs(arrs[1][1]).max()])          # lib/dataset.py:214
        out['coef%d' % i] = np.float32(coef)
        ts.append([arrs[0][0], arrs[1][0], coef])
    u, s = 3, BINS - 4
    rw = np.concatenate([np.linspace(0, 1, u, dtype=np.float32)[:, None], np.linspace(1, 0, s - u, dtype=np.float32)[:, None],
                         np.zeros((BINS - s, 1), dtype=np.float32)], axis=0) * 0.2              # train.py:197-205 shape
    out['reduction_weight'] = rw
    ds = ref_dataset.VocalRemoverTrainingSet(ts * 2, cropsize=CROP, reduction_weight=rw, **PARAMS)
    for seed in range(NSEEDS):
        np.random.seed(seed)
        X_mag, y_mag = ds[seed % len(ds)]
        out['seed%d_X' % seed] = np.asarray(X_mag, np.float32)
        out['seed%d_y' % seed] = np.asarray(y_mag, np.float32)
    np.savez_compressed(os.path.join(HERE, 'dataset_pipeline.npz'), **out)
    print('wrote dataset_pipeline.npz:', sum(v.nbytes for v in out.values()) // 1024, 'KiB uncompressed')


if __name__ == '__main__':
    main()
