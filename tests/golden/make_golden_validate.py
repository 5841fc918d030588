"""Generate tests/golden/validate_epoch.npz by running the REFERENCE's train.validate_epoch / train_epoch on CPU.

Build container only (needs /root/reference):  python tests/golden/make_golden_validate.py
Inputs are regenerated in the tests from the same seeds (oracle.train_step.synth_batch, oracle.weights);
a checksum of the weights is stored so RNG drift is detected instead of silently mis-compared.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for name in ('librosa', 'soundfile', 'cv2'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['cv2'].IMREAD_COLOR = 1
sys.path.insert(0, '/root/reference')

from lib import nets as ref_nets            # noqa: E402  reference
import train as ref_train                   # noqa: E402  reference
from oracle import train_step, weights      # noqa: E402

N_FFT, NOUT, NL = 512, 8, 32
VAL_B, VAL_T, VAL_BATCH, VAL_SEED = 5, 160, 2, 21      # 5 samples in batches of 2: a ragged last batch


def main():
    torch.set_num_threads(8)
    out = {}
    sd = weights.make_state_dict(11, n_fft=N_FFT, nout=NOUT, nout_lstm=NL)
    out['wsum'] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values() if v.is_floating_point()))
    ref = ref_nets.CascadedNet(N_FFT, N_FFT // 2, NOUT, NL)
    ref.load_state_dict(sd)
    X, y = train_step.synth_batch(VAL_B, T=VAL_T, n_fft=N_FFT, seed=VAL_SEED)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, y), batch_size=VAL_BATCH, shuffle=False)
    out['val_loss'] = np.float64(ref_train.validate_epoch(dl, ref, torch.device('cpu')))
    # per-batch values (what one vr_validate_step returns)
    ref.eval()
    per = []
    with torch.no_grad():
        for Xb, yb in dl:
            pred = ref.predict(Xb)
            per.append(float(torch.nn.L1Loss()(pred, yb[:, :, :, 64:-64])))
    out['val_batch_losses'] = np.array(per, dtype=np.float64)
    # train-mode forward with batch statistics and NO dropout (p=0 modules): model(X) under model.train()
    ref.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    with torch.no_grad():
        mask = ref(X[:2])
    out['train_fwd_mask'] = mask[:, :, ::7].numpy()
    after = ref.state_dict()
    for k in ('stg1_low_band_net.0.enc1.conv.1.running_mean', 'stg3_full_band_net.dec1.conv1.conv.1.running_var',
              'stg2_high_band_net.lstm_dec2.dense.1.running_var'):
        out['train_fwd_after::' + k] = after[k].numpy()
    path = os.path.join(HERE, 'validate_epoch.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', 'val_loss', out['val_loss'])


if __name__ == '__main__':
    main()
