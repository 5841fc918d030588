"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference cannot travel to the GPU box, so its outputs on seeded inputs are committed as small
fixtures; weights are regenerated from oracle.weights.make_state_dict(seed) (torch CPU RNG) and a
checksum of them is stored so RNG drift is detected instead of silently mis-compared.
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
import inference as ref_inference           # noqa: E402  reference
from oracle import train_step, weights      # noqa: E402


def weight_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.is_floating_point()))


def main():
    torch.set_num_threads(8)
    out = {}

    # ---- small net (topology identical to the default one): eval forward + separator -------------
    n_fft, nout, nl = 512, 8, 32
    sd = weights.make_state_dict(11, n_fft=n_fft, nout=nout, nout_lstm=nl)
    ref = ref_nets.CascadedNet(n_fft, n_fft // 2, nout, nl)
    ref.load_state_dict(sd)
    ref.eval()
    x = torch.rand(2, 2, n_fft // 2 + 1, 160, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out['small_mask'] = ref.predict_mask(x).numpy()
        out['small_pred'] = ref.predict(x).numpy()
    out['small_wsum'] = np.float64(weight_checksum(sd))
    rng = np.random.default_rng(5)
    T = 300
    X = (rng.standard_normal((2, n_fft // 2 + 1, T)) + 1j * rng.standard_normal((2, n_fft // 2 + 1, T))).astype(np.complex64)
    sp = ref_inference.Separator(ref, torch.device('cpu'), batchsize=2, cropsize=160)
    y, v = sp.separate(X.copy())
    yt, vt = sp.separate_tta(X.copy())
    out['sep_y'] = y[:, ::5].astype(np.complex64)          # every 5th bin keeps the fixture small
    out['sep_v'] = v[:, ::5].astype(np.complex64)
    out['sep_tta_y'] = yt[:, ::5].astype(np.complex64)

    # ---- default net CascadedNet(2048, 1024, 32, 128): one 144-frame crop -------------------------
    sd_full = weights.make_state_dict(1234)
    ref_full = ref_nets.CascadedNet(2048, 1024, 32, 128)
    ref_full.load_state_dict(sd_full)
    ref_full.eval()
    xf = torch.rand(1, 2, 1025, 144, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        out['full_mask'] = ref_full.predict_mask(xf).numpy()
    out['full_wsum'] = np.float64(weight_checksum(sd_full))

    # ---- train step in float64 on the small net (reference modules + torch.optim.Adam) --------------
    ref64 = ref_nets.CascadedNet(n_fft, n_fft // 2, nout, nl).double()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref64.load_state_dict(sd64)
    ref64.train()
    B = 2
    Xb, yb = train_step.synth_batch(B, T=64, n_fft=n_fft, seed=5)
    masks = train_step.dropout_masks(B, seed=9, nout=nout)

    class Inject(torch.nn.Module):
        def __init__(self, keep):
            super().__init__()
            self.keep = keep

        def forward(self, t):
            return t * self.keep[:, :, None, None]

    for name, keep in masks.items():
        ref64.get_submodule(name).dropout = Inject(keep.double())
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, ref64.parameters()), lr=1e-3)
    loss = torch.nn.L1Loss()(ref64(Xb.double()) * Xb.double(), yb.double())
    loss.backward()
    out['train_loss'] = np.float64(loss.item())
    gnames, gnorms = [], []
    for k, p in ref64.named_parameters():
        if p.grad is not None:
            gnames.append(k)
            gnorms.append(float(p.grad.norm()))
    out['train_grad_names'] = np.array(gnames)
    out['train_grad_norms'] = np.array(gnorms)
    for k in ('stg3_full_band_net.dec1.conv1.conv.0.weight', 'stg1_low_band_net.0.enc1.conv.0.weight',
              'stg2_low_band_net.0.lstm_dec2.lstm.weight_hh_l0', 'out.weight'):
        out['train_grad::' + k] = dict(ref64.named_parameters())[k].grad.numpy().astype(np.float64)
    opt.step()
    sd_after = ref64.state_dict()
    for k in ('stg3_full_band_net.dec1.conv1.conv.0.weight', 'stg3_full_band_net.dec1.conv1.conv.1.running_var',
              'stg1_high_band_net.aspp.bottleneck.conv.1.running_mean', 'out.weight'):
        out['train_after::' + k] = sd_after[k].numpy().astype(np.float64)

    path = os.path.join(HERE, 'reference_outputs.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
