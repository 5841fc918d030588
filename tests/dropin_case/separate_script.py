"""A script shaped like the reference's inference.py, run by tests through vocal-remover_amd/run.py (the reference checkout
itself cannot travel to the GPU box).  The import block is inference.py:1-13's; `Separator._separate`, `separate`,
`separate_tta` and `_postprocess` below carry the reference's own call sequence (inference.py:26-102) so that the test drives
the native CascadedNet exactly the way the reference's loop does: complex64 crops -> torch.from_numpy(...).to(device) ->
torch.abs -> model.predict_mask -> .detach().cpu().numpy() -> np.concatenate."""
import argparse

import librosa  # noqa: F401   (module-level imports of the reference: must resolve, real or stand-in)
import numpy as np
import soundfile as sf  # noqa: F401
import torch

from lib import dataset
from lib import nets
from lib import spec_utils
from lib import utils


class Separator(object):

    def __init__(self, model, device=None, batchsize=1, cropsize=256, postprocess=False):
        self.model = model
        self.offset = model.offset
        self.device = device
        self.batchsize = batchsize
        self.cropsize = cropsize
        self.postprocess = postprocess

    def _postprocess(self, X_spec, mask):
        if self.postprocess:
            mask_mag = np.abs(mask)
            mask_mag = spec_utils.merge_artifacts(mask_mag)
            mask = mask_mag * np.exp(1.j * np.angle(mask))
        X_mag = np.abs(X_spec)
        X_phase = np.angle(X_spec)
        y_spec = mask * X_mag * np.exp(1.j * X_phase)
        v_spec = (1 - mask) * X_mag * np.exp(1.j * X_phase)
        return y_spec, v_spec

    def _separate(self, X_spec_pad, roi_size):
        X_dataset = []
        patches = (X_spec_pad.shape[2] - 2 * self.offset) // roi_size
        for i in range(patches):
            start = i * roi_size
            X_spec_crop = X_spec_pad[:, :, start:start + self.cropsize]
            X_dataset.append(X_spec_crop)
        X_dataset = np.asarray(X_dataset)
        self.model.eval()
        with torch.no_grad():
            mask_list = []
            for i in range(0, patches, self.batchsize):
                X_batch = X_dataset[i: i + self.batchsize]
                X_batch = torch.from_numpy(X_batch).to(self.device)
                mask = self.model.predict_mask(torch.abs(X_batch))
                mask = mask.detach().cpu().numpy()
                mask = np.concatenate(mask, axis=2)
                mask_list.append(mask)
            mask = np.concatenate(mask_list, axis=2)
        return mask

    def separate(self, X_spec):
        n_frame = X_spec.shape[2]
        pad_l, pad_r, roi_size = dataset.make_padding(n_frame, self.cropsize, self.offset)
        X_spec_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
        X_spec_pad /= np.abs(X_spec).max()
        mask = self._separate(X_spec_pad, roi_size)
        mask = mask[:, :, :n_frame]
        return self._postprocess(X_spec, mask)

    def separate_tta(self, X_spec):
        n_frame = X_spec.shape[2]
        pad_l, pad_r, roi_size = dataset.make_padding(n_frame, self.cropsize, self.offset)
        X_spec_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
        X_spec_pad /= X_spec_pad.max()
        mask = self._separate(X_spec_pad, roi_size)
        pad_l += roi_size // 2
        pad_r += roi_size // 2
        X_spec_pad = np.pad(X_spec, ((0, 0), (0, 0), (pad_l, pad_r)), mode='constant')
        X_spec_pad /= X_spec_pad.max()
        mask_tta = self._separate(X_spec_pad, roi_size)
        mask_tta = mask_tta[:, :, roi_size // 2:]
        mask = (mask[:, :, :n_frame] + mask_tta[:, :, :n_frame]) * 0.5
        return self._postprocess(X_spec, mask)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--gpu', '-g', type=int, default=-1)
    p.add_argument('--pretrained_model', '-P', type=str, required=True)
    p.add_argument('--input', '-i', required=True)            # .npy spectrogram [2, bins, T] complex64, or .wav
    p.add_argument('--n_fft', '-f', type=int, default=2048)
    p.add_argument('--hop_length', '-H', type=int, default=1024)
    p.add_argument('--nout', type=int, default=32)
    p.add_argument('--nout_lstm', type=int, default=128)
    p.add_argument('--batchsize', '-B', type=int, default=4)
    p.add_argument('--cropsize', '-c', type=int, default=256)
    p.add_argument('--tta', '-t', action='store_true')
    p.add_argument('--postprocess', '-p', action='store_true')
    p.add_argument('--output', '-o', type=str, required=True)
    p.add_argument('--probe', action='store_true')
    args = p.parse_args()

    device = torch.device('cpu')
    if args.gpu >= 0 and torch.cuda.is_available():
        device = torch.device('cuda:{}'.format(args.gpu))
    model = nets.CascadedNet(args.n_fft, args.hop_length, args.nout, args.nout_lstm)
    model.load_state_dict(torch.load(args.pretrained_model, map_location='cpu'))
    model.to(device)
    if args.probe:                         # CPU tests stop here: which modules did the imports above resolve to?
        np.savez(args.output, nets=nets.__file__, model_class=type(model).__module__, utils=utils.ORIGIN,
                 spec_utils=spec_utils.__file__, dataset=dataset.__file__, image=spec_utils.spectrogram_to_image(None),
                 librosa=getattr(librosa, '__file__', 'stand-in'))
        return
    if args.input.endswith('.wav'):
        X, sr = librosa.load(args.input, sr=44100, mono=False, dtype=np.float32, res_type='kaiser_fast')
        X_spec = spec_utils.wave_to_spectrogram(X, args.hop_length, args.n_fft)
    else:
        X_spec = np.load(args.input)
    sp = Separator(model=model, device=device, batchsize=args.batchsize, cropsize=args.cropsize, postprocess=args.postprocess)
    y_spec, v_spec = sp.separate_tta(X_spec) if args.tta else sp.separate(X_spec)
    out = dict(y_spec=y_spec, v_spec=v_spec)
    if args.input.endswith('.wav'):
        wave = spec_utils.spectrogram_to_wave(y_spec, hop_length=args.hop_length)
        sf.write(args.output + '_Instruments.wav', wave.T, sr)
        out['y_wave'] = wave
    np.savez(args.output, **out)


if __name__ == '__main__':
    main()
