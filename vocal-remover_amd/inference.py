"""inference.Separator look-alike (inference.py:16-102): same constructor, same methods, same
numpy-in / numpy-out contract; the crop loop, stitching and mask application run on the GPU."""
import numpy as np

from . import native


class Separator(object):

    def __init__(self, model, device=None, batchsize=1, cropsize=256, postprocess=False):
        self.model = model
        self.offset = model.offset
        self.device = device
        self.batchsize = batchsize
        self.cropsize = cropsize
        self.postprocess = postprocess      # spec_utils.merge_artifacts (lib/spec_utils.py:60-93), on device

    def _flags(self, tta):
        return (1 if tta else 0) | (2 if self.postprocess else 0)

    def _run(self, X_spec, tta):
        h = self.model._need_handle()
        X_spec = np.ascontiguousarray(np.asarray(X_spec).astype(np.complex64))
        if X_spec.ndim != 3 or X_spec.shape[0] != 2 or X_spec.shape[1] != self.model.output_bin:
            raise ValueError('X_spec must be [2, %d, T]' % self.model.output_bin)
        self.model.eval()                        # inference.py:52
        y_spec = np.empty_like(X_spec)
        v_spec = np.empty_like(X_spec)
        native.check(native.lib().vr_separate(h.h, native.np_ptr(X_spec), 0, X_spec.shape[2], self._flags(tta),
                                              int(self.batchsize), int(self.cropsize),
                                              native.np_ptr(y_spec), native.np_ptr(v_spec), 0))
        return y_spec, v_spec

    def separate(self, X_spec):
        """inference.py:70-81."""
        return self._run(X_spec, False)

    def separate_tta(self, X_spec):
        """inference.py:83-102 (incl. the complex lexicographic-max normaliser of :87,94)."""
        return self._run(X_spec, True)

    def separate_wave(self, wave, tta=False):
        """Whole inference.py:147-176 pipeline in one device-resident call.

        wave: numpy [2, L] float32 (host) or a torch cuda tensor [2, L]; returns two arrays / tensors
        [2, hop*(L//hop)] (instruments, vocals) on the same side.
        """
        h = self.model._need_handle()
        hop = self.model.hop_length
        self.model.eval()
        try:
            import torch
        except ImportError:              # pragma: no cover
            torch = None
        if torch is not None and torch.is_tensor(wave) and wave.is_cuda:
            wave = wave.detach().to(torch.float32).contiguous()
            L = int(wave.shape[1])
            out_len = hop * (L // hop)
            y = torch.empty((2, out_len), dtype=torch.float32, device=wave.device)
            v = torch.empty_like(y)
            torch.cuda.current_stream(wave.device).synchronize()
            native.check(native.lib().vr_separate_wave(h.h, wave.data_ptr(), 1, L, self._flags(tta), int(self.batchsize),
                                                       int(self.cropsize), y.data_ptr(), v.data_ptr(), 1))
            return y, v
        wave = np.ascontiguousarray(np.asarray(wave, dtype=np.float32))
        L = wave.shape[1]
        out_len = hop * (L // hop)
        y = np.empty((2, out_len), dtype=np.float32)
        v = np.empty_like(y)
        native.check(native.lib().vr_separate_wave(h.h, native.np_ptr(wave), 0, L, self._flags(tta), int(self.batchsize),
                                                   int(self.cropsize), native.np_ptr(y), native.np_ptr(v), 0))
        return y, v


def main(argv=None):
    """inference.py main() (inference.py:107-185) with the same flags; decoding / resampling / WAV writing come from
    vocal_remover_amd.audio (no librosa / soundfile), everything numeric from the library.  --output_image is the only
    flag not carried over (cv2 image dump, SURVEY section 2: out of scope)."""
    import argparse
    import os

    import torch

    from . import audio, nets
    p = argparse.ArgumentParser()
    p.add_argument('--gpu', '-g', type=int, default=0)
    p.add_argument('--pretrained_model', '-P', type=str, required=True)
    p.add_argument('--input', '-i', required=True)
    p.add_argument('--sr', '-r', type=int, default=44100)
    p.add_argument('--n_fft', '-f', type=int, default=2048)
    p.add_argument('--hop_length', '-H', type=int, default=1024)
    p.add_argument('--batchsize', '-B', type=int, default=4)
    p.add_argument('--cropsize', '-c', type=int, default=256)
    p.add_argument('--tta', '-t', action='store_true')
    p.add_argument('--postprocess', '-p', action='store_true')
    p.add_argument('--output_image', '-I', action='store_true')      # accepted for command-line compatibility; no image is written
    p.add_argument('--output_dir', '-o', type=str, default="")
    args = p.parse_args(argv)

    if args.output_image:
        print('--output_image: not written by this entry point (spectrogram_to_image + cv2 are outside the MI355X path); run the '
              "reference's inference.py through vocal-remover_amd/run.py to get the image dumps")
    device = torch.device('cuda:{}'.format(max(args.gpu, 0)))
    model = nets.CascadedNet(args.n_fft, args.hop_length, 32, 128)
    model.load_state_dict(torch.load(args.pretrained_model, map_location='cpu'))
    model.to(device)
    X, sr = audio.load(args.input, sr=args.sr, mono=False, dtype=np.float32, res_type='kaiser_fast')
    basename = os.path.splitext(os.path.basename(args.input))[0]
    if X.ndim == 1:
        X = np.asarray([X, X])                   # mono to stereo (inference.py:143-145)
    sp = Separator(model=model, device=device, batchsize=args.batchsize, cropsize=args.cropsize, postprocess=args.postprocess)
    y_wave, v_wave = sp.separate_wave(X, tta=args.tta)      # STFT -> separate -> iSTFT x2 in one device-resident call
    output_dir = args.output_dir
    if output_dir != "":
        output_dir = output_dir.rstrip('/') + '/'
        os.makedirs(output_dir, exist_ok=True)
    audio.write('{}{}_Instruments.wav'.format(output_dir, basename), y_wave.T, sr)
    audio.write('{}{}_Vocals.wav'.format(output_dir, basename), v_wave.T, sr)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
