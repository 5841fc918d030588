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
