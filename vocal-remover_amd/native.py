"""ctypes binding of libvr_mi355.so (include/vr_mi355.h).  Fails loudly when the library or a GPU
is missing -- there is no CPU or PyTorch fallback in this package."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VR_LIB_PATH: another build of the same library, for A / B runs of two builds in one gpurun call)
LIB_PATH = os.environ.get('VR_LIB_PATH') or os.path.join(_HERE, 'libvr_mi355.so')

c_f32p = ctypes.c_void_p
c_i64p = ctypes.POINTER(ctypes.c_int64)

_SIGNATURES = {
    'vr_last_error': (ctypes.c_char_p, []),
    'vr_create': (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_void_p)]),
    'vr_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_num_params': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_param_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, c_i64p,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int)]),
    'vr_set_param': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, c_i64p, ctypes.c_int]),
    'vr_get_param': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]),
    'vr_set_mode': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'vr_forward': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  c_f32p, ctypes.c_int]),
    'vr_stft': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_int]),
    'vr_istft': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int]),
    'vr_separate': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, c_f32p, c_f32p, ctypes.c_int]),
    'vr_separate_wave': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int]),
    'vr_train_step': (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.POINTER(ctypes.c_float), c_f32p, ctypes.c_int]),
    'vr_forward_train': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int]),
    'vr_backward': (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int]),
    'vr_param_arena': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    'vr_adam_step': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_double] * 5),
    'vr_zero_grad': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_get_adam_state': (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int64, c_i64p]),
    'vr_set_adam_state': (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int64]),
    'vr_graph_generation': (ctypes.c_int, [ctypes.c_void_p, c_i64p, ctypes.POINTER(ctypes.c_int)]),
    'vr_get_grad': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_f32p, ctypes.c_int64]),
    'vr_set_dropout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, c_f32p, ctypes.c_int]),
    'vr_grad_arena': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    'vr_set_option': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    'vr_augment_batch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'vr_validate_step': (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_float)]),
    'vr_comm_unique_id': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_comm_init': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'vr_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_allreduce_grads': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'vr_broadcast_params': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'vr_debug_kernel': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    'vr_resample': (ctypes.c_int, [ctypes.c_int, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int64]),
    'vr_xcorr_argmax': (ctypes.c_int, [ctypes.c_int, c_f32p, ctypes.c_int64, c_f32p, ctypes.c_int64, c_i64p]),
    'vr_profile_begin': (ctypes.c_int, [ctypes.c_void_p]),
    'vr_profile_end': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_double)]),
    'vr_profile_report': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    'vr_debug_conv2d': (ctypes.c_int, [ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 4 + [c_f32p] + [ctypes.c_int] * 6
                        + [c_f32p, ctypes.c_float, c_f32p, c_f32p, c_f32p]),
    'vr_debug_conv2d_backward': (ctypes.c_int, [ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 4 + [c_f32p]
                                 + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_float, c_f32p, c_f32p, c_f32p]),
    'vr_debug_merge_artifacts_weight': (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                       c_f32p]),
    'vr_debug_record_taps': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'vr_debug_get_tap': (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, c_f32p, ctypes.c_int64, c_i64p]),
}

_lib = None


def lib():
    """Load libvr_mi355.so once; raise if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('%s is missing: build it with `python __graft_entry__.py` '
                               '(there is no CPU fallback)' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


class VRError(RuntimeError):
    pass


def check(rc):
    """Map vr_status to the exception the reference would raise at the same spot."""
    if rc >= 0:
        return rc
    msg = lib().vr_last_error().decode('utf-8', 'replace')
    if rc == -5:
        raise ValueError(msg)              # spec_utils.crop_center ValueError (lib/spec_utils.py:15)
    if rc == -6:
        raise AssertionError(msg)          # assert mask.size()[3] > 0 (lib/nets.py:129,139)
    if rc == -7:
        raise IndexError(msg)              # merge_artifacts on a mask with no frame above the threshold
    if rc == -2:
        raise ValueError(msg)
    if rc == -8:
        raise VRError('libvr_mi355 RCCL error: %s' % msg)
    raise VRError('libvr_mi355 error %d: %s' % (rc, msg))


def debug_kernel(handle, name, dims, fparams, inputs, outputs):
    """vr_debug_kernel: `inputs` / `outputs` are lists of C-contiguous float32 numpy arrays (or None)."""
    dims_a = (ctypes.c_int64 * max(len(dims), 1))(*[int(d) for d in dims])
    fp_a = (ctypes.c_float * max(len(fparams), 1))(*[float(f) for f in fparams])
    ins = (ctypes.c_void_p * max(len(inputs), 1))(*[a.ctypes.data if a is not None else None for a in inputs])
    outs = (ctypes.c_void_p * max(len(outputs), 1))(*[a.ctypes.data if a is not None else None for a in outputs])
    for a in list(inputs) + list(outputs):
        assert a is None or (a.flags['C_CONTIGUOUS'] and a.dtype == np.float32)
    check(lib().vr_debug_kernel(handle.h, name.encode(), dims_a, len(dims), fp_a, len(fparams), ins, len(inputs), outs,
                                len(outputs)))


def np_ptr(a):
    assert a.flags['C_CONTIGUOUS']
    return ctypes.c_void_p(a.ctypes.data)


class Handle:
    """Owns one vr_handle (one GPU, one stream)."""

    def __init__(self, device, n_fft, hop_length, nout, nout_lstm):
        self._h = ctypes.c_void_p()
        check(lib().vr_create(int(device), int(n_fft), int(hop_length), int(nout), int(nout_lstm),
                              ctypes.byref(self._h)))
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            lib().vr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def h(self):
        if not self._h.value:
            raise VRError('handle is closed')
        return self._h

    def param_infos(self):
        out = []
        n = lib().vr_num_params(self.h)
        buf = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int64 * 4)()
        nd, is64, tr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for i in range(n):
            check(lib().vr_param_info(self.h, i, buf, 256, shape, ctypes.byref(nd), ctypes.byref(is64),
                                      ctypes.byref(tr)))
            out.append((buf.value.decode(), tuple(int(shape[j]) for j in range(nd.value)), bool(is64.value),
                        bool(tr.value)))
        return out

    def set_param(self, key, arr):
        arr = np.require(arr, requirements=['C'])      # (ascontiguousarray would turn 0-d into 1-d)
        shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
        check(lib().vr_set_param(self.h, key.encode(), np_ptr(arr), shape, arr.ndim))

    def get_param(self, key, shape, is_int64):
        arr = np.empty(shape, dtype=np.int64 if is_int64 else np.float32)
        check(lib().vr_get_param(self.h, key.encode(), np_ptr(arr), arr.nbytes))
        return arr
