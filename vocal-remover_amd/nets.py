"""CascadedNet facade with the reference's surface (lib/nets.py:44-141) over libvr_mi355.so.

What callers touch and what happens here:

    nets.CascadedNet(n_fft, hop_length, nout=32, nout_lstm=128)  -> host-side state (689 tensors)
    .load_state_dict(dict) / .state_dict()                      -> reference keys / torch layouts
    .to(device)                                                 -> creates the native handle on that GPU
    .eval() / .train()                                          -> vr_set_mode
    .offset, .n_fft, .hop_length, .max_bin, .output_bin         -> same attributes
    .forward(x) / __call__(x) / .predict_mask(x) / .predict(x)  -> vr_forward (modes 0 / 1 / 2)

Inputs and outputs are torch tensors; a tensor already on the handle's GPU is passed by device
pointer (no host round trip), a CPU tensor is copied in by the library.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import native


def _cba(spec, p, nin, nout, k):
    spec.append((p + '.conv.0.weight', (nout, nin, k, k), 'conv'))
    spec.append((p + '.conv.1.weight', (nout,), 'ones'))
    spec.append((p + '.conv.1.bias', (nout,), 'zeros'))
    spec.append((p + '.conv.1.running_mean', (nout,), 'zeros'))
    spec.append((p + '.conv.1.running_var', (nout,), 'ones'))
    spec.append((p + '.conv.1.num_batches_tracked', (), 'nbt'))


def _base_net(spec, p, nin, c, nin_lstm, nout_lstm):
    _cba(spec, p + '.enc1', nin, c, 3)
    chans = (c, 2 * c, 4 * c, 6 * c, 8 * c)
    for i in range(4):
        _cba(spec, '%s.enc%d.conv1' % (p, i + 2), chans[i], chans[i + 1], 3)
        _cba(spec, '%s.enc%d.conv2' % (p, i + 2), chans[i + 1], chans[i + 1], 3)
    _cba(spec, p + '.aspp.conv1.1', 8 * c, 8 * c, 1)
    _cba(spec, p + '.aspp.conv2', 8 * c, 8 * c, 1)
    for i in (3, 4, 5):
        _cba(spec, '%s.aspp.conv%d' % (p, i), 8 * c, 8 * c, 3)
    _cba(spec, p + '.aspp.bottleneck', 40 * c, 8 * c, 1)
    _cba(spec, p + '.dec4.conv1', 14 * c, 6 * c, 3)
    _cba(spec, p + '.dec3.conv1', 10 * c, 4 * c, 3)
    _cba(spec, p + '.dec2.conv1', 6 * c, 2 * c, 3)
    q = p + '.lstm_dec2'
    _cba(spec, q + '.conv', 2 * c, 1, 1)
    hid = nout_lstm // 2
    for sfx in ('', '_reverse'):
        spec.append((q + '.lstm.weight_ih_l0' + sfx, (4 * hid, nin_lstm), ('uniform', 1.0 / math.sqrt(hid))))
        spec.append((q + '.lstm.weight_hh_l0' + sfx, (4 * hid, hid), ('uniform', 1.0 / math.sqrt(hid))))
        spec.append((q + '.lstm.bias_ih_l0' + sfx, (4 * hid,), ('uniform', 1.0 / math.sqrt(hid))))
        spec.append((q + '.lstm.bias_hh_l0' + sfx, (4 * hid,), ('uniform', 1.0 / math.sqrt(hid))))
    spec.append((q + '.dense.0.weight', (nin_lstm, nout_lstm), ('uniform', 1.0 / math.sqrt(nout_lstm))))
    spec.append((q + '.dense.0.bias', (nin_lstm,), ('uniform', 1.0 / math.sqrt(nout_lstm))))
    spec.append((q + '.dense.1.weight', (nin_lstm,), 'ones'))
    spec.append((q + '.dense.1.bias', (nin_lstm,), 'zeros'))
    spec.append((q + '.dense.1.running_mean', (nin_lstm,), 'zeros'))
    spec.append((q + '.dense.1.running_var', (nin_lstm,), 'ones'))
    spec.append((q + '.dense.1.num_batches_tracked', (), 'nbt'))
    _cba(spec, p + '.dec1.conv1', 3 * c + 1, c, 3)


def state_spec(n_fft, nout, nout_lstm):
    """(key, shape, init) in the reference's registration order (lib/nets.py:59-80)."""
    nin = 2
    nin_lstm = (n_fft // 2) // 2
    spec = []
    _base_net(spec, 'stg1_low_band_net.0', nin, nout // 2, nin_lstm // 2, nout_lstm)
    _cba(spec, 'stg1_low_band_net.1', nout // 2, nout // 4, 1)
    _base_net(spec, 'stg1_high_band_net', nin, nout // 4, nin_lstm // 2, nout_lstm // 2)
    _base_net(spec, 'stg2_low_band_net.0', nout // 4 + nin, nout, nin_lstm // 2, nout_lstm)
    _cba(spec, 'stg2_low_band_net.1', nout, nout // 2, 1)
    _base_net(spec, 'stg2_high_band_net', nout // 4 + nin, nout // 2, nin_lstm // 2, nout_lstm // 2)
    _base_net(spec, 'stg3_full_band_net', 3 * nout // 4 + nin, nout, nin_lstm, nout_lstm)
    spec.append(('out.weight', (nin, nout, 1, 1), 'conv'))
    spec.append(('aux_out.weight', (nin, 3 * nout // 4, 1, 1), 'conv'))
    return spec


def _init_tensor(shape, init):
    if init == 'ones':
        return torch.ones(shape)
    if init == 'zeros':
        return torch.zeros(shape)
    if init == 'nbt':
        return torch.zeros((), dtype=torch.int64)
    if init == 'conv':      # torch's default Conv2d init: kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))
        bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
        return torch.empty(shape).uniform_(-bound, bound)
    kind, bound = init
    assert kind == 'uniform'
    return torch.empty(shape).uniform_(-bound, bound)


class _ForwardTrain(torch.autograd.Function):
    """mask = model(X) with the graph kept inside the native handle (vr_forward_train); backward hands dLoss/dmask to
    vr_backward, which ACCUMULATES the parameter gradients in the gradient arena (= flat.grad), like autograd would."""

    @staticmethod
    def forward(ctx, flat, x, model):
        h = model._need_handle()
        if x.dim() != 4 or x.shape[1] != 2 or x.shape[2] != model.output_bin:
            raise ValueError('expected input [B, 2, %d, T], got %s' % (model.output_bin, tuple(x.shape)))
        on_dev = x.is_cuda
        if on_dev and x.device.index != h.device:
            raise RuntimeError('input is on %s but the model is on cuda:%d' % (x.device, h.device))
        xc = x.detach().to(torch.float32).contiguous()
        mask = torch.empty_like(xc)
        # a pending torch-side write to the parameter arena (torch.optim.Adam.step) must land before the library's own
        # stream reads it -- also when X itself comes from the host
        torch.cuda.current_stream(torch.device('cuda', h.device)).synchronize()
        native.check(native.lib().vr_forward_train(h.h, xc.data_ptr(), int(on_dev), int(xc.shape[0]), int(xc.shape[3]),
                                                   mask.data_ptr(), int(on_dev)))
        ctx.model = model
        ctx.graph = model._graph_generation()[0]       # the handle keeps ONE graph: backward must name this one
        ctx.handle_gen = model._handle_gen
        model._host_stale = True
        return mask

    @staticmethod
    def backward(ctx, dmask):
        model = ctx.model
        h = model._need_handle()
        gen, valid = model._graph_generation()
        if ctx.handle_gen != model._handle_gen or gen != ctx.graph or not valid:
            raise RuntimeError('backward through a CascadedNet forward whose graph is gone: the native handle keeps the graph of '
                               'the LAST model(X) only (another forward / predict / validate call, a second backward or a '
                               '.to(device) in between frees it)')
        on_dev = dmask.is_cuda
        d = dmask.detach().to(torch.float32).contiguous()
        torch.cuda.current_stream(torch.device('cuda', h.device)).synchronize()
        native.check(native.lib().vr_backward(h.h, d.data_ptr(), int(on_dev)))
        model._flat_parameter()                     # (re-)attach .grad to the arena view
        return None, None, None


class CascadedNet(object):

    def __init__(self, n_fft, hop_length, nout=32, nout_lstm=128, is_complex=False):
        if is_complex:
            raise NotImplementedError('is_complex=True is unreachable from every reference caller '
                                      '(lib/nets.py:83-84,104-107) and is not part of the MI355X hot path')
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.is_complex = False
        self.nout = nout
        self.nout_lstm = nout_lstm
        self.max_bin = n_fft // 2
        self.output_bin = n_fft // 2 + 1
        self.nin_lstm = self.max_bin // 2
        self.offset = 64
        self.training = True
        self._spec = state_spec(n_fft, nout, nout_lstm)
        self._state = OrderedDict((k, _init_tensor(shape, init)) for k, shape, init in self._spec)
        self._handle = None
        self._handle_gen = 0          # bumped whenever a native handle is created or closed
        self._flat = self._flat_grad = None
        self._flat_key = None
        self._device = torch.device('cpu')
        self._host_stale = False      # device weights newer than self._state (after training steps)

    # ---- nn.Module-like surface -------------------------------------------------------------------
    def state_dict(self):
        self._pull()
        return OrderedDict((k, v.clone()) for k, v in self._state.items())

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._state if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._state]
        if strict and (missing or unexpected):
            raise RuntimeError('Error(s) in loading state_dict for CascadedNet: missing %s unexpected %s'
                               % (missing[:5], unexpected[:5]))
        for k, v in state_dict.items():
            if k not in self._state:
                continue
            v = torch.as_tensor(v).detach().cpu()
            if tuple(v.shape) != tuple(self._state[k].shape):
                raise RuntimeError('size mismatch for %s: %s vs %s' % (k, tuple(v.shape), tuple(self._state[k].shape)))
            self._state[k] = v.to(self._state[k].dtype).contiguous().clone()
        self._host_stale = False
        self._push()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda':
            index = device.index if device.index is not None else torch.cuda.current_device()
            if self._handle is None or self._handle.device != index:
                self._pull()
                self._drop_handle()
                self._handle = native.Handle(index, self.n_fft, self.hop_length, self.nout, self.nout_lstm)
                self._handle_gen += 1
                self._device = torch.device('cuda', index)
                self._push()
                native.check(native.lib().vr_set_mode(self._handle.h, int(self.training)))
                # nn.Dropout2d(0.1) on the ASPP outputs is live in train mode (lib/layers.py:90): the library's
                # generator is seeded from torch's seed, so torch.manual_seed() makes runs repeatable
                native.check(native.lib().vr_set_dropout(self._handle.h, 1, (torch.initial_seed() + index) & (2 ** 63 - 1),
                                                         None, 0))
        elif device.type == 'cpu':
            self._pull()
            if self._flat is not None and self._handle is not None:
                import warnings
                warnings.warn('CascadedNet.to("cpu"): the parameter handed out by parameters() is empty until the model moves back '
                              'to a GPU (there is no CPU compute path); an optimizer that holds it steps nothing meanwhile')
            self._drop_handle()
            self._device = device
        else:
            raise RuntimeError('CascadedNet (MI355X-native) supports cuda devices only, got %s' % device)
        return self

    def _drop_handle(self):
        """Close the native handle.  The flat Parameter / gradient views point into its arenas: detach them first, so that an
        optimizer that still holds the Parameter never touches freed device memory.  The Parameter OBJECT is kept: the next handle
        (`.to(cuda:N)`) rebinds its .data / .grad to the new arenas, so an optimizer built from model.parameters() before the move
        keeps stepping the model, as with nn.Module.to (a torch optimizer's own state tensors do not move -- build it after .to(),
        as the reference does, train.py:211-218)."""
        if self._handle is None:
            return
        if self._flat is not None:
            self._flat.grad = None
            self._flat.data = torch.empty(0)
        self._flat_grad = None
        self._flat_key = None
        self._handle.close()
        self._handle = None
        self._handle_gen += 1

    def _graph_generation(self):
        import ctypes
        gen, valid = ctypes.c_int64(), ctypes.c_int()
        native.check(native.lib().vr_graph_generation(self._need_handle().h, ctypes.byref(gen), ctypes.byref(valid)))
        return int(gen.value), bool(valid.value)

    def cuda(self, index=None):
        return self.to(torch.device('cuda', index if index is not None else 0))

    def train(self, mode=True):
        self.training = bool(mode)
        if self._handle is not None:
            native.check(native.lib().vr_set_mode(self._handle.h, int(self.training)))
        return self

    def eval(self):
        return self.train(False)

    # ---- compute ------------------------------------------------------------------------------------
    def _need_handle(self):
        if self._handle is None:
            raise RuntimeError('CascadedNet has no MI355X handle: call .to(torch.device("cuda:N")) first '
                               '(this package has no CPU fallback)')
        return self._handle

    def _run(self, x, mode):
        h = self._need_handle()
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if x.is_complex():
            raise NotImplementedError('is_complex inputs are not supported (pass torch.abs(X))')
        if x.dim() != 4 or x.shape[1] != 2 or x.shape[2] != self.output_bin:
            raise ValueError('expected input [B, 2, %d, T], got %s' % (self.output_bin, tuple(x.shape)))
        B, T = int(x.shape[0]), int(x.shape[3])
        Wm = T if mode == 0 else T - 2 * self.offset
        on_dev = x.is_cuda
        if on_dev and x.device.index != h.device:
            raise RuntimeError('input is on %s but the model is on cuda:%d' % (x.device, h.device))
        x = x.detach().to(torch.float32).contiguous()
        out = torch.empty((B, 2, self.output_bin, max(Wm, 1)), dtype=torch.float32,   # Wm <= 0 -> native raises
                          device=x.device if on_dev else 'cpu')
        if on_dev:
            torch.cuda.current_stream(x.device).synchronize()
        native.check(native.lib().vr_forward(h.h, x.data_ptr(), int(on_dev), B, T, mode, out.data_ptr(), int(on_dev)))
        return out

    def forward(self, x):
        """CascadedNet.forward (lib/nets.py:82-117): mask [B,2,n_fft/2+1,T].

        Under model.train() with autograd enabled (the reference's train_epoch, train.py:81) the returned mask is
        differentiable: `loss.backward()` reaches the native backward pass through _ForwardTrain and the gradients
        accumulate in the library's gradient arena = `.grad` of the flat parameter `parameters()` hands to the optimizer."""
        if self.training and torch.is_grad_enabled() and self._handle is not None:
            if not torch.is_tensor(x):
                x = torch.as_tensor(x)
            return _ForwardTrain.apply(self._flat_parameter(), x, self)
        return self._run(x, 0)

    __call__ = forward

    def predict_mask(self, x):
        """CascadedNet.predict_mask (lib/nets.py:124-131): mask[..., 64:-64]."""
        return self._run(x, 1)

    def predict(self, x):
        """CascadedNet.predict (lib/nets.py:133-141): (x * mask)[..., 64:-64]."""
        return self._run(x, 2)

    # ---- training (train.py:77-96) ----------------------------------------------------------------------
    def _arena_tensor(self, fn):
        import ctypes
        h = self._need_handle()
        ptr, n = ctypes.c_void_p(), ctypes.c_int64()
        native.check(fn(h.h, ctypes.byref(ptr), ctypes.byref(n)))

        class _Arr(object):
            __cuda_array_interface__ = {'shape': (int(n.value),), 'typestr': '<f4', 'data': (int(ptr.value), False), 'version': 2}
        return torch.as_tensor(_Arr(), device=torch.device('cuda', h.device))

    def _flat_parameter(self):
        """ONE torch Parameter = a zero-copy view of the library's flat fp32 parameter arena (kernel layouts, padding
        included), `.grad` = a view of the gradient arena.  Element-wise optimizers (torch.optim.Adam of train.py:215,
        or vocal_remover_amd.train.Adam's fused kernel) do not care about the layout; padding has zero gradient."""
        self._need_handle()
        key = self._handle_gen                       # (not id(handle): CPython reuses ids of closed handles)
        if self._flat_key != key:
            arena = self._arena_tensor(native.lib().vr_param_arena)
            if self._flat is None:
                self._flat = torch.nn.Parameter(arena, requires_grad=True)
                self._flat._vr_model = self
            else:                                    # a new handle after .to(): the SAME Parameter object, rebound
                self._flat.grad = None
                self._flat.data = arena
            self._flat_grad, self._flat_key = self._arena_tensor(native.lib().vr_grad_arena), key
        if self._flat.grad is None or self._flat.grad.data_ptr() != self._flat_grad.data_ptr():
            self._flat.grad = self._flat_grad
        return self._flat

    def parameters(self):
        """Stands in for nn.Module.parameters() (train.py:216 filters on .requires_grad): the flat parameter."""
        return [self._flat_parameter()]

    def zero_grad(self, set_to_none=False):
        if self._handle is not None:
            native.check(native.lib().vr_zero_grad(self._handle.h))
            if self._flat is not None:
                self._flat.grad = self._flat_grad                     # stays the arena view (never None)

    def train_step(self, X, y, accumulation_steps=1, return_mask=False):
        """mask = model(X); loss = L1Loss()(mask * X, y); (loss / accumulation_steps).backward()
        in one native call.  Returns loss.item() (and the mask if asked)."""
        import ctypes
        h = self._need_handle()
        X = torch.as_tensor(X)
        y = torch.as_tensor(y)
        if X.shape != y.shape or X.dim() != 4 or X.shape[1] != 2 or X.shape[2] != self.output_bin:
            raise ValueError('expected X, y of shape [B, 2, %d, T]' % self.output_bin)
        on_dev = X.is_cuda
        if on_dev != y.is_cuda:
            raise ValueError('X and y must live on the same device')
        X = X.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).contiguous()
        B, T = int(X.shape[0]), int(X.shape[3])
        mask = torch.empty_like(X) if return_mask else None
        if on_dev:
            torch.cuda.current_stream(X.device).synchronize()
        loss = ctypes.c_float()
        native.check(native.lib().vr_train_step(h.h, X.data_ptr(), y.data_ptr(), int(on_dev), B, T,
                                                int(accumulation_steps), ctypes.byref(loss),
                                                mask.data_ptr() if return_mask else None, int(on_dev)))
        self._host_stale = True
        return (loss.value, mask) if return_mask else loss.value

    def validate_step(self, X, y):
        """One batch of train.validate_epoch (train.py:117-127): L1(predict(X), crop_center(y)) -> float."""
        import ctypes
        h = self._need_handle()
        X = torch.as_tensor(X)
        y = torch.as_tensor(y)
        if X.shape != y.shape or X.dim() != 4 or X.shape[1] != 2 or X.shape[2] != self.output_bin:
            raise ValueError('expected X, y of shape [B, 2, %d, T]' % self.output_bin)
        on_dev = X.is_cuda
        if on_dev != y.is_cuda:
            raise ValueError('X and y must live on the same device')
        X = X.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).contiguous()
        if on_dev:
            torch.cuda.current_stream(X.device).synchronize()
        loss = ctypes.c_float()
        native.check(native.lib().vr_validate_step(h.h, X.data_ptr(), y.data_ptr(), int(on_dev), int(X.shape[0]),
                                                   int(X.shape[3]), ctypes.byref(loss)))
        return loss.value

    def grads(self, keys=None):
        """{key: gradient} in torch layouts (param.grad of the reference), for tests."""
        h = self._need_handle()
        out = OrderedDict()
        for k, shape, init in self._spec:
            if init == 'nbt' or k.endswith('running_mean') or k.endswith('running_var'):
                continue
            if keys is not None and k not in keys:
                continue
            arr = np.empty(shape, dtype=np.float32)
            native.check(native.lib().vr_get_grad(h.h, k.encode(), native.np_ptr(arr), arr.nbytes))
            out[k] = torch.from_numpy(arr)
        return out

    def set_option(self, name, value):
        """Numerical options of the library (include/vr_mi355.h: vr_set_option), e.g. 'train_winograd'."""
        h = self._need_handle()
        native.check(native.lib().vr_set_option(h.h, name.encode(), int(value)))

    def set_dropout_masks(self, masks):
        """Inject Dropout2d keep-masks {'<net>.aspp': [B, 8c] tensor of 0 / (1/0.9)} (parity tests);
        None switches dropout OFF (explicit opt-out: the default is on, as in the reference); an int
        (re)seeds the library's own generator."""
        h = self._need_handle()
        if masks is None:
            native.check(native.lib().vr_set_dropout(h.h, 0, 0, None, 0))
            return
        if isinstance(masks, int):
            native.check(native.lib().vr_set_dropout(h.h, 1, masks, None, 0))
            return
        order = ['stg1_low_band_net.0', 'stg1_high_band_net', 'stg2_low_band_net.0', 'stg2_high_band_net',
                 'stg3_full_band_net']
        B = int(next(iter(masks.values())).shape[0])
        buf = np.zeros((5, B * 8 * self.nout), dtype=np.float32)
        for i, name in enumerate(order):
            m = masks[name + '.aspp'].to(torch.float32).contiguous().numpy().reshape(-1)
            buf[i, :m.size] = m
        native.check(native.lib().vr_set_dropout(h.h, 2, 0, native.np_ptr(buf), B))

    # ---- host <-> device weights ----------------------------------------------------------------------
    def _push(self):
        if self._handle is None:
            return
        for k, v in self._state.items():
            self._handle.set_param(k, v.numpy())

    def _pull(self):
        if self._handle is None or not (self._host_stale or self.training):
            return
        for k, shape, init in self._spec:
            arr = self._handle.get_param(k, shape, init == 'nbt')
            self._state[k] = torch.from_numpy(np.array(arr)).reshape(shape)
        self._host_stale = False
