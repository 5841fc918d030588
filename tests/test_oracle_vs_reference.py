"""Pin the oracle restatements against the reference's own Python (CPU, build container only)."""
import numpy as np
import pytest
import torch

from oracle import cascaded_net, separator, stft_np, train_step, weights


def test_state_dict_spec_matches_reference(reference_lib):
    ref = reference_lib.nets.CascadedNet(2048, 1024, 32, 128)
    ref_sd = ref.state_dict()
    spec = weights.state_dict_spec()
    assert [k for k, _, _ in spec] == list(ref_sd.keys())
    for k, shape, _ in spec:
        assert tuple(ref_sd[k].shape) == tuple(shape), k
    assert len(spec) == 689


@pytest.fixture(scope='module')
def small_pair(reference_lib):
    # nout=8 keeps the CPU test fast; topology identical to the default net.
    ref = reference_lib.nets.CascadedNet(512, 256, 8, 32)
    sd = weights.make_state_dict(7, n_fft=512, nout=8, nout_lstm=32)
    ref.load_state_dict(sd)
    return ref, sd


def test_forward_eval_matches_reference(small_pair):
    ref, sd = small_pair
    ref.eval()
    x = torch.rand(2, 2, 257, 160, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = ref.predict_mask(x)
        got = cascaded_net.predict_mask(x, sd, n_fft=512)
        want_p = ref.predict(x)
        got_p = cascaded_net.predict(x, sd, n_fft=512)
    assert got.shape == want.shape == (2, 2, 257, 32)
    assert torch.allclose(got, want, atol=2e-6, rtol=0)
    assert torch.allclose(got_p, want_p, atol=2e-6, rtol=0)
    assert 0.02 < float(want.std())        # mask is not saturated


def test_train_step_matches_reference(small_pair, reference_lib):
    """fp64 on both sides: the restatement is exact (fp32 gradients through tiny-batch
    BatchNorm are chaotic at the 10 % level -- see DESIGN.md, training tolerance)."""
    ref32, sd0 = small_pair
    import copy
    ref = copy.deepcopy(ref32).double()
    sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    ref.load_state_dict(sd)
    ref.train()
    B = 2
    X, y = train_step.synth_batch(B, T=64, n_fft=512, seed=5)
    X, y = X.double(), y.double()
    masks = {k: v.double() for k, v in train_step.dropout_masks(B, seed=9, nout=8).items()}

    class Inject(torch.nn.Module):
        def __init__(self, keep):
            super().__init__()
            self.keep = keep

        def forward(self, t):
            return t * self.keep[:, :, None, None]

    for name, keep in masks.items():
        ref.get_submodule(name).dropout = Inject(keep)

    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, ref.parameters()), lr=1e-3)
    my_opt = train_step.Adam(lr=1e-3)
    for _ in range(2):
        mask = ref(X)
        loss = torch.nn.L1Loss()(mask * X, y)
        loss.backward()
        ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
        opt.step()
        ref.zero_grad()

        my_loss, grads = train_step.loss_and_grads(sd, X, y, n_fft=512, dropout=masks)
        assert abs(my_loss - float(loss.detach())) < 1e-12
        assert set(grads) == set(ref_grads)
        assert 'aux_out.weight' not in grads
        for k in grads:
            assert float((grads[k] - ref_grads[k]).abs().max()) <= 1e-9, k
        my_opt.step(sd, grads)

    ref_sd = ref.state_dict()
    for k in sd:
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == int(ref_sd[k]) == 2
            continue
        if k.endswith('dense.0.bias'):
            continue    # true gradient is exactly 0 (BatchNorm1d follows): Adam amplifies noise
        assert float((sd[k] - ref_sd[k]).abs().max()) <= 1e-7, k


def test_separator_matches_reference(small_pair, reference_lib):
    ref, sd = small_pair
    ref.eval()
    rng = np.random.default_rng(0)
    T = 300
    X_spec = (rng.standard_normal((2, 257, T)) + 1j * rng.standard_normal((2, 257, T))).astype(np.complex64)
    sp = reference_lib.inference.Separator(ref, torch.device('cpu'), batchsize=2, cropsize=160)
    sp.offset = 64
    import tqdm  # reference progress bar writes to stderr; harmless
    for tta in (False, True):
        want_y, want_v = (sp.separate_tta if tta else sp.separate)(X_spec.copy())
        got_y, got_v = separator.separate(X_spec.copy(), sd, tta=tta, n_fft=512, batchsize=2, cropsize=160)
        assert got_y.shape == want_y.shape == (2, 257, T)
        assert np.abs(got_y - want_y).max() < 1e-4 * np.abs(X_spec).max()
        assert np.abs(got_v - want_v).max() < 1e-4 * np.abs(X_spec).max()
    assert separator.make_padding(1292, 256, 64) == (64, 180, 128)
    assert separator.make_padding(1280, 256, 64) == (64, 192, 128)


def test_stft_restatement_vs_torch():
    wave = separator.synth_wave(2.0, seed=1)
    spec = stft_np.wave_to_spectrogram(wave, 1024, 2048)
    L = wave.shape[1]
    assert spec.shape == (2, 1025, 1 + L // 1024) and spec.dtype == np.complex64
    win = torch.hann_window(2048, periodic=True)
    want = torch.stft(torch.from_numpy(wave), 2048, 1024, window=win, center=True,
                      pad_mode='constant', return_complex=True).numpy()
    assert np.abs(spec - want).max() < 2e-5 * np.abs(want).max()
    back = stft_np.spectrogram_to_wave(spec, 1024)
    T = spec.shape[2]
    assert back.shape == (2, 1024 * (T - 1)) and back.dtype == np.float32
    want_b = torch.istft(torch.from_numpy(spec), 2048, 1024, window=win, center=True,
                         length=1024 * (T - 1)).numpy()
    assert np.abs(back - want_b).max() < 1e-5
    assert np.abs(back - wave[:, :back.shape[1]]).max() < 1e-5


def test_stft_restatement_vs_scipy():
    """A second, independent implementation (scipy.signal) of the same transform: periodic Hann, centred frames
    with zero padding of n_fft/2 -- librosa 0.10's defaults, which the reference relies on (lib/spec_utils.py:26-31).
    scipy scales by 1/sum(window); librosa does not."""
    import scipy.signal as sig
    wave = separator.synth_wave(1.5, seed=4)
    n_fft, hop = 2048, 1024
    spec = stft_np.wave_to_spectrogram(wave, hop, n_fft)
    win = sig.get_window('hann', n_fft, fftbins=True)
    _, _, Z = sig.stft(wave.astype(np.float64), window=win, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft,
                       boundary='zeros', padded=False, return_onesided=True)
    Z = Z * win.sum()
    T = spec.shape[2]
    assert Z.shape[1] == 1025 and Z.shape[2] >= T
    assert np.abs(spec - Z[:, :, :T]).max() < 2e-5 * np.abs(Z).max()


# ---- training sample pipeline: lib/dataset.py VocalRemoverTrainingSet ---------------------
def _synthetic_training_set(tmp_path, bins=33, lengths=(90, 140, 75), seed=0):
    """Cached spectrograms in the reference's on-disk format: [T, 2, bins] complex64 .npy + coef."""
    rng = np.random.RandomState(seed)
    ts = []
    for i, T in enumerate(lengths):
        pair = []
        for tag in ('X', 'y'):
            a = (rng.randn(T, 2, bins) + 1j * rng.randn(T, 2, bins)).astype(np.complex64) * (0.3 + i)
            a[rng.rand(T, 2, bins) < 0.02] = 0          # exact zeros exercise angle(0)
            path = str(tmp_path / ('song%d_%s.npy' % (i, tag)))
            np.save(path, a)
            pair.append((path, a))
        coef = np.max([np.abs(pair[0][1]).max(), np.abs(pair[1][1]).max()])    # lib/dataset.py:214
        ts.append([pair[0][0], pair[1][0], coef])
    return ts


def _reduction_weight(bins, level=0.2):
    # train.py:197-205 with these bins
    u, s = max(1, bins // 10), bins - bins // 8
    return np.concatenate([np.linspace(0, 1, u, dtype=np.float32)[:, None],
                           np.linspace(1, 0, s - u, dtype=np.float32)[:, None],
                           np.zeros((bins - s, 1), dtype=np.float32)], axis=0) * level


def test_training_sample_pipeline_matches_reference(reference_lib, tmp_path):
    import importlib
    from oracle import dataset_np
    ref_ds_mod = importlib.import_module('lib.dataset')
    ts = _synthetic_training_set(tmp_path)
    rw = _reduction_weight(33)
    ref = ref_ds_mod.VocalRemoverTrainingSet(ts * 2, cropsize=32, reduction_rate=0.5, reduction_weight=rw,
                                             mixup_rate=0.5, mixup_alpha=0.4)
    seen = set()
    for seed in range(24):
        idx = seed % len(ref)
        np.random.seed(seed)
        want_X, want_y = ref[idx]
        nxt_ref = np.random.uniform()
        np.random.seed(seed)
        got_X, got_y = dataset_np.training_sample(ts * 2, idx, 32, 0.5, rw, 0.5, 0.4)
        nxt = np.random.uniform()
        assert nxt == nxt_ref, 'random stream consumed differently (seed %d)' % seed
        assert got_X.shape == want_X.shape == (2, 33, 32)
        assert np.abs(got_X - want_X).max() < 1e-6 and np.abs(got_y - want_y).max() < 1e-6
        seen.add(bool(np.abs(want_X - want_y).max() == 0))
    assert len(seen) >= 1


def test_training_set_host_plan_consumes_rng_like_reference(reference_lib, tmp_path):
    """The product's host half (vocal_remover_amd/dataset.py: plan) draws the same numbers in the same order."""
    import importlib
    import __graft_entry__
    pkg = __graft_entry__.load_package()
    ref_ds_mod = importlib.import_module('lib.dataset')
    ts = _synthetic_training_set(tmp_path)
    rw = _reduction_weight(33)
    ref = ref_ds_mod.VocalRemoverTrainingSet(ts, 32, 0.5, rw, 0.5, 0.4)
    mine = pkg.dataset.VocalRemoverTrainingSet(ts, 32, 0.5, rw, 0.5, 0.4)
    for seed in range(16):
        np.random.seed(seed)
        ref[seed % 3]
        a = np.random.uniform()
        np.random.seed(seed)
        mine.plan(seed % 3)
        assert np.random.uniform() == a
    with pytest.raises(RuntimeError):
        mine[0]                                   # no model / no GPU: loud failure, no CPU fallback
