"""Committed outputs of the REFERENCE's own Python (tests/golden/make_golden.py) as the anchor:
   - CPU (not gpu): the oracle restatements reproduce them;
   - GPU: the HIP path reproduces them (no oracle in the loop)."""
import os

import numpy as np
import pytest
import torch

from oracle import cascaded_net, separator, train_step, weights

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_outputs.npz'))


def _wsum(sd):
    return float(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.is_floating_point()))


def _small_sd():
    sd = weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32)
    assert abs(_wsum(sd) - float(G['small_wsum'])) < 1e-6 * float(G['small_wsum']), 'seeded weights drifted'
    return sd


def _small_inputs():
    x = torch.rand(2, 2, 257, 160, generator=torch.Generator().manual_seed(0))
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((2, 257, 300)) + 1j * rng.standard_normal((2, 257, 300))).astype(np.complex64)
    return x, X


def test_oracle_reproduces_reference_fixtures():
    sd = _small_sd()
    x, X = _small_inputs()
    with torch.no_grad():
        assert np.abs(cascaded_net.predict_mask(x, sd, n_fft=512).numpy() - G['small_mask']).max() < 2e-6
        assert np.abs(cascaded_net.predict(x, sd, n_fft=512).numpy() - G['small_pred']).max() < 2e-6
    y, v = separator.separate(X.copy(), sd, n_fft=512, batchsize=2, cropsize=160)
    assert np.abs(y[:, ::5] - G['sep_y']).max() < 1e-5 and np.abs(v[:, ::5] - G['sep_v']).max() < 1e-5
    yt, _ = separator.separate(X.copy(), sd, tta=True, n_fft=512, batchsize=2, cropsize=160)
    assert np.abs(yt[:, ::5] - G['sep_tta_y']).max() < 1e-5


def test_oracle_full_net_fixture():
    sd = weights.make_state_dict(1234)
    assert abs(_wsum(sd) - float(G['full_wsum'])) < 1e-6 * float(G['full_wsum'])
    xf = torch.rand(1, 2, 1025, 144, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        got = cascaded_net.predict_mask(xf, sd).numpy()
    assert got.shape == G['full_mask'].shape == (1, 2, 1025, 16)
    assert np.abs(got - G['full_mask']).max() < 2e-6


def test_oracle_train_step_fixture():
    sd = _small_sd()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    X, y = train_step.synth_batch(2, T=64, n_fft=512, seed=5)
    masks = {k: v.double() for k, v in train_step.dropout_masks(2, seed=9, nout=8).items()}
    loss, grads = train_step.loss_and_grads(sd64, X.double(), y.double(), n_fft=512, dropout=masks)
    assert abs(loss - float(G['train_loss'])) < 1e-12
    names = [str(s) for s in G['train_grad_names']]
    assert sorted(names) == sorted(grads)
    for k, nrm in zip(names, G['train_grad_norms']):
        assert abs(float(grads[k].norm()) - float(nrm)) <= 1e-9 * max(1.0, float(nrm)), k
    for key in G.files:
        if key.startswith('train_grad::'):
            assert np.abs(grads[key[12:]].numpy() - G[key]).max() < 1e-10, key
    opt = train_step.Adam(lr=1e-3)
    opt.step(sd64, grads)
    for key in G.files:
        if key.startswith('train_after::'):
            assert np.abs(sd64[key[13:]].numpy() - G[key]).max() < 1e-9, key


@pytest.mark.gpu
def test_hip_path_reproduces_reference_fixtures(vr):
    sd = _small_sd()
    x, X = _small_inputs()
    model = vr.nets.CascadedNet(512, 256, 8, 32)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0')).eval()
    assert np.abs(model.predict_mask(x).numpy() - G['small_mask']).max() < 1e-4
    assert np.abs(model.predict(x).numpy() - G['small_pred']).max() < 1e-4
    sp = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=2, cropsize=160)
    y, v = sp.separate(X.copy())
    scale = np.abs(X).max()
    assert np.abs(y[:, ::5] - G['sep_y']).max() < 1e-4 * scale and np.abs(v[:, ::5] - G['sep_v']).max() < 1e-4 * scale
    yt, _ = sp.separate_tta(X.copy())
    assert np.abs(yt[:, ::5] - G['sep_tta_y']).max() < 1e-4 * scale


@pytest.mark.gpu
def test_hip_full_net_fixture(vr):
    sd = weights.make_state_dict(1234)
    model = vr.nets.CascadedNet(2048, 1024, 32, 128)
    model.load_state_dict(sd)
    model.to(torch.device('cuda:0')).eval()
    xf = torch.rand(1, 2, 1025, 144, generator=torch.Generator().manual_seed(2))
    got = model.predict_mask(xf.to('cuda:0')).cpu().numpy()
    diff = np.abs(got - G['full_mask'])
    assert diff.max() < 1e-4 and diff.mean() < 1e-5


# ---- training input pipeline fixtures (tests/golden/make_golden_dataset.py, the reference's lib/dataset.py) ------
GD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dataset_pipeline.npz'))
_DS_CROP, _DS_PARAMS, _DS_SEEDS = 32, dict(reduction_rate=0.5, mixup_rate=0.5, mixup_alpha=0.4), 12


def _golden_training_set(tmp_path):
    ts = []
    for i in range(3):
        paths = []
        for tag in ('X', 'y'):
            p = str(tmp_path / ('song%d_%s.npy' % (i, tag)))
            np.save(p, GD['song%d_%s' % (i, tag)])
            paths.append(p)
        ts.append([paths[0], paths[1], GD['coef%d' % i][()]])
    return ts * 2


def test_oracle_training_pipeline_reproduces_reference_fixture(tmp_path):
    from oracle import dataset_np
    ts = _golden_training_set(tmp_path)
    for seed in range(_DS_SEEDS):
        np.random.seed(seed)
        X, y = dataset_np.training_sample(ts, seed % len(ts), _DS_CROP, _DS_PARAMS['reduction_rate'], GD['reduction_weight'],
                                          _DS_PARAMS['mixup_rate'], _DS_PARAMS['mixup_alpha'])
        assert np.abs(X - GD['seed%d_X' % seed]).max() < 1e-6 and np.abs(y - GD['seed%d_y' % seed]).max() < 1e-6


@pytest.mark.gpu
def test_hip_training_pipeline_reproduces_reference_fixture(vr, tmp_path):
    """Device pipeline (host draws + vr_augment_batch) against the reference's own outputs, no oracle in the loop."""
    model = vr.nets.CascadedNet(512, 256, 8, 32)
    model.to(torch.device('cuda:0'))
    ts = _golden_training_set(tmp_path)
    ds = vr.dataset.VocalRemoverTrainingSet(ts, cropsize=_DS_CROP, reduction_weight=GD['reduction_weight'], model=model,
                                            **_DS_PARAMS)
    for seed in range(_DS_SEEDS):
        np.random.seed(seed)
        X, y = ds[seed % len(ds)]
        scale = float(np.abs(GD['seed%d_X' % seed]).max()) + 1e-6
        assert float(np.abs(X.cpu().numpy() - GD['seed%d_X' % seed]).max()) < 3e-6 * scale, seed
        assert float(np.abs(y.cpu().numpy() - GD['seed%d_y' % seed]).max()) < 3e-6 * scale, seed
