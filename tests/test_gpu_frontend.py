"""GPU tests of the pieces around the hot path (SURVEY section 8b autograd shim, section 8f ranks 3 and 4):

  * the reference's own train_epoch statement sequence (train.py:81-96) -- `model(X)`, `crit(pred * X, y)`,
    `loss.backward()`, `torch.optim.Adam.step()`, `model.zero_grad()` -- runs on the native model through the autograd
    shim and the zero-copy flat parameter, and equals the fused native step;
  * resampling (vr_resample) against the numpy restatement of resampy's 'kaiser_fast' and against its defining
    properties; cross-correlation lag; WAV decode -> separate -> WAV encode through inference.main;
  * dataset preparation + the epoch loop (fit) with best-model saving, the loss json and a resumable checkpoint.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import audio_np, train_step, weights

pytestmark = pytest.mark.gpu

N_FFT, NOUT, NL = 512, 8, 32
DEV = 'cuda:0'


def _model(vr, seed=11):
    sd = weights.make_state_dict(seed, n_fft=N_FFT, nout=NOUT, nout_lstm=NL)
    m = vr.nets.CascadedNet(N_FFT, N_FFT // 2, NOUT, NL)
    m.load_state_dict(sd)
    m.to(torch.device(DEV))
    return m, sd


def test_reference_train_epoch_statements_run_through_the_autograd_shim(vr):
    from vocal_remover_amd import train as vtrain
    X, y = train_step.synth_batch(4, T=64, n_fft=N_FFT, seed=3)
    Xd, yd = X.to(DEV), y.to(DEV)
    # fused native path
    a, _ = _model(vr)
    a.train(); a.set_dropout_masks(None)
    opt_a = vtrain.Adam(a.parameters(), lr=1e-3)
    a.zero_grad()
    losses_a = []
    for i in range(2):
        losses_a.append(a.train_step(Xd[2 * i:2 * i + 2], yd[2 * i:2 * i + 2], 2))
    g_a = a.grads()
    opt_a.step()
    p_a = a.state_dict()
    # the reference's statements, torch's own optimizer (train.py:77-96,215-218)
    b, _ = _model(vr)
    b.set_dropout_masks(None)
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, b.parameters()), lr=1e-3)
    crit_l1 = torch.nn.L1Loss()
    b.train()
    b.zero_grad()
    accumulation_steps = 2
    losses_b = []
    for itr in range(2):
        X_batch, y_batch = Xd[2 * itr:2 * itr + 2], yd[2 * itr:2 * itr + 2]
        pred = b(X_batch)
        assert pred.requires_grad and pred.shape == X_batch.shape
        loss = crit_l1(pred * X_batch, y_batch)
        accum_loss = loss / accumulation_steps
        accum_loss.backward()
        losses_b.append(loss.item())
    g_b = b.grads()
    assert np.abs(np.array(losses_a) - np.array(losses_b)).max() < 1e-6
    for k in g_a:
        if k.endswith('dense.0.bias'):
            continue                                 # exact gradient 0 (a BatchNorm follows the bias): rounding noise only
        s = float(g_a[k].abs().max()) + 1e-12
        assert float((g_a[k] - g_b[k]).abs().max()) <= 1e-3 * s + 1e-9, k      # same kernels, different head/loss split
    optimizer.step()
    b.zero_grad()
    p_b = b.state_dict()
    worst = max(float((p_a[k].float() - p_b[k].float()).abs().max()) for k in p_a if p_a[k].is_floating_point())
    assert worst <= 2.1e-3                       # |Adam step 1| = lr per element; sign flips of ~0 gradients bound the gap
    close = np.mean([float(((p_a[k] - p_b[k]).abs() < 1e-5).float().mean()) for k in p_a if p_a[k].dim() >= 1 and p_a[k].is_floating_point()])
    assert close > 0.99
    assert all(float(v.abs().max()) == 0.0 for v in b.grads(keys={'out.weight'}).values())
    # a forward in between frees the graph: backward must fail loudly, not corrupt
    pred = b(Xd[:2])
    b.eval(); b(Xd[:2]); b.train()
    with pytest.raises(RuntimeError):
        (pred.sum()).backward()
    # two forwards, then backward through the FIRST: the handle holds the second one's graph -- must fail loudly,
    # while the second one's backward still works
    b.zero_grad()
    pred1 = b(Xd[:2])
    pred2 = b(Xd[2:4])
    with pytest.raises(RuntimeError):
        (pred1.sum()).backward()
    (pred2.sum()).backward()
    assert any(float(v.abs().max()) > 0 for v in b.grads(keys={'stg3_full_band_net.dec1.conv1.conv.0.weight'}).values())


def test_flat_parameter_follows_the_model_across_devices(vr):
    """ADVICE r2 + r3: the flat Parameter / .grad are zero-copy views of the native arenas.  Moving the model off the GPU closes the
    handle: an optimizer built before must not touch freed memory (the Parameter is emptied, with a warning), and -- like
    nn.Module.to, which keeps Parameter objects valid for existing optimizers -- the SAME Parameter is rebound to the next handle's
    arenas, so a torch optimizer built before the move keeps updating the model.  The native Adam's moments live in the closed
    handle: it raises instead of silently restarting."""
    from vocal_remover_amd import train as vtrain
    m, sd = _model(vr)
    m.train(); m.set_dropout_masks(None)
    old = m.parameters()[0]
    torch_opt = torch.optim.SGD([old], lr=1e-3)            # (stateless: torch's own optimizer state would not move with .to())
    native_opt = vtrain.Adam(m.parameters(), lr=1e-3)
    X, y = train_step.synth_batch(2, T=64, n_fft=N_FFT, seed=3)
    m.train_step(X.to(DEV), y.to(DEV), 1)
    assert old.numel() > 1000 and old.grad is not None and float(old.grad.abs().max()) > 0
    with pytest.warns(UserWarning, match='empty until the model moves back'):
        m.to('cpu')
    assert old.numel() == 0 and old.grad is None          # detached: nothing left that points into the freed arenas
    torch_opt.step()                                       # steps an empty tensor meanwhile: harmless
    m.to(torch.device(DEV))
    new = m.parameters()[0]
    assert new is old and new.numel() > 1000 and new.data_ptr() != 0      # the same object, rebound to the new arenas
    with pytest.raises(RuntimeError):
        native_opt.step()                                  # its moments lived in the closed handle
    # the optimizer built BEFORE the move still trains the model
    m.train(); m.set_dropout_masks(None)
    m.zero_grad()
    before = m.state_dict()['out.weight'].clone()
    m.train_step(X.to(DEV), y.to(DEV), 1)
    assert old.grad is not None and float(old.grad.abs().max()) > 0
    torch_opt.step()
    torch.cuda.synchronize()
    m.set_option('params_dirty', 1)
    m._host_stale = True
    assert not torch.equal(m.state_dict()['out.weight'], before)
    # and a fresh native optimizer works on the new handle
    opt = vtrain.Adam(m.parameters(), lr=1e-3)
    before = m.state_dict()['out.weight'].clone()
    m.train_step(X.to(DEV), y.to(DEV), 1)
    opt.step()
    assert not torch.equal(m.state_dict()['out.weight'], before)


def test_resample_kaiser_fast_vs_restatement_and_properties(vr):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3000)).astype(np.float32)
    for sr_in, sr_out in ((48000, 44100), (22050, 44100), (44100, 16000)):
        want = audio_np.resample_kaiser_fast(x, sr_in, sr_out)
        got = vr.audio.resample(x, sr_in, sr_out)
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max()), (sr_in, sr_out)
    # properties that pin the filter itself (resampy is not available to compare with): a 1 kHz tone keeps its
    # amplitude and frequency; DC gain is 1; a tone above the new Nyquist is attenuated
    sr_in, sr_out, n = 48000, 44100, 48000
    t = np.arange(n) / sr_in
    tone = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    y = vr.audio.resample(tone, sr_in, sr_out)
    tt = np.arange(len(y)) / sr_out
    ref = np.sin(2 * np.pi * 1000.0 * tt)
    core = slice(200, len(y) - 200)
    assert np.abs(y[core] - ref[core]).max() < 2e-3
    dc = vr.audio.resample(np.ones(n, np.float32), sr_in, sr_out)
    assert np.abs(dc[core] - 1.0).max() < 1e-3
    hi = np.sin(2 * np.pi * 23500.0 * t).astype(np.float32)          # above 22.05 kHz
    assert np.abs(vr.audio.resample(hi, sr_in, sr_out)[core]).max() < 0.05
    assert vr.audio.resample(tone, 44100, 44100) is not None and len(vr.audio.resample(tone, 48000, 44100)) == int(np.ceil(n * 44100 / 48000))


def test_align_wave_head_and_tail_lag_and_trim(vr):
    sr = 8000
    rng = np.random.default_rng(1)
    core = (rng.standard_normal((2, sr * 5)) * 0.3).astype(np.float32)
    a = np.concatenate([np.zeros((2, 700), np.float32), core, np.zeros((2, 900), np.float32)], axis=1)
    b = np.concatenate([np.zeros((2, 300), np.float32), core[:, 137:], np.zeros((2, 500), np.float32)], axis=1)
    a2, b2 = vr.spec_utils.align_wave_head_and_tail(a, b, sr)
    assert a2.shape == b2.shape
    n = min(a2.shape[1], sr * 4)
    assert np.abs(a2[:, 300:n] - b2[:, 300:n]).max() < 1e-6         # aligned wherever both carry signal (b starts 300 samples in)
    at, _ = vr.audio.trim(a)
    bt, _ = vr.audio.trim(b)
    am = at[:, :sr * 4].sum(axis=0); am = (am - am.mean()).astype(np.float32)
    bm = bt[:, :sr * 4].sum(axis=0); bm = (bm - bm.mean()).astype(np.float32)
    import ctypes
    best = ctypes.c_int64()
    vr.native.check(vr.native.lib().vr_xcorr_argmax(0, vr.native.np_ptr(am), len(am), vr.native.np_ptr(bm), len(bm), ctypes.byref(best)))
    assert int(best.value) - (len(am) - 1) == audio_np.align_head_and_tail_delay(am[:4000], bm[:4000]) or \
        int(best.value) - (len(am) - 1) == audio_np.align_head_and_tail_delay(am, bm)


def _write_song(vr, path, seconds, sr, seed, scale=0.3):
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    w = scale * rng.standard_normal((2, n)) * 0.2 + 0.2 * np.sin(2 * np.pi * (220.0 + 40 * seed) * t)[None]
    vr.audio.write(path, w.T.astype(np.float32), sr)
    return w.astype(np.float32)


def test_inference_main_wav_in_wav_out(vr, tmp_path):
    """inference.py main() end to end on the default net: 48 kHz WAV in (resampled to --sr), two 16-bit WAVs out."""
    sr = 44100
    wav = str(tmp_path / 'song.wav')
    w = _write_song(vr, wav, 1.5, 48000, 3)
    sd = weights.make_state_dict(1234)
    ckpt = str(tmp_path / 'model.pth')
    torch.save(sd, ckpt)
    out_dir = str(tmp_path / 'out')
    assert vr.inference.main(['--gpu', '0', '-P', ckpt, '-i', wav, '-o', out_dir, '--sr', str(sr)]) == 0
    yi, sr_i = vr.audio.read_wav(os.path.join(out_dir, 'song_Instruments.wav'))
    vi, sr_v = vr.audio.read_wav(os.path.join(out_dir, 'song_Vocals.wav'))
    assert sr_i == sr_v == sr and yi.shape == vi.shape and yi.shape[0] == 2
    # the same steps by hand
    X, got_sr = vr.audio.load(wav, sr=sr, mono=False, dtype=np.float32, res_type='kaiser_fast')
    assert got_sr == sr and abs(X.shape[1] - int(np.ceil(w.shape[1] * sr / 48000))) <= 1
    m = vr.nets.CascadedNet(2048, 1024, 32, 128)
    m.load_state_dict(sd)
    m.to(torch.device(DEV)).eval()
    y_wave, v_wave = vr.inference.Separator(m, torch.device(DEV), batchsize=4, cropsize=256).separate_wave(X)
    assert yi.shape == y_wave.shape
    q = 0.5 / 32768 + 1e-6                                                        # 16-bit PCM quantisation
    k = 32767 / 32768                                                             # written x * 0x7FFF, read q / 0x8000
    assert np.abs(yi - np.clip(y_wave, -1, 1) * k).max() <= q and np.abs(vi - np.clip(v_wave, -1, 1) * k).max() <= q
    assert np.abs(y_wave + v_wave - X[:, :y_wave.shape[1]]).max() < 1e-4


def test_dataset_preparation_fit_loop_and_checkpoint(vr, tmp_path, monkeypatch):
    """train.py main() in miniature: WAV pairs -> train_val_split -> cache_or_load (.npy cache in the reference's
    layout) -> training / validation sets -> two epochs of fit() with ReduceLROnPlateau -> best model + loss json +
    resumable checkpoint; resuming reproduces the third epoch of an uninterrupted run."""
    from vocal_remover_amd import train as vtrain
    monkeypatch.chdir(tmp_path)
    sr, hop = 8000, N_FFT // 2
    for sub in ('mixtures', 'instruments'):
        os.makedirs(os.path.join('data', sub))
    for i in range(3):
        w = _write_song(vr, 'data/instruments/s%d.wav' % i, 6.0, sr, 10 + i)
        voc = 0.1 * np.sin(2 * np.pi * 700.0 * np.arange(w.shape[1]) / sr)[None].astype(np.float32)
        vr.audio.write('data/mixtures/s%d.wav' % i, (w + voc).T, sr)
    import random
    random.seed(0)
    train_fl, val_fl = vr.dataset.train_val_split('data', 'random', 0.34, [])
    assert len(train_fl) == 2 and len(val_fl) == 1
    m, sd = _model(vr)
    training_set = vr.dataset.make_training_set(train_fl, sr, hop, N_FFT)
    assert os.path.exists(training_set[0][0]) and np.load(training_set[0][0]).ndim == 3           # [T, 2, bins] cache
    bins = N_FFT // 2 + 1
    rw = np.zeros((bins, 1), np.float32)
    ds = vr.dataset.VocalRemoverTrainingSet(training_set * 4, cropsize=160, reduction_rate=0.0, reduction_weight=rw,
                                            mixup_rate=0.0, mixup_alpha=1.0, model=m)
    patches = vr.dataset.make_validation_set(val_fl, 160, sr, hop, N_FFT, m.offset)
    assert len(patches) >= 1 and os.path.dirname(patches[0]) == 'cs160_sr%d_hl%d_nf%d_of64' % (sr, hop, N_FFT)
    vds = vr.dataset.VocalRemoverValidationSet(patches, model=m)

    def run(epochs, resume_from=None, tag='a'):
        mm, _ = _model(vr)
        mm.set_dropout_masks(None)
        np.random.seed(5)
        dsx = vr.dataset.VocalRemoverTrainingSet(training_set * 4, 160, 0.0, rw, 0.0, 1.0, model=mm)
        tl = vr.dataset.DeviceLoader(dsx, batch_size=4, shuffle=False)
        vl = vr.dataset.DeviceLoader(vr.dataset.VocalRemoverValidationSet(patches, model=mm), batch_size=2, shuffle=False)
        opt = vtrain.Adam(mm.parameters(), lr=1e-3)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.9, patience=6, threshold=1e-6, min_lr=1e-4)
        start, best, log = 0, None, None
        if resume_from:
            start, best, log = vtrain.load_checkpoint(resume_from, mm, opt, sched)
            np.random.seed(5)
            for _ in range(start * len(dsx)):           # the data order is a function of numpy's stream: replay it
                dsx.plan(0)
        log, best = vtrain.fit(mm, torch.device(DEV), tl, vl, opt, sched, epochs, 1, model_dir='models_' + tag,
                               log_path='loss_%s.json' % tag, checkpoint_path='ckpt_%s.pt' % tag, start_epoch=start,
                               best_loss=best, log=log)
        return mm, log

    m3, log3 = run(3, tag='full')
    assert len(log3) == 3 and json.load(open('loss_full.json')) == log3
    saved = sorted(os.listdir('models_full'))
    assert 'model_iter0.pth' in saved
    sd0 = torch.load(os.path.join('models_full', saved[-1]), map_location='cpu')
    assert set(sd0) == set(sd)                                                  # the reference's 689-key state dict
    m2, log2 = run(2, tag='part')
    m2b, log2b = run(3, resume_from='ckpt_part.pt', tag='resumed')
    assert len(log2b) == 3 and log2b[:2] == log2
    assert abs(log2b[2][0] - log3[2][0]) < 2e-4 and abs(log2b[2][1] - log3[2][1]) < 2e-4    # epoch 3 after a resume == uninterrupted
    assert len(ds) == 8 and len(vds) == len(patches)
