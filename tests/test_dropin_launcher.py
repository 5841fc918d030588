"""Row (b), "inference.py drops in unchanged": the launcher vocal-remover_amd/run.py runs a script of the reference with
`from lib import nets, spec_utils, dataset` resolved to the MI355X package and everything else to the checkout's own lib/.

CPU: the reference's REAL inference.py / train.py (build container only) and the reference-shaped tests/dropin_case script are
executed through the launcher up to model construction; which modules their imports resolved to is asserted.
GPU: tests/dropin_case/separate_script.py -- the reference's own Separator call sequence (inference.py:26-102) -- drives the
native CascadedNet and must reproduce the outputs of the REFERENCE committed in tests/golden/reference_outputs.npz."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, 'vocal-remover_amd', 'run.py')
CASE = os.path.join(ROOT, 'tests', 'dropin_case', 'separate_script.py')
REFERENCE = '/root/reference'
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_outputs.npz'))
SMALL = ['--n_fft', '512', '--hop_length', '256', '--nout', '8', '--nout_lstm', '32']


def _env():
    env = dict(os.environ)
    env.pop('PYTHONPATH', None)                  # the launcher must not depend on it
    return env


def _launch(args, cwd, expect_ok=True):
    r = subprocess.run([sys.executable, RUN] + args, capture_output=True, text=True, cwd=cwd, env=_env(), timeout=900)
    if expect_ok:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def _small_model_file(tmp_path):
    p = str(tmp_path / 'small.pth')
    torch.save(weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32), p)
    return p


def test_reference_shaped_script_resolves_to_the_native_package(tmp_path):
    out = str(tmp_path / 'probe.npz')
    _launch([CASE, '-P', _small_model_file(tmp_path), '-i', 'unused', '-o', out, '--probe'] + SMALL, cwd=str(tmp_path))
    got = np.load(out)
    assert os.path.samefile(str(got['nets']), os.path.join(ROOT, 'vocal-remover_amd', 'dropin', 'lib', 'nets.py'))
    assert str(got['model_class']) == 'vocal_remover_amd.nets'
    assert str(got['utils']) == 'checkout'                            # not shadowed: the script's own lib/utils.py
    assert str(got['image']) == 'checkout spectrogram_to_image'       # a name outside the hot path falls through
    assert 'dropin' in str(got['spec_utils']) and 'dropin' in str(got['dataset'])


PROBE = r'''
import importlib.util, json, os, sys
spec = importlib.util.spec_from_file_location('vr_run', sys.argv[1]); run = importlib.util.module_from_spec(spec); spec.loader.exec_module(run)
g = run.run(sys.argv[2], run_name='reference_script_probe')           # module body only: main() is guarded by __name__
import vocal_remover_amd as vr
import lib.utils
res = {'nets': g['nets'].CascadedNet is vr.nets.CascadedNet,
       'stft': g['spec_utils'].wave_to_spectrogram is vr.spec_utils.wave_to_spectrogram,
       'istft': g['spec_utils'].spectrogram_to_wave is vr.spec_utils.spectrogram_to_wave,
       'padding': g['dataset'].make_padding is vr.dataset.make_padding,
       'trainset': g['dataset'].VocalRemoverTrainingSet is vr.dataset.VocalRemoverTrainingSet,
       'utils_file': lib.utils.__file__,
       'image_module': g['spec_utils'].spectrogram_to_image.__module__,
       'has': sorted(k for k in ('Separator', 'train_epoch', 'validate_epoch', 'main') if k in g)}
if 'Separator' in g:                                                   # inference.py:130-132,150 up to the first device call
    import torch
    model = g['nets'].CascadedNet(512, 256, 8, 32)
    model.load_state_dict(torch.load(sys.argv[3], map_location='cpu'))
    model.to(torch.device('cpu'))
    sp = g['Separator'](model=model, device=torch.device('cpu'), batchsize=4, cropsize=160, postprocess=False)
    res['separator_model'] = type(sp.model).__module__
    res['offset'] = sp.offset
print('PROBE' + json.dumps(res))
'''


@pytest.mark.parametrize('script', ['inference.py', 'train.py'])
def test_reference_scripts_bind_to_the_native_package(script, tmp_path):
    """The reference's own files, unchanged, through the launcher (runpy): every hot-path name they import is this package's."""
    path = os.path.join(REFERENCE, script)
    if not os.path.isfile(path):
        pytest.skip('/root/reference not present (GPU box)')
    r = subprocess.run([sys.executable, '-c', PROBE, RUN, path, _small_model_file(tmp_path)], capture_output=True, text=True,
                       cwd=str(tmp_path), env=_env(), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('PROBE')][-1][5:])
    assert res['nets'] and res['stft'] and res['istft'] and res['padding'] and res['trainset'], res
    assert os.path.samefile(res['utils_file'], os.path.join(REFERENCE, 'lib', 'utils.py'))        # pass-through, not a copy
    assert res['image_module'] == 'lib._reference_spec_utils'
    if script == 'inference.py':
        assert res['has'] == ['Separator', 'main'] and res['separator_model'] == 'vocal_remover_amd.nets' and res['offset'] == 64
    else:
        assert res['has'] == ['main', 'train_epoch', 'validate_epoch']


def test_reference_inference_cli_runs_to_the_first_device_call(tmp_path, vr):
    """`run.py /root/reference/inference.py --input x.wav ...` as a command: argparse, model construction, load_state_dict, .to(),
    librosa.load all succeed; without a GPU the first compute call (spec_utils.wave_to_spectrogram -> vr_stft) raises -- there
    is no CPU fallback to fall into."""
    path = os.path.join(REFERENCE, 'inference.py')
    if not os.path.isfile(path):
        pytest.skip('/root/reference not present (GPU box)')
    if torch.cuda.is_available():
        pytest.skip('CPU-only check')
    wav = str(tmp_path / 'song.wav')
    rng = np.random.default_rng(0)
    vr.audio.write(wav, 0.1 * rng.standard_normal((44100, 2)).astype(np.float32), 44100)
    pth = str(tmp_path / 'full.pth')
    torch.save(weights.make_state_dict(3), pth)
    r = _launch([path, '--input', wav, '--pretrained_model', pth, '--gpu', '-1', '--output_dir', str(tmp_path / 'out')],
                cwd=str(tmp_path), expect_ok=False)
    assert 'loading model... done' in r.stdout and 'loading wave source... done' in r.stdout, r.stdout + r.stderr[-3000:]
    assert 'stft of wave source...' in r.stdout and 'inverse stft' not in r.stdout
    assert r.returncode != 0 and 'libvr_mi355' in r.stderr, r.stderr[-3000:]


def _run_case(tmp_path, extra):
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((2, 257, 300)) + 1j * rng.standard_normal((2, 257, 300))).astype(np.complex64)
    xin = str(tmp_path / 'X.npy')
    np.save(xin, X)
    out = str(tmp_path / 'out.npz')
    _launch([CASE, '-P', _small_model_file(tmp_path), '-i', xin, '-o', out, '--gpu', '0', '-B', '2', '-c', '160'] + SMALL + extra,
            cwd=str(tmp_path))
    return X, np.load(out)


@pytest.mark.gpu
def test_reference_separator_loop_over_the_native_model(tmp_path):
    """inference.py:42-81 (the reference's crop loop, batch by batch, complex64 cuda tensor -> torch.abs -> predict_mask ->
    .detach().cpu().numpy()) over the native CascadedNet == the reference's committed outputs."""
    X, got = _run_case(tmp_path, [])
    scale = np.abs(X).max()
    assert np.abs(got['y_spec'][:, ::5] - G['sep_y']).max() < 1e-4 * scale
    assert np.abs(got['v_spec'][:, ::5] - G['sep_v']).max() < 1e-4 * scale


@pytest.mark.gpu
def test_reference_separator_tta_and_postprocess_over_the_native_model(tmp_path, vr):
    X, got = _run_case(tmp_path, ['--tta'])
    scale = np.abs(X).max()
    assert np.abs(got['y_spec'][:, ::5] - G['sep_tta_y']).max() < 1e-4 * scale
    # --postprocess through the reference's own _postprocess (spec_utils.merge_artifacts of this package: host half) against
    # this package's device-side Separator(postprocess=True)
    model = vr.nets.CascadedNet(512, 256, 8, 32)
    model.load_state_dict(weights.make_state_dict(11, n_fft=512, nout=8, nout_lstm=32))
    model.to(torch.device('cuda:0')).eval()
    try:
        y, v = vr.inference.Separator(model, torch.device('cuda:0'), batchsize=2, cropsize=160, postprocess=True).separate(X.copy())
    except IndexError:
        pytest.skip('random-weight mask has no frame above the threshold')
    X, got = _run_case(tmp_path, ['--postprocess'])
    assert np.abs(got['y_spec'] - y).max() < 1e-4 * scale and np.abs(got['v_spec'] - v).max() < 1e-4 * scale


@pytest.mark.gpu
def test_launcher_wav_in_wav_out(tmp_path, vr):
    """The stand-in librosa.load / soundfile.write + wave_to_spectrogram / spectrogram_to_wave of the shadow, end to end."""
    rng = np.random.default_rng(1)
    wav = str(tmp_path / 'song.wav')
    t = np.arange(44100 * 2) / 44100.0
    wave = (0.05 * rng.standard_normal((2, t.size)) + 0.2 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    vr.audio.write(wav, wave.T, 44100)
    out = str(tmp_path / 'o')
    _launch([CASE, '-P', _small_model_file(tmp_path), '-i', wav, '-o', out, '--gpu', '0', '-B', '4', '-c', '160'] + SMALL,
            cwd=str(tmp_path))
    got = np.load(out + '.npz')
    back, sr = vr.audio.read_wav(out + '_Instruments.wav')
    assert sr == 44100 and back.shape == got['y_wave'].shape
    assert np.abs(back - np.clip(got['y_wave'], -1, 1)).max() < 2.0 / 32768
    assert np.isfinite(got['y_wave']).all() and np.abs(got['y_wave']).max() > 1e-3
