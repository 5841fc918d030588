"""Launcher: run one of the reference's own scripts, UNCHANGED, on the MI355X path.

    python /path/to/this/repo/vocal-remover_amd/run.py /path/to/vocal-remover/inference.py --input song.wav --gpu 0 [--tta]

Why a launcher and not PYTHONPATH: Python puts the directory of the script it runs at sys.path[0], AHEAD of PYTHONPATH, so
`from lib import nets` inside the reference's inference.py (inference.py:10-13) always finds the reference's own `lib/` first.
This file runs the script through `runpy` with `dropin/` in front of the script's directory instead:

    from lib import nets / spec_utils / dataset   -> dropin/lib/*.py       -> vocal_remover_amd.* (libvr_mi355.so)
    from lib import utils / layers / anything else -> the script's own lib/ (dropin/lib/__init__.py appends it to lib.__path__)
    import inference (pseudo.py:9)                 -> dropin/inference.py   -> vocal_remover_amd.inference.Separator

The script itself is executed from its own file: its argparse, its `Separator` class (inference.py:16-102), its `train_epoch`
(train.py:68-105) are the reference's statements; only the names they import resolve to this package.

librosa / soundfile / cv2 are imported at module level by the reference (inference.py:4-6, lib/utils.py:3) although the hot
path reaches librosa only through `spec_utils` (shadowed).  Where one of them is NOT installed, a stand-in module is registered
so that the import succeeds: librosa.load / librosa.effects.trim / soundfile.write / soundfile.read are backed by
vocal_remover_amd.audio (RIFF/WAVE only), everything else raises on use.  An installed librosa / soundfile / cv2 is never replaced.
"""
import importlib.util
import os
import runpy
import sys
import types

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
DROPIN = os.path.join(_PKG_DIR, 'dropin')


def _package():
    if _ROOT not in sys.path:
        sys.path.append(_ROOT)          # for `import __graft_entry__`; appended: it must never shadow the script's modules
    import __graft_entry__
    return __graft_entry__.load_package()


def _missing(name):
    if name in sys.modules:
        return False
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


class _Absent(types.ModuleType):
    """Module stand-in whose every attribute access explains what is missing."""

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        raise ImportError('%s.%s: %s is not installed and vocal_remover_amd.run only stands in for the calls the hot path makes'
                          % (self.__name__, item, self.__name__))


def install_standins(verbose=True):
    """Register stand-ins for librosa / soundfile / cv2 where they are not installed.  -> names installed."""
    done = []
    if _missing('librosa'):
        audio = _package().audio
        lb = _Absent('librosa')
        lb.load = audio.load                                        # inference.py:136-138, lib/spec_utils.py:139-142
        lb.resample = lambda y, orig_sr, target_sr, res_type='kaiser_fast', **kw: audio.resample(y, orig_sr, target_sr, res_type)
        fx = _Absent('librosa.effects')
        fx.trim = audio.trim                                        # lib/spec_utils.py:97-98
        lb.effects = fx
        sys.modules['librosa'], sys.modules['librosa.effects'] = lb, fx
        done.append('librosa')
    if _missing('soundfile'):
        audio = _package().audio
        sf = _Absent('soundfile')
        sf.write = lambda file, data, samplerate, *a, **kw: audio.write(file, data, samplerate)   # inference.py:173,178

        def _read(file, dtype='float64', always_2d=False, **kw):
            x, sr = audio.read_wav(file)
            x = x.T.astype(dtype)
            return (x if (always_2d or x.shape[1] > 1) else x[:, 0]), sr
        sf.read = _read
        sys.modules['soundfile'] = sf
        done.append('soundfile')
    if _missing('cv2'):
        cv = _Absent('cv2')
        cv.IMREAD_COLOR = 1                                         # default argument of lib/utils.py:7
        sys.modules['cv2'] = cv
        done.append('cv2')
    if done and verbose:
        sys.stderr.write('[vocal_remover_amd.run] not installed, stand-ins registered: %s\n' % ', '.join(done))
    return done


def prepare(script):
    """Arrange sys.path the way `python script` would, with dropin/ in front of the script's directory."""
    script = os.path.abspath(script)
    if not os.path.isfile(script):
        raise FileNotFoundError(script)
    script_dir = os.path.dirname(script)
    # what `python run.py` (or a caller) left at the front must not shadow the script's imports: this package's own directory
    # holds an inference.py / train.py / dataset.py of its own
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (_PKG_DIR, DROPIN, script_dir)]
    sys.path.insert(0, script_dir)
    sys.path.insert(0, DROPIN)
    for name in [m for m in sys.modules if m == 'lib' or m.startswith('lib.') or m == 'inference']:
        del sys.modules[name]           # a `lib` imported earlier in this process (tests) would win over both
    _package()
    install_standins()
    return script


def run(script, argv=(), run_name='__main__'):
    """Execute `script` (a path to the reference's inference.py / train.py / pseudo.py) with sys.argv = [script] + argv.
    -> the script's globals (runpy.run_path), e.g. run(..., run_name='ref_train')['train_epoch'] without running main()."""
    script = prepare(script)
    saved = sys.argv
    sys.argv = [script] + list(argv)
    try:
        return runpy.run_path(script, run_name=run_name)
    finally:
        sys.argv = saved


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        sys.stderr.write(__doc__)
        return 2
    run(argv[0], argv[1:])
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
