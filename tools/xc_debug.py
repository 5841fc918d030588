import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
vr = __graft_entry__.load_package()
rng = np.random.default_rng(1)
for na, nb in ((1000, 1000), (32000, 32000), (5000, 3000)):
    a = rng.standard_normal(na).astype(np.float32); b = np.roll(a, -37)[:nb].copy() if nb <= na else rng.standard_normal(nb).astype(np.float32)
    best = ctypes.c_int64(-5)
    rc = vr.native.lib().vr_xcorr_argmax(0, vr.native.np_ptr(a), na, vr.native.np_ptr(b), nb, ctypes.byref(best))
    want = int(np.argmax(np.correlate(a, b, 'full')))
    print(na, nb, 'rc', rc, 'got', best.value, 'want', want)
