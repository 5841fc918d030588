"""GPU debug helper: Winograd wgrad vs torch on tiny shapes (tools only)."""
import ctypes, os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
vr = __graft_entry__.load_package()
nat = vr.native
m = vr.nets.CascadedNet(512, 256, 8, 32); m.to(torch.device('cuda:0'))
def run(N, Cin, H, W, Cout, seed=0, pattern=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_(True)
    xin = x.clone().requires_grad_(True)
    out = F.conv2d(xin, w, None, 1, 1)
    dz = torch.randn(out.shape, generator=g)
    if pattern == 'delta':
        dz = torch.zeros_like(dz); dz[0, 0, 0, 0] = 1.0
    out.backward(dz)
    dx = np.empty(tuple(x.shape), np.float32); dwt = np.empty(tuple(w.shape), np.float32)
    nat.check(nat.lib().vr_debug_conv2d_backward(m._handle.h, nat.np_ptr(x.numpy()), N, Cin, H, W, nat.np_ptr(w.detach().numpy()), Cout, 3, 1, 1, 1, 0,
                                                 None, ctypes.c_float(1.0), nat.np_ptr(dz.numpy()), nat.np_ptr(dx), nat.np_ptr(dwt)))
    want = w.grad.numpy()
    err = np.abs(dwt - want)
    print('N%d Cin%d H%d W%d Cout%d %s: wgrad err %.3e (scale %.3e)  dgrad err %.3e' % (N, Cin, H, W, Cout, pattern or '', err.max(), np.abs(want).max(),
          np.abs(dx - xin.grad.numpy()).max()))
    if err.max() > 1e-3 * np.abs(want).max():
        e = err.reshape(Cout, Cin, 9)
        print('  err by tap', e.max(axis=(0, 1)).round(4))
        print('  err by ci ', e.max(axis=(0, 2)).round(3)[:16])
        print('  err by co ', e.max(axis=(1, 2)).round(3)[:16])
        print('  got/want [0,0]', dwt[0, 0].round(3).tolist(), want[0, 0].round(3).tolist())
for args in [(1, 8, 2, 16, 32), (1, 8, 2, 32, 32), (1, 8, 4, 16, 32), (2, 8, 2, 16, 32), (1, 8, 16, 32, 32), (2, 8, 16, 32, 32), (1, 32, 2, 16, 64), (1, 8, 2, 16, 32, 0, 'delta')]:
    run(*args)
