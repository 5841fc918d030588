"""Build-time audit of the kernels that keep inline-asm loads in flight across hand-placed `s_waitcnt vmcnt(N)`: hipcc treats
an asm load's destination as written when the asm statement ends, so nothing between a load and the wait that covers it may read or
overwrite that register.  tools/asm_inflight_audit.py scans the generated ISA linearly (LDS-DMA loads and stores take vmcnt slots with an
empty register set); it must report nothing for `wgrad_wino_r_kernel` (the default weight gradient, all four instantiations) and for the
plain TH = 8 tilings of `conv_x3h_kernel`.  (The 32 x 16 tiling and the fused-upsample forms show reports from the rare source-switch
path that reuse a load's destination as a temporary BEFORE that load is issued -- the linear scan cannot tell; see the tool's docstring.)
No GPU: hipcc cross-compiles to assembly (~1 min)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'vocal-remover_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
AUDIT = os.path.join(ROOT, 'tools', 'asm_inflight_audit.py')
AUDIT2 = os.path.join(ROOT, 'tools', 'asm_inflight_audit2.py')


def _asm(tmp_path, src):
    out = str(tmp_path / (src + '.s'))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def _audit(asm, kernel):
    r = subprocess.run([sys.executable, AUDIT, asm, kernel], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for ln in r.stdout.splitlines():
        if ln.rstrip().endswith('violations') and ' register loads, ' in ln:
            name, rest = ln.split(': ', 1)
            rows[name] = (int(rest.split(' register loads, ')[0]), int(rest.split(' register loads, ')[1].split(' ')[0]))
    return rows, r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_instruction_touches_an_in_flight_asm_load(tmp_path):
    asm = _asm(tmp_path, 'wgrad_wino.hip')
    rows, out = _audit(asm, 'wgrad_wino_r_kernel')
    assert len(rows) == 4, out[-2000:]
    for name, (loads, bad) in rows.items():
        assert loads >= 32 and bad == 0, (name, loads, bad, out[-2000:])
    asm = _asm(tmp_path, 'conv_x3h.hip')
    for inst in ('ILi64ELi8ELb0ELb0ELb0', 'ILi32ELi8ELb0ELb0ELb0'):          # <64, 8> and <32, 8>, plain, default occupancy, no trace
        rows, out = _audit(asm, 'conv_x3h_kernel' + inst)
        assert len(rows) == 1, out[-2000:]
        (name, (loads, bad)), = rows.items()
        assert loads >= 100 and bad == 0, (name, loads, bad, out[-2000:])
        # round 6: the second audit follows the registers by NAME (x3h_wait8 lists the registers it releases in a `; landed` comment), no
        # vmcnt arithmetic: it is the tool that caught hipcc copying in-flight registers of the parked ping-pong kernel
        r = subprocess.run([sys.executable, AUDIT2, asm, 'conv_x3h_kernel' + inst], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and ' 0 reports' in r.stdout, r.stdout[-2000:]
    # conv_x3d.hip (round 6): the same loads and waits behind a dilated halo tile; every single-conv instantiation.  (The ASPP kernel holds
    # four of these bodies behind a branch on blockIdx.y: the linear scan concatenates mutually exclusive paths and reports across them.)
    asm = _asm(tmp_path, 'conv_x3d.hip')
    rows, out = _audit(asm, 'conv_x3d_kernel')
    assert len(rows) == 10, out[-2000:]
    for name, (loads, bad) in rows.items():
        assert loads >= 100 and bad == 0, (name, loads, bad, out[-2000:])
    r = subprocess.run([sys.executable, AUDIT2, asm, 'conv_x3d_kernel'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count(' 0 reports') == 10, r.stdout[-2000:]
