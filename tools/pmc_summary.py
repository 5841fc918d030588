"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in KB).
Usage: pmc_summary.py fetch.db write.db [steps_in_pass out.json [mode]]  -- the JSON is what bench.py reads for
`roofline.traffic` (conv family, FETCH_SIZE x2 = the gfx950 correction of MI355X_MICROARCH.md).
mode = infer (default) | tta | train: names the bench.py command the passes ran."""
import json
import re
import sqlite3
import sys

FAMILY = 'conv family (conv_x3 + conv_wino + conv_dma + conv_thin + conv_ws + conv_mfma + wgrad)'
KEYS = ('conv_x3_kernel', 'conv_x3h_kernel', 'conv_x3d_kernel', 'conv_x3d_aspp_kernel', 'conv_x3b_kernel', 'conv_x3p_kernel', 'conv_mfma_kernel', 'conv_ws_kernel', 'conv_dma_kernel', 'conv_dma_s2d_kernel',
        'conv_thin_kernel', 'conv_wino_kernel', 'wgrad_ws_kernel', 'wgrad_mfma_kernel', 'wgrad_wino_kernel', 'wgrad_wino_r_kernel', 'wgrad_gemm_kernel')


MEMBERS = {}          # counter -> {kernel name inside the conv family: (launches, value)}


def load(path, counter):
    db = sqlite3.connect(path)
    out = {}
    mem = MEMBERS.setdefault(counter, {})
    for name, val in db.execute("select name, counter_value from pmc_events where counter_name=? order by start", (counter,)):
        name = re.sub(r'^void ', '', name)
        conv = any(k in name for k in KEYS)
        if conv:
            short = re.sub(r'\(.*$', '', name)[:70]
            n, s = mem.get(short, (0, 0.0))
            mem[short] = (n + 1, s + val)
        name = FAMILY if conv else re.sub(r'\(.*$', '', name)[:60]
        n, s = out.get(name, (0, 0.0))
        out[name] = (n + 1, s + val)
    return out


f = load(sys.argv[1], 'FETCH_SIZE')
w = load(sys.argv[2], 'WRITE_SIZE')
print('| kernel | launches | FETCH_SIZE MB (raw) | WRITE_SIZE MB (raw) |')
print('|---|---|---|---|')
for k in sorted(f, key=lambda k: -f[k][1]):
    print('| %s | %d | %.1f | %.1f |' % (k, f[k][0], f[k][1] / 1024, w.get(k, (0, 0))[1] / 1024))
print()
print('| member of the conv family | launches | FETCH_SIZE MB (raw; x2 = bytes) | WRITE_SIZE MB (raw) |')
print('|---|---|---|---|')
for k in sorted(MEMBERS['FETCH_SIZE'], key=lambda k: -MEMBERS['FETCH_SIZE'][k][1]):
    print('| %s | %d | %.1f | %.1f |' % (k, MEMBERS['FETCH_SIZE'][k][0], MEMBERS['FETCH_SIZE'][k][1] / 1024, MEMBERS.get('WRITE_SIZE', {}).get(k, (0, 0))[1] / 1024))
if len(sys.argv) > 4:
    mode = sys.argv[5] if len(sys.argv) > 5 else 'infer'
    n, fk = f[FAMILY]
    wk = w[FAMILY][1]
    json.dump({
        'command': 'VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py '
                   '--mode %s --steps 1 --warmup 0 --no-cpu-baseline (two separate passes, single stream so kernels do not overlap)' % mode,
        'kernel': FAMILY, 'launches_in_pass': n, 'steps_in_pass': int(sys.argv[3]),
        'fetch_size_kb_raw': fk, 'write_size_kb_raw': wk, 'fetch_correction': 2.0,
        'note': 'gfx950 FETCH_SIZE reads 1/2 of streamed bytes (MI355X_MICROARCH.md, HBM section); calibrated in round 1 '
                'on kernels with known traffic (thin_conv_kernel<2,true>: 369.1 MB algorithmic vs 176.0 MB '
                'reported = 0.477; WRITE_SIZE matches: 11.5 MB vs 11.0 MB).',
        'bytes_per_launch': (2.0 * fk + wk) * 1024.0 / n,
    }, open(sys.argv[4], 'w'), indent=1)
