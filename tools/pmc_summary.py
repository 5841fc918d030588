"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in KB)."""
import re
import sqlite3
import sys


def load(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, val in db.execute("select name, counter_value from pmc_events where counter_name=? order by start", (counter,)):
        name = re.sub(r'^void ', '', name)
        name = 'conv family (conv_ws + conv_mfma)' if ('conv_mfma_kernel' in name or 'conv_ws_kernel' in name) else re.sub(r'\(.*$', '', name)[:60]
        n, s = out.get(name, (0, 0.0))
        out[name] = (n + 1, s + val)
    return out


f = load(sys.argv[1], 'FETCH_SIZE')
w = load(sys.argv[2], 'WRITE_SIZE')
print('| kernel | launches | FETCH_SIZE MB (raw) | WRITE_SIZE MB (raw) |')
print('|---|---|---|---|')
for k in sorted(f, key=lambda k: -f[k][1]):
    print('| %s | %d | %.1f | %.1f |' % (k, f[k][0], f[k][1] / 1024, w.get(k, (0, 0))[1] / 1024))
