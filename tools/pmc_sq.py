"""Per-dispatch SQ counters for the conv kernels of the last step (largest launches first)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events order by dispatch_id").fetchall()
d = {}
for did, name, cn, cv, dur in rows:
    if 'conv_mfma' not in name and 'wgrad_mfma' not in name and 'conv_ws' not in name: continue
    e = d.setdefault(did, {'name': re.search(r'<([^>]*)>', name).group(1).replace(' ', ''), 'dur': dur})
    e[cn] = cv
items = sorted(d.values(), key=lambda e: -e['dur'])[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]
keys = sorted({k for e in items for k in e if k not in ('name', 'dur')})
print('variant dur_us ' + ' '.join(keys))
for e in items:
    print(e['name'], '%.0f' % (e['dur'] / 1e3), ' '.join('%.3g' % e.get(k, float('nan')) for k in keys))
