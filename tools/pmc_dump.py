"""Per-kernel means of every counter of one rocprofv3 --pmc pass (rocpd database), normalised where that helps:
wave-state counters as a fraction of SQ_WAVE_CYCLES, byte counters per launch, GRBM_GUI_ACTIVE per ns (= effective GHz).
Usage: pmc_dump.py run.db [name-filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
per = {}
for did, name, cn, cv, dur in db.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events"):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    if flt and flt not in name:
        continue
    e = per.setdefault(did, {'name': name, 'dur': dur})
    e[cn] = e.get(cn, 0.0) + cv
groups = {}
order = []
for did in sorted(per):
    e = per[did]
    key = e['name'][:60]
    if key not in groups:
        order.append(key)
    g = groups.setdefault(key, {'n': 0})
    g['n'] += 1
    for k, v in e.items():
        if k != 'name':
            g[k] = g.get(k, 0.0) + v
for key in order:
    g = groups[key]
    n = g['n']
    wc = g.get('SQ_WAVE_CYCLES', 0.0)
    parts = ['%-60s n=%-3d %8.1f us' % (key, n, g['dur'] / n / 1e3)]
    for k in sorted(g):
        if k in ('n', 'dur'):
            continue
        v = g[k]
        if k == 'GRBM_GUI_ACTIVE':
            parts.append('clk %.2f GHz' % (v / g['dur']))
        elif k.startswith('SQ_') and wc and k != 'SQ_WAVE_CYCLES':
            parts.append('%s %.3f' % (k[3:], v / wc))
        elif k in ('FETCH_SIZE', 'WRITE_SIZE'):
            parts.append('%s %.1f MB' % (k, v / n * 1024 / 1e6))       # (KB units)
        else:
            parts.append('%s %.4g' % (k, v / n))
    print(' | '.join(parts))
