"""Overlap picture of one step from a rocprofv3 kernel trace taken WITH the concurrent executor (rocpd database):
per-queue busy time, the union (some kernel running), the sum of kernel durations and the largest idle gaps.  The step is the
span between two launches of a marker kernel: adam_kernel (train step, default) or e.g. stft_tile_kernel (inference: the first
kernel of a separate_wave call).   Usage: timeline.py run.db [steps back from the end] [marker] [end|start]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute('select name, start, end, %s from kernels order by start' % (qcol or '0')).fetchall()
marker = sys.argv[3] if len(sys.argv) > 3 else 'adam_kernel'
at_start = len(sys.argv) > 4 and sys.argv[4] == 'start'      # the marker OPENS a step (inference) instead of closing it (train)
adam = [i for i, r in enumerate(rows) if marker in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1: the last step (bench.py: the serialised, profiled one), 2: the one before
if len(adam) >= back + 1:
    rows = rows[adam[-back - 1]:adam[-back]] if at_start else rows[adam[-back - 1] + 1:adam[-back] + 1]
t0, t1 = rows[0][1], max(r[2] for r in rows)
print('kernels %d, span %.2f ms, sum of durations %.2f ms' % (len(rows), (t1 - t0) / 1e6, sum(r[2] - r[1] for r in rows) / 1e6))
byq = {}
for name, s, e, q in rows:
    byq.setdefault(q, []).append((s, e, name))
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    print('  queue %s: %4d kernels, busy %.2f ms, first at %.2f ms, last end %.2f ms' % (q, len(lst), sum(e - s for s, e, _ in lst) / 1e6,
                                                                                    (lst[0][0] - t0) / 1e6, (max(e for _, e, _ in lst) - t0) / 1e6))
# union of busy intervals
iv = sorted((s, e) for _, s, e, _ in rows)
union, gaps, cs, ce = 0, [], iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs
        gaps.append((s - ce, ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print('some kernel running %.2f ms; idle %.2f ms in %d gaps; gaps > 20 us: %d totalling %.2f ms' % (
    union / 1e6, (t1 - t0 - union) / 1e6, len(gaps), sum(1 for g, _ in gaps if g > 20000), sum(g for g, _ in gaps if g > 20000) / 1e6))
# time with exactly k kernels running
ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
conc, prev, k = {}, ev[0][0], 0
for t, d in ev:
    conc[k] = conc.get(k, 0) + (t - prev)
    prev, k = t, k + d
print('concurrency histogram (ms):', ' '.join('%d:%.2f' % (k, v / 1e6) for k, v in sorted(conc.items()) if v > 0))
# per family busy on the step
fam = {}
for name, s, e, q in rows:
    n = re.sub(r'^void ', '', name)
    n = re.sub(r'[<(].*$', '', n)
    fam[n] = fam.get(n, 0) + (e - s)
print('by kernel family (ms):', ', '.join('%s %.2f' % (k.replace('vr::', ''), v / 1e6) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:24]))
