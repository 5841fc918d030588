"""Kernel names of a rocprofv3 kernel trace (rocpd database) in start order, run-length compressed: `name xN (total us)`.
Usage: kernel_sequence.py run.db [first [count]]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = db.execute('select name, start, duration from kernels order by start').fetchall()[first:first + count]
out, prev, n, tot = [], None, 0, 0.0
for name, start, dur in rows:
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)[:70]
    if name == prev:
        n += 1; tot += dur
    else:
        if prev is not None:
            out.append('%s x%d (%.1f us)' % (prev, n, tot / 1e3))
        prev, n, tot = name, 1, dur
out.append('%s x%d (%.1f us)' % (prev, n, tot / 1e3))
print('\n'.join(out))
