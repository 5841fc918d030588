"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as the --stats table:
per kernel: calls, total ms, average us, % of GPU kernel time.  Usage: rocpd_summary.py db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'vr::conv_mfma_kernel<([^>]*)>', name)
    if m:
        return 'conv_mfma<KS,S,DH,DW,MT,TH,TW,CK,WM=%s>' % m.group(1).replace(' ', '')
    return re.sub(r'\(.*$', '', name)[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                      'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for name, n, tot, avg, mn, mx in rows:
        lines.append('| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f |' % (short(name), n, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                         mx / 1e3, 100.0 * tot / total))
    lines.append('| **all kernels** | %d | %.3f | | | | 100 |' % (sum(r[1] for r in rows), total / 1e6))
    text = '\n'.join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text + '\n')


if __name__ == '__main__':
    main()
