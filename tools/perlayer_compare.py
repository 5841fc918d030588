import re, sys, collections
def parse(path):
    rows = collections.OrderedDict()
    for ln in open(path):
        if not ln.startswith('[vr-prof]'): continue
        m = re.match(r'\[vr-prof\] (\S.*?)\s{2,}(\S.*?)\s+([\d.]+) us\s+([\d.]+) GFLOP\s+([\d.]+) MB', ln)
        if not m: continue
        kern, tag, us, gf, mb = m.group(1).strip(), m.group(2).strip(), float(m.group(3)), float(m.group(4)), float(m.group(5))
        if gf <= 0: continue
        key = re.sub(r' \(planes\)$', '', tag)
        rows.setdefault(key, []).append((kern, us, gf))
    return rows
a, b = parse(sys.argv[1]), parse(sys.argv[2])
tot = [0, 0]
for key in a:
    if key not in b: continue
    ka, ua, gf = a[key][-1]; kb, ub, _ = b[key][-1]
    if 'x3' not in ka and 'x3' not in kb: continue
    tot[0] += ua; tot[1] += ub
    print('%-58s %-26s %7.1f us %5.0f TF | %-26s %7.1f us %5.0f TF  %+5.1f%%' % (key, ka.replace('vr::','').replace('_kernel',''), ua, gf/ua*1e-3*1e3/1e3*1e3 if False else gf/ua*1e3/1e3, kb.replace('vr::','').replace('_kernel',''), ub, gf/ub*1e3/1e3, (ub/ua-1)*100))
print('total', tot)
