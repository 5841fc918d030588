"""Linear-scan audit of a hipcc -save-temps .s file for kernels with inline-asm loads and hand-placed waits: between a
global_load / buffer_load and the s_waitcnt vmcnt(N) that covers it, no instruction may read or overwrite the load's destination
registers (hipcc treats an asm load's output as written at ASMEND and may copy or reuse it; cdna_hip_programming.md "What hipcc does
not do").  Usage: asm_inflight_audit.py file.s kernel_name_substring"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2]
    cur, funcs = None, {}
    for ln in lines:
        m = re.match(r'^(\w+):', ln)
        if m and want in m.group(1) and not m.group(1).startswith('.L'):
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None:
            if ln.startswith('.Lfunc_end'):
                cur = None
                continue
            funcs[cur].append(ln)
    for name, body in funcs.items():
        inflight, bad, nload = [], 0, 0
        for ln in body:
            t = ln.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            op = t.split()[0]
            if (op.startswith('global_load') or op.startswith('buffer_load')) and ' lds' not in t:
                args = t[len(op):].split(',')
                live = set().union(*inflight) if inflight else set()
                if regs(','.join(args[1:])) & live:
                    bad += 1
                    print(name, 'ADDRESS USES AN IN-FLIGHT REGISTER:', t)
                inflight.append(regs(args[0]))
                nload += 1
                continue
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if 0 < n < len(inflight) else ([] if n == 0 else inflight)
                continue
            live = set().union(*inflight) if inflight else set()
            if live and regs(t[len(op):]) & live:
                bad += 1
                if bad <= 10:
                    print(name, 'TOUCHES IN-FLIGHT', sorted(regs(t[len(op):]) & live)[:6], ':', t)
        print('%s: %d register loads, %d violations' % (name, nload, bad))


if __name__ == '__main__':
    main()
