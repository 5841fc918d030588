"""Audit of the hand-waited inline-asm pixel loads of conv_x3pp / conv_x3h in a hipcc .s file (round 6).

hipcc treats the destination of an inline-asm `buffer_load_dword` as written when the asm statement ends; the data arrives later.  Any
instruction that reads, copies or overwrites such a register before the covering wait works on stale data -- silently, and only when
the load is slow (large layers, HBM misses).  x3h_wait8 names the registers it releases in a comment (`; landed v1 v2 ...`), so the
state of every destination register is known without modelling vmcnt arithmetic:

    asm load writes v  ->  v is IN FLIGHT  ->  `; landed ... v ...` or `s_waitcnt vmcnt(0)`  ->  v is free

A linear scan in text order: mutually exclusive paths are concatenated (conservative for the rare source-switch path; the two group
loops of conv_x3pp start from the same state, the one the prologue leaves).  Usage: asm_inflight_audit2.py file.s kernel_substring
Exit status 1 when something is reported."""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
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
    total = 0
    for name, body in funcs.items():
        inflight = {}          # reg -> line number of the load
        bad = nload = nwait = 0
        in_asm = False
        for i, ln in enumerate(body):
            t = ln.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if in_asm and t.startswith('; landed'):
                for r in regs(t):
                    inflight.pop(r, None)
                nwait += 1
                continue
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            op = t.split()[0]
            code = t.split(';')[0]
            if in_asm and op.startswith('buffer_load_dword') and ' lds' not in code:
                args = code[len(op):].split(',')
                live = set(inflight)
                if regs(','.join(args[1:])) & live:
                    bad += 1
                    print(name, 'line', i, 'ADDRESS USES AN IN-FLIGHT REGISTER:', t)
                for r in regs(args[0]):
                    inflight[r] = i
                nload += 1
                continue
            if in_asm and t.startswith(';') is False and False:
                pass
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', code)
            if m:
                if ';' in t and 'landed' in t:
                    for r in regs(t.split('landed')[1]):
                        inflight.pop(r, None)
                    nwait += 1
                elif int(m.group(1)) == 0:
                    inflight.clear()
                continue
            if op.startswith('s_') or op.startswith('ds_') and False:
                continue
            touched = regs(code[len(op):]) & set(inflight)
            if touched:
                bad += 1
                if bad <= 12:
                    print(name, 'line', i, 'TOUCHES IN-FLIGHT', sorted(touched), 'loaded at', sorted({inflight[r] for r in touched})[:3], ':', t)
        print('%s: %d asm loads, %d waits, %d reports' % (name, nload, nwait, bad))
        total += bad
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
