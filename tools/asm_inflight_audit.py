"""Linear-scan audit of a hipcc -save-temps .s file for kernels with inline-asm loads and hand-placed waits: between a
global_load / buffer_load and the s_waitcnt vmcnt(N) that covers it, no instruction may read or overwrite the load's destination
registers (hipcc treats an asm load's output as written at ASMEND and may copy or reuse it; cdna_hip_programming.md "What hipcc does
not do").  Usage: asm_inflight_audit.py file.s kernel_name_substring

Round 5: LDS-DMA loads and stores take vmcnt slots in the model (they were left out: every wait looked two or three entries short and
conv_x3h showed 18 false reports).  State with that: wgrad_wino_r_kernel (all four instantiations) and the plain 64x8 / 32x8 tilings
of conv_x3h: 0 reports.  What is still reported for conv_x3h<32,16> and the fused-upsample forms are v_mad_u64_u32 / v_cndmask pairs of
the rare source-switch path (pixel_offset recomputed in next_source) that use the DESTINATION of the load they precede as a
temporary -- a dead register at that point; the linear scan cannot see that, because it concatenates mutually exclusive paths and
carries the in-flight list across the loop's back edge.  A heuristic for a first look, not a proof: the parity and bit-stability tests
are the check."""
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
        burst, since_dma = 0, 1 << 30
        for ln in body:
            t = ln.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            op = t.split()[0]
            since_dma += 1
            # LDS-DMA loads (`... lds`) and, on gfx9, stores count towards vmcnt like register loads: they enter the in-flight list with an
            # EMPTY register set, so that the vmcnt(N) arithmetic matches the hardware's (leaving them out made every wait look
            # NWMIN entries short -- 18 false 'TOUCHES IN-FLIGHT' reports on conv_x3h).  Limitation of the linear scan: a DMA that the source
            # writes as `if (full) dma else if (partial) dma` is two instructions of which one executes; the DMAs of a burst
            # (at most 24 instructions apart) are therefore counted in pairs.
            if (op.startswith('global_load') or op.startswith('buffer_load')) and ' lds' in t:
                burst = burst + 1 if since_dma <= 24 else 1      # the 1st, 3rd, 5th ... DMA of a burst count: one per `if / else if` pair
                if burst & 1:
                    inflight.append(set())
                since_dma = 0
                continue
            if op.startswith('global_store') or op.startswith('buffer_store'):
                inflight.append(set())
                continue
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
