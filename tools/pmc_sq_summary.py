"""MFMA-pipe utilisation and wave-state counters per kernel from one rocprofv3 --pmc pass with the SQ counters
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT.
Usage: pmc_sq_summary.py run.db calib.db [out.json]
calib.db = the same counters on tools/mfma_peak.hip (v_mfma_f32_32x32x2_f32 back to back from registers, 146 of the 157.3
TFLOP/s peak = 0.93 busy by construction): its SQ_VALU_MFMA_BUSY_CYCLES per ns of kernel time fixes the scale, so the
utilisation below needs neither the counter's unit nor the clock."""
import json
import re
import sqlite3
import sys

CAL_UTIL = 146.0 / 157.3


def load(path):
    db = sqlite3.connect(path)
    per = {}
    for did, name, cn, cv, dur in db.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events"):
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)
        e = per.setdefault(did, {'name': name, 'dur': dur})
        e[cn] = e.get(cn, 0.0) + cv
    return per


cal = [e for e in load(sys.argv[2]).values() if 'mfma_loop' in e['name'] and 'bf16' not in e['name']]
cal_rate = max(e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['dur'] for e in cal)       # busy counts per ns at CAL_UTIL
full = cal_rate / CAL_UTIL
groups = {}
for e in load(sys.argv[1]).values():
    g = groups.setdefault(e['name'][:70], {'n': 0})
    g['n'] += 1
    for k, v in e.items():
        if k != 'name':
            g[k] = g.get(k, 0.0) + v
rows = sorted(groups.items(), key=lambda kv: -kv[1]['dur'])
tot = sum(g['dur'] for _, g in rows)
print('| kernel | launches | time ms | MFMA pipe busy | waves: waiting (s_waitcnt / barrier) | waves: issue-stalled | VALU active | LDS active | LDS bank-conflict / wave-cycles |')
print('|---|---|---|---|---|---|---|---|---|')
out = {}
for name, g in rows:
    if g['dur'] < 0.002 * tot:
        continue
    wc = max(g.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    util = g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / g['dur'] / full
    out[name] = util
    print('| %s | %d | %.3f | %.3f | %.2f | %.2f | %.2f | %.2f | %.3f |' % (
        name, g['n'], g['dur'] / 1e6, util, g.get('SQ_WAIT_ANY', 0) / wc, g.get('SQ_WAIT_INST_ANY', 0) / wc,
        g.get('SQ_ACTIVE_INST_VALU', 0) / wc, g.get('SQ_ACTIVE_INST_LDS', 0) / wc, g.get('SQ_LDS_BANK_CONFLICT', 0) / wc))
allutil = sum(g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for _, g in rows) / tot / full
print('| **all kernels** | %d | %.3f | %.3f | | | | | |' % (sum(g['n'] for _, g in rows), tot / 1e6, allutil))
print('\ncalibration: tools/mfma_peak.hip mfma_loop = %.4f SQ_VALU_MFMA_BUSY_CYCLES per ns at %.2f of the fp32-MFMA peak' % (cal_rate, CAL_UTIL))
if len(sys.argv) > 3:
    json.dump({'mfma_busy_by_kernel': out, 'mfma_busy_all_kernels': allutil, 'calibration_counts_per_ns': cal_rate,
               'calibration_util': CAL_UTIL}, open(sys.argv[3], 'w'), indent=1)
