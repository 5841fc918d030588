"""bench.py's multi-rank control flow, end to end, on CPU (the RCCL path has never had a peer, so at least the
harness around it must have run before the driver's first 8-GPU launch).

VR_BENCH_STUB=1 swaps the GPU workloads for sleeps (rank r is 1 + r/4 times slower) and the RCCL gradient exchange for a gloo
all-reduce of a CPU bucket; everything else is the code the driver runs: the launcher contract (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`), bench.py's own
self-launch, init_process_group, barrier-bracketed timing, the all-gather of per-rank wall times and MAX over ranks, the separate
all-reduce timing, ONE JSON line from rank 0 only, destroy_process_group."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


DETAIL = None          # set per test: the stub run's detail file goes to the test's tmp dir (a real gpurun_out/bench_detail.json stays untouched)


def _run(cmd):
    env = dict(os.environ, VR_BENCH_STUB='1', OMP_NUM_THREADS='1', VR_BENCH_DETAIL=DETAIL)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0 only), got %d:\n%s' % (len(lines), r.stdout[-2000:])
    # round 4's line was 33 KB and the driver, which keeps the last 8,001 characters of stdout, could not parse it (BENCH_r04.parsed: null):
    # the line must survive ANY tail buffer of that size, also with stderr merged in front of it
    assert len(lines[0]) < 4096, len(lines[0])
    tail = (r.stderr + r.stdout)[-8000:]
    last = json.loads(tail.splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'config', 'roofline'):
        assert key in last, key
    return json.loads(lines[0])


def _check(out, world, steps, warmup):
    assert out['n_gpus'] == world and out['steps'] == steps and out['warmup'] == warmup
    assert out['scaling'] == 'weak' and out['higher_is_better'] is True and out['vs_baseline'] is None
    per_rank = out['ms_per_step_per_rank']
    assert len(per_rank) == world
    assert abs(out['ms_per_step'] - max(per_rank)) < 1e-9                     # MAX over ranks
    # the barriers on both sides make every rank's wall equal the slowest rank's work: >= 1 ms x (1 + (world-1)/4) per step
    assert out['ms_per_step'] >= 1.0 * (1.0 + 0.25 * (world - 1)) * 0.95
    # whole-job aggregate: frames of ALL ranks / the max-over-ranks time
    assert abs(out['value'] - world * 1292 * steps / (out['ms_per_step'] * 1e-3 * steps)) < 1e-6 * out['value']
    tr = out['train']
    assert tr['global_batch'] == 16 * world and len(tr['ms_per_step_per_rank']) == world
    assert abs(tr['value'] - world * 16 * 256 / (tr['ms_per_step'] * 1e-3)) < 1e-6 * tr['value']
    if world > 1:
        assert tr['allreduce_ms'] is not None and tr['allreduce_ms'] > 0
        assert 'cpu_baseline' not in out                                       # rank 0 at N = 1 only
    else:
        assert tr['allreduce_ms'] is None and out['cpu_baseline']['kind'] == 'port'
    roof = out['roofline']
    assert roof['classes'][0][0].startswith('conv_x3') and roof['bound'] == 'mfma' and roof['peak'] == 2500.0
    assert len(roof['classes']) <= 6 and all(len(row) == 4 for row in roof['classes'])      # [class, ms per step, bound, frac]
    assert 'kernels' not in roof and out['detail'] == DETAIL
    detail = json.load(open(DETAIL))
    assert detail['roofline']['kernels'] and detail['roofline']['classes'][0]['class'].startswith('conv_x3')
    assert abs(roof['frac'] - 6 * 1e11 / 2500e12 / 0.5e-3) < 1e-9             # stub rows: 1e11 FLOPs in 0.5 ms on the bf16 pipe


@pytest.mark.parametrize('world', [1, 2, 8])
def test_bench_control_flow_under_the_drivers_launcher(world, tmp_path):
    global DETAIL
    DETAIL = str(tmp_path / 'bench_detail.json')
    steps, warmup = 3, 1
    if world == 1:
        cmd = [sys.executable, BENCH, '--gpus', '1', '--steps', str(steps), '--warmup', str(warmup)]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', str(_port()), BENCH, '--gpus', str(world), '--steps', str(steps), '--warmup', str(warmup)]
    _check(_run(cmd), world, steps, warmup)


def test_bench_self_launch_two_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher environment starts its own ranks (torch.distributed.run, 127.0.0.1)."""
    global DETAIL
    DETAIL = str(tmp_path / 'bench_detail.json')
    _check(_run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1']), 2, 2, 1)


def test_roofline_classes_price_each_kernel_against_its_own_roof():
    sys.path.insert(0, ROOT)
    import bench
    rows = [('vr::conv_x3_kernel<64, 8>', 4, 2.0, 4e11, 4e8, 4),                           # 6 x 4e11 / 2 ms = 1200 TF of 2500
            ('vr::conv_dma_kernel<3, 2, 1, 1, 32, 8, 32, 4, false>', 2, 1.0, 1e11, 1e8, 2),   # 100 TF of 157.3
            ('vr::conv_dma_kernel<1, 1, 1, 1, 32, 8, 16, 32, false>', 2, 0.2, 2e9, 8e8, 2),   # byte roof 0.1 ms > flop roof 0.0127 ms
            ('vr::bn_bwd_apply4_kernel', 3, 0.5, 0.0, 2e9, 3),                               # 4 TB/s of 8
            ('vr::bilstm_reg_kernel<64>', 2, 0.3, 0.0, 0.0, 0)]
    r = bench.roofline_from_rows(rows, 'does_not_exist.json', (3.2, 5.02e11, 8, 1.3e9))
    by = {c['class'].split(':')[0].split(' (')[0]: c for c in r['classes']}
    assert r['kernel'].startswith('conv_x3') and abs(r['frac'] - 0.48) < 1e-9 and r['peak'] == 2500.0 and r['unit'] == 'TFLOP/s'
    c = by['conv 3x3 stride-2 forward, fp32 MFMA']
    assert c['bound'] == 'mfma' and abs(c['achieved'] - 100.0) < 1e-9 and abs(c['frac'] - 100.0 / 157.3) < 1e-9
    c = by['conv 1x1']
    assert c['bound'] == 'hbm' and abs(c['achieved'] - 4000.0) < 1e-6 and abs(c['frac'] - 0.5) < 1e-9
    c = by['element-wise / thin / STFT kernels with algorithmic bytes']
    assert c['bound'] == 'hbm' and abs(c['frac'] - 0.5) < 1e-9
    c = by['other']
    assert c['bound'] is None and c['frac'] is None
    assert abs(r['frac_fp32_equivalent'] - 5.02e11 / 3.2e-3 / 1e12 / 157.3) < 1e-9
    assert r['traffic'] is None and len(r['kernels']) == 5


def test_every_conv_and_wgrad_kernel_of_the_library_has_a_matrix_pipe_class():
    """A conv / weight-gradient kernel that bench.classify() does not know would be priced as an HBM streaming kernel and could become
    the 'dominant class' with a meaningless fraction (round 4: wgrad_wino_r_kernel and conv_x3p_kernel did exactly that once).  The
    kernel names come from the built library's host stubs."""
    import shutil
    sys.path.insert(0, ROOT)
    import bench
    import __graft_entry__
    if not os.path.exists(__graft_entry__.LIB):
        __graft_entry__.build()
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-C', __graft_entry__.LIB], capture_output=True, text=True).stdout
    names = sorted({ln.split('__device_stub__', 1)[1].split('(')[0] for ln in out.splitlines() if '__device_stub__' in ln})
    assert len(names) > 100, len(names)
    # (wgrad_reduce_kernel / wgrad_reduce_batched_kernel sum partial slabs: streaming kernels, not multiplies)
    mac = [n for n in names if n.startswith(('conv_', 'wgrad_')) and 'weights' not in n and not n.startswith('wgrad_reduce')]
    assert any(n.startswith('conv_x3h_kernel') for n in mac) and any(n.startswith('wgrad_wino_r_kernel') for n in mac)
    missing = [n for n in mac if bench.classify('vr::' + n)[1] not in ('bf16', 'f16x3', 'f16w', 'fp32')]
    assert not missing, missing
    # ... and nothing else is claimed by a matrix-pipe class
    wrong = [n for n in names if n not in mac and bench.classify('vr::' + n)[0] is not None]
    assert not wrong, wrong


def test_compact_line_of_a_real_33kb_result_fits_the_tail():
    """The full round-4 result (profiles/r04_bench_all_builder_run.json, 33 KB: three roofline objects with ~90 kernel rows each) through
    compact_line(): the contract's keys survive, the line is < 4 KB, and the last 8000 characters of stdout parse."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_all_builder_run.json')))
    assert len(json.dumps(full)) > 30000
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_LIMIT < 4096
    line = json.loads((('x' * 20000) + '\n' + text + '\n')[-8000:].splitlines()[-1])
    assert line['value'] == full['value'] and line['ms_per_step'] == full['ms_per_step'] and line['steps'] == full['steps']
    assert line['metric'] == full['metric'] and line['config']['workload'] == full['config']['workload']
    r = line['roofline']
    for key in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch'):
        assert key in r, key
    assert abs(r['frac'] - full['roofline']['frac']) < 1e-4 and 1 <= len(r['classes']) <= 6
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] == full['cpu_baseline']['cores']
    assert abs(line['train']['value'] - full['train']['value']) < 1e-9 and 'frac' in line['train'] and 'frac' in line['tta']
    assert line['fp32_mfma']['infer']['ms_per_step'] > 0


COMM_PROBE = r'''
import ctypes, json, sys
sys.path.insert(0, sys.argv[1])
import __graft_entry__
vr = __graft_entry__.load_package()
L, nat = vr.native.lib(), vr.native
out = {}
buf = ctypes.create_string_buffer(128)
for name, call in (('unique_id', lambda: L.vr_comm_unique_id(buf)), ('unique_id_null', lambda: L.vr_comm_unique_id(None)),
                   ('init_null_handle', lambda: L.vr_comm_init(None, 0, 1, buf)), ('allreduce_null_handle', lambda: L.vr_allreduce_grads(None, 0)),
                   ('broadcast_null_handle', lambda: L.vr_broadcast_params(None, 0, 1)), ('destroy_null_handle', lambda: L.vr_comm_destroy(None))):
    rc = call()
    try:
        nat.check(rc)
        out[name] = [rc, 'ok', '']
    except Exception as e:
        out[name] = [rc, type(e).__name__, str(e)]
print('PROBE' + json.dumps(out))
'''


def test_comm_error_paths_without_rccl():
    """comm.hip: RCCL is dlopen'ed at first use; when it cannot be loaded every entry point reports -8 (VRError) instead of crashing,
    and null arguments are refused before RCCL is touched."""
    env = dict(os.environ, VR_RCCL_LIB='/nonexistent/librccl.so.1')
    r = subprocess.run([sys.executable, '-c', COMM_PROBE, ROOT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('PROBE')][-1][5:])
    assert res['unique_id'][0] == -8 and res['unique_id'][1] == 'VRError' and 'cannot load RCCL' in res['unique_id'][2]
    assert res['unique_id_null'][0] == -2 and res['unique_id_null'][1] == 'ValueError'
    for k in ('init_null_handle', 'allreduce_null_handle', 'broadcast_null_handle', 'destroy_null_handle'):
        assert res[k][0] < 0 and res[k][1] in ('ValueError', 'VRError'), (k, res[k])
