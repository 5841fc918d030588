#!/usr/bin/env python
"""bench.py -- spectrogram-frames/sec of the MI355X-native vocal-remover hot path (BASELINE.json's metric:
"spectrogram-frames/sec (infer + train-step), CascadedNet n_fft=2048").

A "step" of the headline `value` is one pass of the inference hot path over one synthetic song resident in HBM
(configs[1]):  STFT -> sliding-window CascadedNet.predict_mask over all 256-frame crops -> stitch -> mask apply ->
iSTFT x2  (inference.py:147-176 of the reference) on CascadedNet(n_fft=2048, hop=1024, 32, 128), fp32, seeded random
weights (no baseline.pth is shipped), 30 s stereo 44.1 kHz synthetic audio -> 1292 frames, 11
crops.  With N GPUs every rank separates its own song (songs shard with no collective): weak scaling.

The same JSON line carries the other two single-GPU configurations as sub-objects, each timed over its own K steps
between barriers after the headline region:
    "tta"    configs[2]: the same song through Separator.separate_tta (23 crops)
    "train"  configs[3]: the train.py step (fwd + L1 + bwd [+ RCCL all-reduce of the flat gradient bucket over the N
             ranks] + Adam), batch 16 x [2,1025,256] per GPU
plus `roofline` and `cpu_baseline` (the CPU oracle -- a port of the reference's path, kind "port" -- timed on this box's host
cores, N=1 only).

`roofline` is measured live: the fastest of THREE EXTRA steps after the timed region, run with every kernel serialised on one
stream and EVERY kernel launch bracketed by HIP events on the stream it is launched on (vr_profile_begin / vr_profile_report).
`roofline.classes` lists each kernel class against ITS OWN ceiling -- conv_x3h (3x3 stride-1, three fp16 products per fp32 product:
14 16-deep matrix instructions per 9 taps x 8 channels = 3.11 executed products per product) against the fp16 matrix pipe
(2500 TFLOP/s dense, achieved = 3.11 x the direct-convolution FLOPs / time; conv_x3 of mfma_mode 2: 6 x), the fp32-MFMA convolutions
and weight gradients against 157.3 TFLOP/s, the 1x1 / thin / element-wise / STFT kernels against 8 TB/s of HBM with their
algorithmic bytes -- a class is priced against whichever of its two roofs (FLOPs / peak, bytes / 8 TB/s) is the longer time.
`roofline.frac` is the DOMINANT class's own fraction; `frac_fp32_equivalent` keeps the round-1..3 aggregate (direct-conv FLOPs of
the whole conv family / fp32-MFMA peak).  `roofline.kernels` carries the per-kernel rows the classes are summed from; the
rocprofv3 summaries of the same command are in profiles/.  HBM traffic per launch comes from the committed rocprofv3 PMC passes.

Arithmetic: fp32 storage, accumulation and results throughout; the 3x3 stride-1 convolutions (84 % of the multiply-adds) form
every product from three fp16 products of two-way split, power-of-two scaled operands on the fp16 matrix pipe (mfma_mode 3, the
library default since round 4: 22 significand bits per operand; measured error against fp64 at or below an fp32 direct
convolution's, tests/test_gpu_parity.py / test_gpu_b16.py).  `split_bf16` carries the same workloads with six bf16 products of
three-way split operands (mfma_mode 2, products exact to fp32; the round-3 default), `fp32_mfma` with v_mfma_f32_32x32x2_f32
everywhere (mfma_mode 0).

`python bench.py --gpus N` without a torch.distributed environment launches its own N ranks
(python -m torch.distributed.run); under the driver's launcher it just reads RANK / WORLD_SIZE.
`--mode infer|tta|train` times one configuration only (profiling runs).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 matrix peak (never the 2:1-sparsity figure)
HBM_PEAK_TBS = 8.0
PROFILE_ROUND = 'r06'              # profiles/<round>_{infer,tta,train}_pmc.json: the PMC passes `traffic` is read from

# kernel name -> (class label, matrix pipe or None).  Everything not listed is priced against HBM when the library noted algorithmic
# bytes for it, and reported as 'other' (latency / launch bound: LSTM recurrence, finalize kernels, descriptor refreshes) when not.
KERNEL_CLASSES = (
    ('conv_x3_kernel', 'conv_x3: 3x3 stride-1 forward + data gradient, fp32 products from six bf16 products', 'bf16'),
    ('conv_x3h_kernel', 'conv_x3h: 3x3 stride-1 forward + data gradient, fp32-grade products from three fp16 products (mfma_mode 3)', 'f16x3'),
    ('conv_x3d_', 'conv_x3d: 16-column layers (ASPP dilated 3x3 + 1x1 branches in one launch, enc5.conv2), forward + data gradient, three fp16 products (mfma_mode 3)', 'f16x3'),
    ('wgrad_wino_', 'wgrad_wino: 3x3 stride-1 weight gradient, Winograd F(3x3,2x2), fp32 MFMA', 'fp32'),
    ('conv_wino_kernel', 'conv_wino: 3x3 stride-1, Winograd F(2x2,3x3), fp32 MFMA (mfma_mode 0)', 'fp32'),
    ('conv_dma_kernel<1,', 'conv 1x1 (ASPP, tails, LSTM projection / dense), fp32 MFMA', 'fp32'),
    ('wgrad_gemm_kernel', 'conv 1x1 weight gradient, fp32 MFMA GEMM', 'fp32'),
    ('wgrad_mfma_kernel<1,', 'conv 1x1 weight gradient, fp32 MFMA GEMM', 'fp32'),
    ('conv_dma_kernel<3, 2', 'conv 3x3 stride-2 forward, fp32 MFMA', 'fp32'),
    ('conv_dma_s2d_kernel', 'conv 3x3 stride-2 data gradient, fp32 MFMA', 'fp32'),
    ('wgrad_ws_kernel', 'conv 3x3 stride-2 / leftover weight gradient, fp32 MFMA', 'fp32'),
    ('conv_dma_kernel<3,', 'conv 3x3 dilated (ASPP) + 16-wide stride-1, fp32 MFMA', 'fp32'),
    ('wgrad_mfma_kernel<3,', 'conv 3x3 dilated / 16-wide weight gradient, fp32 MFMA', 'fp32'),
    ('conv_thin_kernel', 'conv 3x3 with <= 16 output channels, fp32 MFMA 16x16x4', 'fp32'),
    ('conv_ws_kernel', 'conv fused-loader leftovers, fp32 MFMA', 'fp32'),
    ('conv_mfma_kernel', 'conv fused-loader leftovers, fp32 MFMA', 'fp32'),
)
SR, N_FFT, HOP, CROP = 44100, 2048, 1024, 256


def synth_wave(seconds, seed):
    rng = np.random.default_rng(seed)
    L = int(round(seconds * SR))
    t = np.arange(L, dtype=np.float64) / SR
    wave = 0.1 * rng.standard_normal((2, L))
    for f in (220.0, 440.0, 3520.0):
        wave += 0.2 * np.sin(2 * np.pi * f * t[None, :] + rng.uniform(0, 2 * np.pi, size=(2, 1)))
    return wave.astype(np.float32)


def seeded_state(vr, seed=1234):
    """Seeded random weights incl. non-trivial BatchNorm statistics (timing is weight independent)."""
    torch.manual_seed(seed)
    net = vr.nets.CascadedNet(N_FFT, HOP, 32, 128)
    sd = net.state_dict()
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith('running_mean'):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.2 - 0.1
        elif k.endswith('running_var') or (k.endswith('.1.weight') and sd[k].dim() == 1):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.6 + 0.7
        elif k.endswith('.1.bias') and sd[k].dim() == 1:
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.3 - 0.15
    net.load_state_dict(sd)
    return net, sd


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 hardware threads behind a 16-CPU quota; oversubscribing makes the
    CPU oracle 5x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


# ---- cpu_baseline legs: the only place bench.py touches oracle/ (the checker, timed beside the product) -----------
def cpu_baseline_infer(sd, frames=1292):
    """CPU oracle (port of the reference's path) on the same song once."""
    from oracle import separator as osep, stft_np
    cores = usable_cores()
    torch.set_num_threads(cores)
    L = HOP * (frames - 1) + 1
    wave = synth_wave(L / SR + 0.01, 0)[:, :L]
    t0 = time.perf_counter()
    spec = stft_np.wave_to_spectrogram(wave, HOP, N_FFT)
    y, v = osep.separate(spec, sd, tta=False, n_fft=N_FFT, batchsize=4, cropsize=CROP)
    stft_np.spectrogram_to_wave(y.astype(np.complex64), HOP)
    stft_np.spectrogram_to_wave(v.astype(np.complex64), HOP)
    dt = time.perf_counter() - t0
    T = spec.shape[2]
    return {'value': T / dt, 'unit': 'spectrogram-frames/sec', 'cores': cores, 'kind': 'port',
            'sample': 'the same workload once: %d frames (%.1f s of synthetic audio, %d crops), oracle (port of the '
                      'reference path, torch CPU) STFT->separate(batch 4)->iSTFT x2, %.1f s wall' % (T, L / SR, -(-T // 128) + 1, dt)}


def _host_memory_gb():
    """Memory this process may use: MemAvailable capped by the cgroup limit."""
    avail = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                avail = int(line.split()[1]) / 1e6
    except OSError:
        pass
    try:
        lim = open('/sys/fs/cgroup/memory.max').read().strip()
        if lim != 'max':
            cur = int(open('/sys/fs/cgroup/memory.current').read().strip())
            left = (int(lim) - cur) / 1e9
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail if avail is not None else 8.0


def cpu_baseline_train(sd, want_batch=16):
    """CPU oracle train step (fwd + L1 + bwd + Adam) at the GPU's batch (16: ~40 GB of host memory, ~30-60 s on 16 cores), halved
    only while the host's memory does not hold it (autograd keeps ~2.6 GB of activations per sample in fp32)."""
    from oracle import train_step as ots, weights as ow
    cores = usable_cores()
    torch.set_num_threads(cores)
    mem = _host_memory_gb()
    B = want_batch
    while B > 2 and 3.0 * B + 4.0 > 0.7 * mem:                # ~2.6 GB of saved activations per sample
        B //= 2
    sd = ow.clone_state_dict(sd)
    X, y = ots.synth_batch(B, T=CROP, n_fft=N_FFT, seed=0)
    opt = ots.Adam(lr=1e-3)
    t0 = time.perf_counter()
    loss, grads = ots.loss_and_grads(sd, X, y)
    opt.step(sd, grads)
    dt = time.perf_counter() - t0
    return {'value': B * CROP / dt, 'unit': 'spectrogram-frames/sec', 'cores': cores, 'kind': 'port',
            'sample': 'one oracle train step (port: autograd over the restated net + restated Adam) at batch %d x [2,1025,256] '
                      '(%.0f GB of host memory available; the GPU runs batch %d) -- %.1f s wall' % (B, mem, want_batch, dt)}


SPLIT_DTYPE = ('f32 (3x3 stride-1 convs: fp32 operands as two power-of-two scaled fp16 planes, three fp16 products per fp32 product on '
               'v_mfma_f32_32x32x16_f16, fp32 accumulate -- error vs fp64 <= an fp32 direct convolution\'s; rest: fp32 MFMA / VALU)')


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


# ---- roofline: per-kernel rows (vr_profile_report) -> classes, each against its own ceiling --------------------------------
def classify(name):
    for prefix, label, pipe in KERNEL_CLASSES:
        if prefix in name:
            return label, pipe
    return None, None


def roofline_from_rows(rows, pmc_name, conv_totals):
    """rows: [(kernel name, calls, ms, algorithmic flops, algorithmic bytes, calls with figures)] of ONE profiled step."""
    classes = {}
    for name, calls, ms, flops, nbytes, noted in rows:
        label, pipe = classify(name)
        if label is None:
            label, pipe = (('element-wise / thin / STFT kernels with algorithmic bytes (HBM streaming)', None) if noted and nbytes > 0
                           else ('other: LSTM recurrence, BatchNorm finalize, weight-table refreshes, copies (latency / launch bound, no '
                                 'algorithmic figure)', 'none'))
        c = classes.setdefault(label, {'class': label, 'pipe': pipe, 'ms_per_step': 0.0, 'launches': 0, 'flops': 0.0, 'bytes': 0.0, 'kernels': []})
        c['ms_per_step'] += ms; c['launches'] += calls; c['flops'] += flops; c['bytes'] += nbytes
        c['kernels'].append(name.replace('vr::', ''))
    out = []
    for c in classes.values():
        ms, pipe = c['ms_per_step'], c.pop('pipe')
        # executed products per fp32 product: six on the bf16 pipe (conv_x3); conv_x3h issues 14 16-deep instructions per 9 taps x 8 channels
        mult = {'bf16': 6.0, 'f16x3': 14 * 16 / 72.0, 'f16w': 3.0}.get(pipe, 1.0)
        pk = {'bf16': BF16_MFMA_PEAK_TFLOPS, 'f16x3': BF16_MFMA_PEAK_TFLOPS, 'f16w': BF16_MFMA_PEAK_TFLOPS, 'fp32': FP32_MFMA_PEAK_TFLOPS}.get(pipe)
        t_flop = (mult * c['flops'] / (pk * 1e12) * 1e3) if pk else 0.0          # ms at the matrix-pipe peak
        t_byte = c['bytes'] / (HBM_PEAK_TBS * 1e12) * 1e3                         # ms at the HBM peak
        if pipe == 'none' or ms <= 0 or (t_flop == 0 and t_byte == 0):
            c.update({'bound': None, 'peak': None, 'achieved': None, 'unit': None, 'frac': None})
        elif t_flop >= t_byte:
            c.update({'bound': 'mfma', 'pipe': 'bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), 6 executed products per fp32 product' if pipe == 'bf16'
                      else 'fp16 matrix pipe (v_mfma_f32_32x32x16_f16, same dense peak as bf16), 3.11 executed products per fp32 product' if pipe == 'f16x3'
                      else 'fp16 matrix pipe (v_mfma_f32_32x32x16_f16), 3 executed products per fp32 product' if pipe == 'f16w'
                      else 'fp32 matrix pipe (v_mfma_f32_32x32x2_f32 / 16x16x4)', 'peak': pk, 'unit': 'TFLOP/s',
                      'achieved': mult * c['flops'] / (ms * 1e-3) / 1e12, 'frac': t_flop / ms})
        else:
            c.update({'bound': 'hbm', 'peak': HBM_PEAK_TBS * 1e3, 'unit': 'GB/s', 'achieved': c['bytes'] / (ms * 1e-3) / 1e9, 'frac': t_byte / ms})
        if 'Winograd' in c['class'] and c.get('frac') is not None and c['bound'] == 'mfma':
            # `achieved` counts the direct convolution's FLOPs; the Winograd kernels execute 2.25x fewer multiply-adds on the matrix pipe
            c['executed_frac'] = c['frac'] / 2.25
            c['executed_note'] = 'Winograd F(2x2,3x3) / F(3x3,2x2): 16 multiplies per tile where the direct form has 36 -- the matrix pipe itself is busy frac / 2.25'
        c['algorithmic_gflop'] = c.pop('flops') / 1e9
        c['algorithmic_mb'] = c.pop('bytes') / 1e6
        out.append(c)
    out.sort(key=lambda c: -c['ms_per_step'])
    dom = next(c for c in out if c['bound'] is not None)
    cms, cfl, cn, cby = conv_totals
    traffic, src = None, None
    # rocprofv3 PMC passes of this same command (counters cannot be read in-process): this round's file, else the newest earlier one
    for cand in [pmc_name] + [pmc_name.replace(PROFILE_ROUND, 'r%02d' % n, 1) for n in range(int(PROFILE_ROUND[1:]) - 1, 0, -1)]:
        path = os.path.join(ROOT, 'profiles', cand)
        if os.path.exists(path):
            traffic = json.load(open(path)).get('bytes_per_launch')
            src = 'profiles/' + cand
            break
    total_ms = sum(c['ms_per_step'] for c in out)
    return {'bound': dom['bound'], 'kernel': dom['class'], 'achieved': dom['achieved'], 'peak': dom['peak'], 'unit': dom['unit'],
            'frac': dom['frac'], 'traffic': traffic, 'traffic_source': src,
            'frac_note': 'the dominant class (largest share of the serialised kernel time: %.2f of %.2f ms) against ITS OWN ceiling; every '
                         'class is in `classes`: frac = max(executed FLOPs / pipe peak, algorithmic bytes / 8 TB/s) / measured time'
                         % (dom['ms_per_step'], total_ms),
            'frac_fp32_equivalent': (cfl / (cms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS) if cms > 0 else None,
            'frac_fp32_equivalent_note': 'rounds 1-3 figure: direct-convolution FLOPs of ALL conv launches / their summed time / %.1f TFLOP/s '
                                         '(fp32 MFMA peak) -- not a ceiling for conv_x3, which runs on the bf16 pipe' % FP32_MFMA_PEAK_TFLOPS,
            'classes': out,
            'kernels': [[n.replace('vr::', ''), calls, round(ms, 4), round(fl / 1e9, 3), round(by / 1e6, 3)] for n, calls, ms, fl, by, _ in
                        sorted(rows, key=lambda r: -r[2])],
            'kernels_columns': ['kernel', 'launches', 'ms', 'algorithmic GFLOP', 'algorithmic MB'],
            'source': 'the fastest of THREE EXTRA steps after the timed region, every launch serialised on one stream and bracketed by HIP events on '
                      'the stream it is launched on (the timed steps overlap lanes and streams, so kernel_ms_per_step can exceed ms_per_step)',
            'traffic_unit': 'HBM bytes per conv launch (mean over the conv launches of a step; rocprofv3 FETCH_SIZE x2 '
                            'gfx950 correction + WRITE_SIZE, %s)' % src,
            'algorithmic_bytes_per_launch': cby / max(cn, 1),
            'algorithmic_bytes_note': 'input-sized tensor + output-sized tensor + weights, once each, for EVERY conv launch (forward, data '
                                      'gradient, weight gradient); rounds 1-3 counted the forward launches only and divided by all launches',
            'launches_per_step': sum(c['launches'] for c in out), 'conv_launches_per_step': cn, 'kernel_ms_per_step': total_ms,
            'conv_kernel_ms_per_step': cms, 'algorithmic_gflop_per_step': cfl / 1e9}


# ---- the stdout line: small enough for any tail buffer (round 4's 33 KB line overflowed the driver's 8 KB tail and went unparsed) ----
LINE_LIMIT = 4000                  # bytes; tests/test_bench_multirank.py asserts the line stays below it
DETAIL_PATH = os.environ.get('VR_BENCH_DETAIL') or os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')     # (tests point it at a tmp dir)


def _r(x, digits=5):
    """Floats to `digits` significant digits (the full-precision figures are in the detail file)."""
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _short_class(label):
    return label.split(':')[0].split(' (')[0].split(',')[0][:44]


def compact_roofline(r, nclasses=6):
    """Scalars of a roofline object + at most `nclasses` one-line class rows [class, ms per step, bound, frac]."""
    if r is None:
        return None
    o = {k: _r(r.get(k)) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'algorithmic_bytes_per_launch', 'frac_fp32_equivalent',
                                   'kernel_ms_per_step', 'conv_kernel_ms_per_step', 'launches_per_step', 'conv_launches_per_step')}
    o['kernel'] = _short_class(r['kernel'])
    o['classes'] = [[_short_class(c['class']), _r(c['ms_per_step'], 4), c['bound'], _r(c['frac'], 3)] for c in r['classes'][:nclasses]]
    return o


def compact_line(out):
    """The ONE stdout line: the contract's keys, `roofline` (scalars + <= 6 class rows), `cpu_baseline`, and the other configurations as
    {value, ms_per_step, frac}.  Everything else (classes in full, per-kernel rows, notes) goes to gpurun_out/bench_detail.json."""
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data') if k in out}
    line['dtype_note'] = str(out.get('dtype_note', ''))[:200]                # what THIS run computed in (VR_MFMA_MODE / --bf16 change it)
    line['ms_per_step_per_rank'] = out.get('ms_per_step_per_rank')          # unrounded: ms_per_step is their MAX
    line['allreduce_ms'] = _r(out.get('allreduce_ms'), 4)
    line['config'] = {k: _r(v) for k, v in out['config'].items()}
    line['roofline'] = compact_roofline(out.get('roofline'))
    for key in ('tta', 'train', 'train_bf16'):
        if key in out:
            sub = out[key]
            c = {'value': sub['value'], 'ms_per_step': sub['ms_per_step']}
            if sub.get('roofline'):
                rr = sub['roofline']
                c.update({'frac': _r(rr['frac'], 3), 'kernel': _short_class(rr['kernel']), 'bound': rr['bound'],
                          'kernel_ms_per_step': _r(rr['kernel_ms_per_step'], 4),
                          'classes': [[_short_class(x['class']), _r(x['ms_per_step'], 4), x['bound'], _r(x['frac'], 3)] for x in rr['classes'][:4]]})
            if sub.get('global_batch') is not None:
                c['global_batch'] = sub['global_batch']
            if 'allreduce_ms' in sub:
                c['allreduce_ms'] = _r(sub['allreduce_ms'], 4)
                c['ms_per_step_per_rank'] = sub.get('ms_per_step_per_rank')
            if sub.get('cpu_baseline'):
                c['cpu_baseline'] = {k: _r(v) for k, v in sub['cpu_baseline'].items() if k != 'sample'}
            line[key] = c
    for key in ('fp32_mfma', 'split_bf16'):
        if key in out:
            line[key] = {w: {'value': _r(out[key][w]['value']), 'ms_per_step': _r(out[key][w]['ms_per_step'])} for w in ('infer', 'train')}
    if 'cpu_baseline' in out:
        cb = dict(out['cpu_baseline'])
        cb['sample'] = cb.get('sample', '')[:160]
        line['cpu_baseline'] = {k: _r(v) for k, v in cb.items()}
    line['detail'] = os.path.relpath(DETAIL_PATH, ROOT) if DETAIL_PATH.startswith(ROOT) else DETAIL_PATH
    text = json.dumps(line, separators=(',', ':'))
    # belt and braces: shed optional parts rather than ever print a line a tail buffer would cut
    for drop in (('split_bf16',), ('fp32_mfma',), ('train_bf16',), ('tta', 'classes'), ('train', 'classes'), ('roofline', 'classes')):
        if len(text) <= LINE_LIMIT:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k) or {}             # (a key that is present but None must not cost the line)
        tgt.pop(drop[-1], None)
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def write_detail(out):
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, 'w') as f:
            json.dump(out, f, indent=1)
    except OSError as e:                      # a read-only checkout must not cost the run its line
        print('bench.py: could not write %s: %s' % (DETAIL_PATH, e), file=sys.stderr)


# ---- runtimes: the real one (HIP library + RCCL) and a stub that keeps ONLY the multi-rank control flow -----------------------
class Runtime(object):
    """World / rank / barrier / max-over-ranks timing shared by every workload.  backend 'nccl' = RCCL on the GPUs; 'gloo' + stub
    workloads (VR_BENCH_STUB=1) run the same control flow on CPU for tests/test_bench_multirank.py."""

    def __init__(self, args):
        self.stub = bool(os.environ.get('VR_BENCH_STUB'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if self.world != args.gpus and self.rank == 0:
            print('bench.py: --gpus %d but the launcher started %d ranks; measuring %d' % (args.gpus, self.world, self.world), file=sys.stderr)
        self.dist = None
        if self.stub:
            self.dev = torch.device('cpu')
            if self.world > 1:
                import torch.distributed as dist
                self.dist = dist
                dist.init_process_group('gloo')
        else:
            if self.world > 1:
                import torch.distributed as dist
                self.dist = dist
                os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                dist.init_process_group('nccl', device_id=torch.device('cuda', self.local_rank))
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device('cuda', self.local_rank)
        self.per_rank = []

    def sync(self):
        if not self.stub:
            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def timed(self, step, steps, warmup):
        """W untimed warm-up steps, then EXACTLY K steps between barriers; MAX over ranks."""
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        dt = time.perf_counter() - t0
        self.per_rank = [dt]
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(self.world)]
            self.dist.all_gather(every, tt)
            self.per_rank = [float(t.item()) for t in every]
            dt = max(self.per_rank)
        return dt

    def max_over_ranks(self, value):
        if self.world > 1:
            tt = torch.tensor([value], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            value = float(tt.item())
        return value

    def finish(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


class NativeWorkloads(object):
    """The product: libvr_mi355.so through vocal_remover_amd."""

    def __init__(self, rt, args):
        self.rt, self.args = rt, args
        if not os.path.exists(__graft_entry__.LIB):
            __graft_entry__.build()
        self.vr = __graft_entry__.load_package()
        self.nat = self.vr.native
        self.net, self.sd = seeded_state(self.vr)
        self.net.to(rt.dev)
        L = int(round(args.seconds * SR))
        self.T = 1 + L // HOP
        pl, pr, roi = self.vr.dataset.make_padding(self.T, CROP, 64)
        self.crops_plain = (self.T + pl + pr - 128) // roi
        self.crops_tta = self.crops_plain + (self.T + pl + pr + roi - 128) // roi
        self.wave_host = synth_wave(args.seconds, rt.rank)
        self.wave = torch.from_numpy(self.wave_host).to(rt.dev)
        self.sp = self.vr.inference.Separator(self.net, rt.dev, batchsize=args.batchsize, cropsize=CROP)

    def infer_step(self, tta, host=False):
        self.net.eval()
        wave = self.wave_host if host else self.wave
        return lambda: self.sp.separate_wave(wave, tta=tta)

    def set_mode(self, mfma_mode=None, bf16=None):
        if bf16 is not None:
            self.net.set_option('mfma_bf16', 1 if bf16 else 0)
        if mfma_mode is not None:
            self.net.set_option('mfma_mode', mfma_mode)

    def trainer(self, wire, exchange_at_world1=False):
        from vocal_remover_amd import train as vtrain
        rt = self.rt
        # exchange_at_world1 (configs[4] slice): the gradient bucket goes through the library's RCCL all-reduce in the wire format
        # even on one GPU, so that the step that is timed is the step the 8-GPU job runs (bf16 rounding of the bucket included)
        rccl = rt.world > 1 or exchange_at_world1
        tr = vtrain.Trainer(self.net, lr=1e-3, world_size=rt.world, rank=rt.rank, backend='rccl' if rccl else 'none',
                            wire=wire if rccl else 'fp32')
        g = torch.Generator().manual_seed(rt.rank)
        B = self.args.train_batch
        X = torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)
        y = (X * torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)).to(rt.dev)
        X = X.to(rt.dev)
        return (lambda: tr.step(X, y)), tr.reduce, self.net.zero_grad

    def end_train(self):
        self.net.eval()

    def profile(self, step):
        """One step with every kernel launch timed (vr_profile_begin/end/report) -> (conv totals, per-kernel rows)."""
        nat, h = self.nat, self.net._handle.h
        nat.check(nat.lib().vr_profile_begin(h))
        step()
        cms, cfl, cn, cby = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        nat.check(nat.lib().vr_profile_end(h, ctypes.byref(cms), ctypes.byref(cfl), ctypes.byref(cn), ctypes.byref(cby)))
        need = nat.lib().vr_profile_report(h, None, 0)
        buf = ctypes.create_string_buffer(int(need))
        nat.lib().vr_profile_report(h, buf, need)
        rows = []
        for ln in buf.value.decode().splitlines():
            name, calls, ms, fl, by, noted = ln.split('\t')
            rows.append((name, int(calls), float(ms), float(fl), float(by), int(noted)))
        return (cms.value, cfl.value, cn.value, cby.value), rows

    def cpu_baseline(self, which):
        return cpu_baseline_infer(self.sd) if which == 'infer' else cpu_baseline_train(self.sd, self.args.train_batch)


class StubWorkloads(object):
    """VR_BENCH_STUB=1: no GPU, no library -- sleeps stand for kernels, a gloo all-reduce of a bucket-sized CPU tensor for the gradient
    exchange.  Exists so that bench.py's multi-rank control flow (self-launch, barriers, per-rank wall all-gather, all-reduce timing,
    rank-0-only JSON line) runs under pytest at world 2 and 8 before the driver's first 8-GPU launch."""

    def __init__(self, rt, args):
        self.rt, self.args = rt, args
        self.T, self.crops_plain, self.crops_tta = 1292, 11, 23
        self.bucket = torch.zeros(1 << 16)

    def infer_step(self, tta, host=False):
        ms = (2.0 if tta else 1.0) * (1.0 + 0.25 * self.rt.rank)          # the slowest rank must set the job's time
        return lambda: time.sleep(ms * 1e-3)

    def set_mode(self, mfma_mode=None, bf16=None):
        pass

    def trainer(self, wire, exchange_at_world1=False):
        rt = self.rt

        def reduce():
            if rt.world > 1:
                rt.dist.all_reduce(self.bucket, op=rt.dist.ReduceOp.SUM)

        def step():
            time.sleep(3e-3 * (1.0 + 0.25 * rt.rank))
            reduce()
        return step, reduce, (lambda: None)

    def end_train(self):
        pass

    def profile(self, step):
        step()
        rows = [('vr::conv_x3_kernel<64, 8>', 10, 0.5, 1e11, 1e8, 10), ('vr::conv_dma_kernel<1, 1, 1, 1, 32, 8, 16, 32, false>', 4, 0.1, 1e9, 5e7, 4),
                ('vr::upsample2x_rows_kernel', 3, 0.05, 0.0, 2e7, 3), ('vr::bilstm_reg_kernel<64>', 2, 0.1, 0.0, 0.0, 0)]
        return (0.6, 1.01e11, 14, 1.5e8), rows

    def cpu_baseline(self, which):
        return {'value': 1.0, 'unit': 'spectrogram-frames/sec', 'cores': 1, 'kind': 'port', 'sample': 'stub'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', choices=['all', 'infer', 'tta', 'train'], default='all')
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--tta', action='store_true', help='same as --mode tta')
    ap.add_argument('--batchsize', type=int, default=0, help='crops per device batch (0 = all crops of a pass)')
    ap.add_argument('--train-batch', type=int, default=16)
    ap.add_argument('--wire', choices=['fp32', 'bf16'], default='fp32', help='gradient bucket format on xGMI (N > 1)')
    ap.add_argument('--bf16', action='store_true', help='--mode train: bf16 MFMA operands + bf16 bucket (an experiment, not configs[4])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if args.tta:
        args.mode = 'tta'
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)               # (the launcher parent only waits for torchrun: no watchdog here -- it would orphan the ranks)
    # A hung communicator must end in a traceback, not in the driver's timeout (1800 s).  The deadline is PER PHASE (build, each workload,
    # each profiled step set, the CPU baselines): `pet()` re-arms it, so a healthy but slow 8-GPU `--mode all` run is never cut short.
    # Default for N > 1: 900 s per phase; VR_BENCH_WATCHDOG=0 turns it off, any other value sets the per-phase seconds.
    watchdog = int(os.environ.get('VR_BENCH_WATCHDOG', '900' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else '0'))

    def pet():
        if watchdog > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(watchdog, exit=True)
    pet()
    # stdout carries ONE JSON line and nothing else: libraries write there too (RCCL prints a version banner at its first communicator,
    # from C stdio), so file descriptor 1 points at stderr for the whole run and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rt = Runtime(args)                 # (builds the library on first use)
    pet()
    wl = (StubWorkloads if rt.stub else NativeWorkloads)(rt, args)
    world, rank, T = rt.world, rt.rank, wl.T
    pmc = lambda which: '%s_%s_pmc.json' % (PROFILE_ROUND, which)        # noqa: E731

    def roofline(step, pmc_name):
        pet()
        # three profiled steps, the fastest one is reported: a single serialised step is exposed to one-off stalls (seen once: a tta
        # step at 2x its usual kernel time while the timed steps of the same run were normal)
        conv_totals, rows = min((wl.profile(step) for _ in range(3)), key=lambda r: sum(x[2] for x in r[1]))
        return roofline_from_rows(rows, pmc_name, conv_totals)

    def run_infer(tta):
        pet()
        step = wl.infer_step(tta)
        dt = rt.timed(step, args.steps, args.warmup)
        crops = wl.crops_tta if tta else wl.crops_plain
        res = {'frames_per_sec': world * T * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3,
               'computed_frames_per_sec': world * crops * CROP * args.steps / dt, 'crops_per_step_per_gpu': crops,
               'frames_per_step_per_gpu': T, 'ms_per_step_per_rank': [t / args.steps * 1e3 for t in rt.per_rank]}
        return res, step

    def run_train(bf16=False, mfma_mode=None, configs4=False):
        pet()
        wl.set_mode(bf16=bf16)
        if mfma_mode is not None:
            wl.set_mode(mfma_mode=mfma_mode)
        wire = 'bf16' if (configs4 or (bf16 and world > 1)) else args.wire
        step, reduce, zero_grad = wl.trainer(wire, exchange_at_world1=configs4)
        B = args.train_batch
        dt = rt.timed(step, args.steps, args.warmup)
        per_rank = [t / args.steps * 1e3 for t in rt.per_rank]
        allreduce_ms = None
        if world > 1:
            # the exchange alone, outside the timed region: the gradient bucket of the last step, summed three more times with a
            # device sync on both sides (vr_allreduce_grads runs on the library's own stream)
            ts = []
            for _ in range(3):
                rt.barrier()
                t0 = time.perf_counter()
                reduce()
                rt.sync()                         # device-wide: also the library's own (non-blocking) stream
                ts.append((time.perf_counter() - t0) * 1e3)
            allreduce_ms = rt.max_over_ranks(min(ts))
            zero_grad()
        res = {'frames_per_sec': world * B * CROP * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3,
               'ms_per_step_per_rank': per_rank, 'allreduce_ms': allreduce_ms,
               'global_batch': world * B, 'frames_per_step_per_gpu': B * CROP,
               'workload': 'configs[%d]: train.py step, batch %d x [2,1025,256] per GPU, fwd + L1 + bwd%s + Adam; '
                           'Dropout2d live (library RNG)' % (4 if configs4 else 3, B,
                                                             ' + RCCL all-reduce (%s bucket)' % wire if (world > 1 or configs4) else ''),
               'dtype': 'bf16 MFMA operands (Winograd forward / data-gradient / weight-gradient convs, 1x1 weight-gradient GEMM), '
                        'fp32 accumulation, storage, master weights and Adam; remaining convs fp32' if bf16 else SPLIT_DTYPE,
               'parallelism': 'dp%d (one RCCL all-reduce of the flat 14.74 M-element gradient bucket per step)' % world}
        return res, step

    primary_mode = 'infer' if args.mode == 'all' else args.mode
    out, extra = None, {}
    if primary_mode in ('infer', 'tta'):
        res, step = run_infer(primary_mode == 'tta')
        workload = ('configs[%d]: 30 s stereo 44.1 kHz synthetic song, STFT -> %d crops of 256 frames in one device batch -> '
                    'predict_mask -> stitch -> mask apply -> iSTFT x2%s' % (2 if primary_mode == 'tta' else 1, res['crops_per_step_per_gpu'],
                                                                             ' (--tta)' if primary_mode == 'tta' else '')) \
            if args.seconds == 30.0 else '%.0f s synthetic song, %d crops' % (args.seconds, res['crops_per_step_per_gpu'])
        metric = 'spectrogram-frames/sec (inference, CascadedNet n_fft=2048)'
        parallelism = 'replicas x%d (songs shard, no collective)' % world
        roof = roofline(step, pmc(primary_mode))
    else:
        res, step = run_train(args.bf16)
        workload, metric, parallelism = res['workload'], 'spectrogram-frames/sec (train-step, CascadedNet n_fft=2048)', res['parallelism']
        roof = roofline(step, pmc('train'))
        wl.end_train()
    if rank == 0:
        out = {
            'metric': metric, 'value': res['frames_per_sec'], 'unit': 'spectrogram-frames/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if (args.bf16 or os.environ.get('VR_MFMA_MODE') == '1') else 'f32', 'dtype_note': res.get('dtype', SPLIT_DTYPE),
            'ms_per_step_per_rank': res.get('ms_per_step_per_rank'), 'allreduce_ms': res.get('allreduce_ms'),
            'data': 'synthetic (seeded noise + sines; seeded random weights, no baseline.pth exists)',
            'config': {'workload': workload, 'n_fft': N_FFT, 'hop': HOP, 'cropsize': CROP,
                       'frames_per_step_per_gpu': res['frames_per_step_per_gpu'], 'parallelism': parallelism,
                       'input_residency': 'inputs resident in HBM when the timed region starts'},
            'roofline': roof,
        }
        if 'computed_frames_per_sec' in res:
            out['config']['crops_per_step_per_gpu'] = res['crops_per_step_per_gpu']
            out['config']['computed_frames_per_sec'] = res['computed_frames_per_sec']

    if args.mode == 'all':
        # PCIe-inclusive rate of the headline workload: host numpy in, host numpy out (2 x 10.6 MB + 2 x 10.6 MB per step)
        if world == 1:
            dt = rt.timed(wl.infer_step(False, host=True), max(3, args.steps // 2), 1)
            extra['pcie_inclusive_frames_per_sec'] = T * max(3, args.steps // 2) / dt
        tta_res, tta_step = run_infer(True)
        tta_roof = roofline(tta_step, pmc('tta'))
        train_res, train_step_fn = run_train()
        train_roof = roofline(train_step_fn, pmc('train'))
        c4_res, _ = run_train(configs4=True)
        # the same workloads with v_mfma_f32_32x32x2_f32 everywhere (mfma_mode 0: Winograd F(2x2,3x3) on the fp32 matrix pipe for the
        # 3x3 stride-1 layers, materialised decoder upsample) -- the round-1/2 arithmetic, beside the default above
        wl.set_mode(mfma_mode=0)
        f32_inf, _ = run_infer(False)
        try:
            f32_trn, _ = run_train(False, mfma_mode=0)
            # ... and with the round-3 default: fp32 products from six bf16 products (mfma_mode 2, conv_x3.hip)
            wl.set_mode(mfma_mode=2)
            b16_inf, _ = run_infer(False)
            b16_trn, _ = run_train(False, mfma_mode=2)
        finally:
            wl.set_mode(mfma_mode=-1)
        wl.end_train()
        if rank == 0:
            out['config']['pcie_inclusive_frames_per_sec'] = extra.get('pcie_inclusive_frames_per_sec')
            out['tta'] = {'metric': 'spectrogram-frames/sec (inference --tta, configs[2])', 'value': tta_res['frames_per_sec'],
                          'ms_per_step': tta_res['ms_per_step'], 'steps': args.steps, 'warmup': args.warmup,
                          'crops_per_step_per_gpu': tta_res['crops_per_step_per_gpu'],
                          'computed_frames_per_sec': tta_res['computed_frames_per_sec'], 'roofline': tta_roof}
            out['train'] = {'metric': 'spectrogram-frames/sec (train-step, configs[3])', 'value': train_res['frames_per_sec'],
                            'ms_per_step': train_res['ms_per_step'], 'steps': args.steps, 'warmup': args.warmup,
                            'global_batch': train_res['global_batch'], 'workload': train_res['workload'],
                            'parallelism': train_res['parallelism'], 'dtype': SPLIT_DTYPE, 'roofline': train_roof,
                            'ms_per_step_per_rank': train_res['ms_per_step_per_rank'], 'allreduce_ms': train_res['allreduce_ms']}
            out['train_bf16'] = {
                'metric': 'spectrogram-frames/sec (train-step, configs[4] slice: data-parallel step with the bf16 gradient bucket, %d GPU%s)' % (world, 's' if world > 1 else ''),
                'value': c4_res['frames_per_sec'], 'ms_per_step': c4_res['ms_per_step'], 'steps': args.steps, 'warmup': args.warmup,
                'global_batch': c4_res['global_batch'], 'workload': c4_res['workload'], 'parallelism': c4_res['parallelism'],
                'ms_per_step_per_rank': c4_res['ms_per_step_per_rank'], 'allreduce_ms': c4_res['allreduce_ms'],
                'dtype': '16-bit where it is exact or harmless: the 3x3 stride-1 convolutions multiply on the 16-bit matrix pipe (two-way split fp16 '
                         'operands, three products per fp32 product: fp32-grade, mfma_mode 3) and the gradient bucket crosses xGMI in bf16 (rounded '
                         'once, summed by RCCL in bf16, widened back); activations, master weights, accumulation and Adam stay fp32',
                'what_it_is_not': 'a bf16-STORAGE pipeline: bf16 operands without the split lose the gradient direction on this net (cosine 0.37 vs fp32 at '
                                  'batch 16, tests/test_gpu_b16.py) and storing the three bf16 planes (6 B per element) measured no faster than fp32 storage '
                                  '(tools/experiments/conv_x3p.hip, DESIGN.md section 3)',
                'parity': 'tests/test_gpu_b16.py::test_b16_configs4_slice_vs_fp32: loss 1e-4 relative, global gradient cosine >= 0.99, per-tensor median >= 0.98 against '
                          'mfma_mode 0 with the fp32 bucket at batch 16'}
            out['fp32_mfma'] = {
                'what': "vr_set_option('mfma_mode', 0): every convolution on v_mfma_f32_32x32x2_f32 / 16x16x4 (fp32 operands; Winograd "
                        'F(2x2,3x3) for the 3x3 stride-1 layers, decoder upsample materialised) -- the default (mode 3) instead forms the '
                        'products of those layers from three fp16 products on v_mfma_f32_32x32x16_f16; all modes pass the same parity '
                        'tests at the same tolerances (tests/test_gpu_parity.py, test_gpu_configs.py, test_gpu_b16.py)',
                'dtype': 'f32',
                'infer': {'value': f32_inf['frames_per_sec'], 'ms_per_step': f32_inf['ms_per_step']},
                'train': {'value': f32_trn['frames_per_sec'], 'ms_per_step': f32_trn['ms_per_step']}}
            out['split_bf16'] = {
                'what': "vr_set_option('mfma_mode', 2): the 3x3 stride-1 layers with fp32 products from SIX bf16 products of three-way split "
                        'operands on v_mfma_f32_32x32x16_bf16 (conv_x3.hip; every product exact to fp32, 27 matrix instructions per 8-channel '
                        'chunk where the default mode 3 issues 14) -- the round-3 default',
                'dtype': 'f32',
                'infer': {'value': b16_inf['frames_per_sec'], 'ms_per_step': b16_inf['ms_per_step']},
                'train': {'value': b16_trn['frames_per_sec'], 'ms_per_step': b16_trn['ms_per_step']}}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            pet()
            out['cpu_baseline'] = wl.cpu_baseline('infer' if primary_mode in ('infer', 'tta') else 'train')
            if args.mode == 'all':
                out['train']['cpu_baseline'] = wl.cpu_baseline('train')
        write_detail(out)
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(json_fd, (compact_line(out) + '\n').encode())
    rt.finish()


if __name__ == '__main__':
    main()
