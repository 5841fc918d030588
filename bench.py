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
plus `roofline` (dominant kernel family = the MFMA convs: algorithmic FLOPs / HIP-event time per launch, summed over the
launches of the fastest of THREE EXTRA steps run with every kernel serialised on one stream -- not the timed steps, which overlap
lanes and streams; HBM traffic per launch from the committed rocprofv3 PMC passes) and `cpu_baseline` (the CPU oracle -- a port of
the reference's path, kind "port" -- timed on this box's host cores, N=1 only).

Arithmetic: fp32 throughout; the 3x3 stride-1 convolutions form every fp32 product from six bf16 products of three-way
split operands on the bf16 matrix pipe (mfma_mode 2, the library default: error against fp64 = an fp32 direct
convolution's).  `fp32_mfma` carries the same workloads with v_mfma_f32_32x32x2_f32 everywhere (mfma_mode 0).

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
    """CPU oracle train step (fwd + L1 + bwd + Adam) at the largest batch <= the GPU's that fits the host's memory
    (autograd keeps ~2.6 GB of activations per sample in fp32) and a ~30 s budget."""
    from oracle import train_step as ots, weights as ow
    cores = usable_cores()
    torch.set_num_threads(cores)
    mem = _host_memory_gb()
    B = want_batch
    while B > 2 and (3.0 * B + 4.0 > 0.7 * mem or B > 8):     # 8 samples ~ 25 s on 16 cores: the bounded-sample budget
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
                      '-- the largest power of two <= 16 that fits %.0f GB of host memory and the ~30 s sample budget (the GPU runs '
                      'batch %d) -- %.1f s wall' % (B, mem, want_batch, dt)}


CONV_FAMILY_INFER = ('conv family: conv_x3_kernel<*> (3x3 stride-1 layers, direct, fp32 products from six bf16 products on '
                     'v_mfma_f32_32x32x16_bf16; decoder bilinear x2 fused) + conv_dma_kernel<*> (stride-2, dilated, 1x1, 16-wide: fp32 MFMA) '
                     '+ conv_thin_kernel<*> (<= 16 output channels, fp32 MFMA); mfma_mode 0: conv_wino_kernel<*> (Winograd F(2x2,3x3), '
                     'fp32 MFMA) instead of conv_x3')
CONV_FAMILY_TRAIN = ('conv family: conv_x3_kernel<*> / conv_dma_kernel<*> / conv_thin_kernel<*> (forward + data gradients over '
                     'materialised plain tensors; stride-2 data gradient = conv_dma_s2d_kernel) + wgrad_*_kernel<*> (weight gradient on '
                     'the fp32 MFMA: Winograd F(3x3,2x2), LDS-DMA GEMMs)')
SPLIT_DTYPE = 'f32 (3x3 stride-1 convs: bf16x3 split operands, six bf16 products per fp32 product, fp32 accumulate; rest: fp32 MFMA / VALU)'


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
    ap.add_argument('--bf16', action='store_true', help='--mode train: configs[4] arithmetic (bf16 MFMA operands, bf16 bucket)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if args.tta:
        args.mode = 'tta'
    if os.environ.get('VR_BENCH_WATCHDOG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['VR_BENCH_WATCHDOG']), exit=True)

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print('bench.py: --gpus %d but the launcher started %d ranks; measuring %d' % (args.gpus, world, world), file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    if not os.path.exists(__graft_entry__.LIB):
        __graft_entry__.build()
    vr = __graft_entry__.load_package()
    nat = vr.native
    net, sd = seeded_state(vr)
    net.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """W untimed warm-up steps, then EXACTLY K steps between barriers; MAX over ranks."""
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        timed.per_rank = [dt]
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)
            timed.per_rank = [float(t.item()) for t in every]
            dt = max(timed.per_rank)
        return dt
    timed.per_rank = []

    def conv_profile(step):
        """HIP events (on the library's stream) around every MFMA-conv launch of one step, kernels serialised."""
        nat.check(nat.lib().vr_profile_begin(net._handle.h))
        step()
        cms, cfl, cn, cby = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        nat.check(nat.lib().vr_profile_end(net._handle.h, ctypes.byref(cms), ctypes.byref(cfl), ctypes.byref(cn), ctypes.byref(cby)))
        return cms.value, cfl.value, cn.value, cby.value

    def roofline(step, kernel, pmc_name):
        # three profiled steps, the fastest one is reported: a single serialised step is exposed to one-off stalls (seen once: a tta
        # step at 2x its usual kernel time while the timed steps of the same run were normal)
        cms, cfl, cn, cby = min((conv_profile(step) for _ in range(3)), key=lambda r: r[0])
        achieved = cfl / (cms * 1e-3) / 1e12 if cms > 0 else 0.0
        traffic, src = None, None
        path = os.path.join(ROOT, 'profiles', pmc_name)
        if os.path.exists(path):        # rocprofv3 PMC passes of this same command (counters cannot be read in-process)
            traffic = json.load(open(path)).get('bytes_per_launch')
            src = 'profiles/' + pmc_name
        return {'bound': 'mfma', 'kernel': kernel, 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                'source': 'the fastest of THREE EXTRA steps after the timed region, every launch serialised on one stream and bracketed by HIP events on '
                          'the library\'s stream (the timed steps overlap 2 lanes x 2 streams, so kernel_ms_per_step can exceed ms_per_step)',
                'peak_note': 'peak = fp32 MFMA (v_mfma_f32_32x32x2_f32 = the fp32 vector rate).  In mfma_mode 2 the 3x3 stride-1 convs run '
                             'on the bf16 pipe instead: 2500 TFLOP/s dense / 6 products = %.0f fp32-equivalent TFLOP/s at 2.4 GHz -- the '
                             'chip clocks down to ~1.7 GHz under that load (profiles/README.md); their MFMA-pipe busy fraction is in '
                             'profiles/r03_infer_sq_pmc.md' % (2500.0 / 6.0),
                'traffic_unit': 'HBM bytes per launch (mean over the conv launches of a step; rocprofv3 FETCH_SIZE x2 '
                                'gfx950 correction + WRITE_SIZE, %s)' % src,
                'algorithmic_bytes_per_launch': cby / max(cn, 1),
                'achieved_note': 'algorithmic FLOPs = 2 x multiply-adds of the direct convolutions / summed launch '
                                 'durations (HIP events); the Winograd launches execute 2.25x fewer',
                'launches_per_step': cn, 'kernel_ms_per_step': cms, 'algorithmic_gflop_per_step': cfl / 1e9}

    L = int(round(args.seconds * SR))
    T = 1 + L // HOP
    pl, pr, roi = vr.dataset.make_padding(T, CROP, 64)
    crops_plain = (T + pl + pr - 128) // roi
    crops_tta = crops_plain + (T + pl + pr + roi - 128) // roi
    wave_host = synth_wave(args.seconds, rank)
    wave = torch.from_numpy(wave_host).to(dev)
    sp = vr.inference.Separator(net, dev, batchsize=args.batchsize, cropsize=CROP)

    def run_infer(tta):
        net.eval()
        step = lambda: sp.separate_wave(wave, tta=tta)                     # noqa: E731
        dt = timed(step, args.steps, args.warmup)
        crops = crops_tta if tta else crops_plain
        res = {'frames_per_sec': world * T * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3,
               'computed_frames_per_sec': world * crops * CROP * args.steps / dt, 'crops_per_step_per_gpu': crops,
               'frames_per_step_per_gpu': T, 'ms_per_step_per_rank': [t / args.steps * 1e3 for t in timed.per_rank]}
        return res, step

    def run_train(bf16=False):
        from vocal_remover_amd import train as vtrain
        net.set_option('mfma_bf16', 1 if bf16 else 0)
        wire = 'bf16' if (bf16 and world > 1) else args.wire
        trainer = vtrain.Trainer(net, lr=1e-3, world_size=world, rank=rank, backend='rccl' if world > 1 else 'none',
                                 wire=wire if world > 1 else 'fp32')
        g = torch.Generator().manual_seed(rank)
        B = args.train_batch
        X = torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)
        y = (X * torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)).to(dev)
        X = X.to(dev)
        step = lambda: trainer.step(X, y)                                   # noqa: E731
        dt = timed(step, args.steps, args.warmup)
        per_rank = [t / args.steps * 1e3 for t in timed.per_rank]
        allreduce_ms = None
        if world > 1:
            # the exchange alone, outside the timed region: the gradient bucket of the last step, summed three more times with a
            # device sync on both sides (vr_allreduce_grads runs on the library's own stream)
            ts = []
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                trainer.reduce()
                torch.cuda.synchronize()          # device-wide: also the library's own (non-blocking) stream
                ts.append((time.perf_counter() - t0) * 1e3)
            tt = torch.tensor([min(ts)], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            allreduce_ms = float(tt.item())
            net.zero_grad()
        res = {'frames_per_sec': world * B * CROP * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3,
               'ms_per_step_per_rank': per_rank, 'allreduce_ms': allreduce_ms,
               'global_batch': world * B, 'frames_per_step_per_gpu': B * CROP,
               'workload': 'configs[%d]: train.py step, batch %d x [2,1025,256] per GPU, fwd + L1 + bwd%s + Adam; '
                           'Dropout2d live (library RNG)' % (4 if bf16 else 3, B, ' + RCCL all-reduce (%s bucket)' % wire if world > 1 else ''),
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
        roof = roofline(step, CONV_FAMILY_INFER, 'r03_infer_pmc.json' if primary_mode == 'infer' else 'r03_tta_pmc.json')
    else:
        res, step = run_train(args.bf16)
        workload, metric, parallelism = res['workload'], 'spectrogram-frames/sec (train-step, CascadedNet n_fft=2048)', res['parallelism']
        roof = roofline(step, CONV_FAMILY_TRAIN, 'r03_train_pmc.json')
        net.eval()
    if rank == 0:
        out = {
            'metric': metric, 'value': res['frames_per_sec'], 'unit': 'spectrogram-frames/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': res.get('dtype', SPLIT_DTYPE),
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
            net.eval()
            dt = timed(lambda: sp.separate_wave(wave_host, tta=False), max(3, args.steps // 2), 1)
            extra['pcie_inclusive_frames_per_sec'] = T * max(3, args.steps // 2) / dt
        tta_res, tta_step = run_infer(True)
        tta_roof = roofline(tta_step, CONV_FAMILY_INFER, 'r03_tta_pmc.json')
        train_res, train_step_fn = run_train()
        train_roof = roofline(train_step_fn, CONV_FAMILY_TRAIN, 'r03_train_pmc.json')
        bf_res, _ = run_train(True)
        net.set_option('mfma_bf16', 0)
        # the same workloads with v_mfma_f32_32x32x2_f32 everywhere (mfma_mode 0: Winograd F(2x2,3x3) on the fp32 matrix pipe for the
        # 3x3 stride-1 layers, materialised decoder upsample) -- the round-1/2 arithmetic, beside the default above
        net.set_option('mfma_mode', 0)
        f32_inf, _ = run_infer(False)
        net.set_option('mfma_mode', 0)          # (run_train returns to the default mode through 'mfma_bf16')
        f32_trn = None
        try:
            from vocal_remover_amd import train as vtrain
            trainer = vtrain.Trainer(net, lr=1e-3, world_size=world, rank=rank, backend='rccl' if world > 1 else 'none',
                                     wire=args.wire if world > 1 else 'fp32')
            g = torch.Generator().manual_seed(rank)
            B = args.train_batch
            Xs = torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)
            ys = (Xs * torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)).to(dev)
            Xs = Xs.to(dev)
            dts = timed(lambda: trainer.step(Xs, ys), args.steps, args.warmup)
            f32_trn = {'frames_per_sec': world * B * CROP * args.steps / dts, 'ms_per_step': dts / args.steps * 1e3}
        finally:
            net.set_option('mfma_mode', -1)
        net.eval()
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
            out['train_bf16'] = {'metric': 'spectrogram-frames/sec (train-step with bf16 MFMA OPERANDS on fp32 storage, %d GPU%s -- NOT configs[4] '
                                           'itself: no bf16 activation storage%s)' % (world, 's' if world > 1 else '', '' if world > 1 else ', no data parallelism'),
                                 'value': bf_res['frames_per_sec'], 'ms_per_step': bf_res['ms_per_step'], 'steps': args.steps,
                                 'warmup': args.warmup, 'global_batch': bf_res['global_batch'], 'workload': bf_res['workload'],
                                 'parallelism': bf_res['parallelism'], 'dtype': bf_res['dtype']}
            out['fp32_mfma'] = {
                'what': "vr_set_option('mfma_mode', 0): every convolution on v_mfma_f32_32x32x2_f32 / 16x16x4 (fp32 operands; Winograd "
                        'F(2x2,3x3) for the 3x3 stride-1 layers, decoder upsample materialised) -- the default (mode 2) instead forms the '
                        'fp32 products of those layers from six bf16 products on v_mfma_f32_32x32x16_bf16; both modes pass the same parity '
                        'tests at the same tolerances (tests/test_gpu_parity.py, test_gpu_configs.py, test_gpu_b16.py)',
                'dtype': 'f32',
                'infer': {'value': f32_inf['frames_per_sec'], 'ms_per_step': f32_inf['ms_per_step']},
                'train': {'value': f32_trn['frames_per_sec'], 'ms_per_step': f32_trn['ms_per_step']} if f32_trn else None}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if primary_mode in ('infer', 'tta'):
                out['cpu_baseline'] = cpu_baseline_infer(sd)
            else:
                out['cpu_baseline'] = cpu_baseline_train(sd, args.train_batch)
            if args.mode == 'all':
                out['train']['cpu_baseline'] = cpu_baseline_train(sd, args.train_batch)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
