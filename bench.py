#!/usr/bin/env python
"""bench.py -- spectrogram-frames/sec of the MI355X-native vocal-remover hot path.

A "step" is one pass of the inference hot path over one synthetic song resident in HBM:
    STFT -> sliding-window CascadedNet.predict_mask over all 256-frame crops -> stitch ->
    mask apply -> iSTFT x2                                   (inference.py:147-176 of the reference)
on CascadedNet(n_fft=2048, hop=1024, 32, 128), fp32, seeded random weights (no baseline.pth is
shipped), 30 s stereo 44.1 kHz synthetic audio -> 1292 frames, 11 crops.
With N GPUs every rank separates its own song (songs shard with no collective): weak scaling.

`--mode train` times the train.py step (fwd + L1 + bwd + Adam, batch 16 x [2,1025,256]) instead.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel: the
fp32-MFMA conv family, algorithmic FLOPs / HIP-event time per launch, summed over the launches of
one step) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded excerpt).
"""
import argparse
import json
import os
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


def cpu_baseline_infer(sd, frames=1292):
    """CPU oracle (port of the reference's path) on a bounded excerpt of the same song."""
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
            'sample': 'the same workload once: %d frames (%.1f s of synthetic audio, %d crops), oracle '
                      'STFT->separate(batch 4)->iSTFT x2, %.1f s wall' % (T, L / SR, -(-T // 128) + 1, dt)}


CONV_FAMILY_INFER = ('conv family on v_mfma_f32_32x32x2_f32: conv_wino_kernel<*> (Winograd F(2x2,3x3), the 3x3 stride-1 '
                     'layers) + conv_dma_kernel<*> (direct implicit GEMM: stride-2, dilated, 1x1, thin layers)')
CONV_FAMILY_TRAIN = ('conv family on v_mfma_f32_32x32x2_f32: conv_wino_kernel<*> / conv_dma_kernel<*> (forward + data '
                     'gradients over materialised plain tensors; stride-2 data gradient = 4 tap-masked parity convs) + '
                     'wgrad_ws_kernel<*> (weight gradient, LDS-DMA loader)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', choices=['infer', 'train'], default='infer')
    ap.add_argument('--seconds', type=float, default=30.0)
    ap.add_argument('--tta', action='store_true')
    ap.add_argument('--batchsize', type=int, default=0, help='crops per device batch (0 = all crops of a pass)')
    ap.add_argument('--train-batch', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if os.environ.get('VR_BENCH_WATCHDOG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['VR_BENCH_WATCHDOG']), exit=True)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    if not os.path.exists(__graft_entry__.LIB):
        __graft_entry__.build()
    vr = __graft_entry__.load_package()
    net, sd = seeded_state(vr)
    net.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    L = int(round(args.seconds * SR))
    T = 1 + L // HOP
    if args.mode == 'infer':
        net.eval()
        sp = vr.inference.Separator(net, dev, batchsize=args.batchsize, cropsize=CROP)
        wave = torch.from_numpy(synth_wave(args.seconds, rank)).to(dev)

        def step():
            return sp.separate_wave(wave, tta=args.tta)
        frames_per_step = T
        pl, pr, roi = vr.dataset.make_padding(T, CROP, 64)
        crops = (T + pl + pr - 128) // roi
        if args.tta:
            crops += (T + pl + pr + roi - 128) // roi
        workload = ('configs[1]: 30 s stereo 44.1 kHz synthetic song, STFT -> %d crops of 256 frames in one '
                    'device batch -> predict_mask -> stitch -> mask apply -> iSTFT x2' % crops) if args.seconds == 30.0 \
            else '%.0f s synthetic song, %d crops' % (args.seconds, crops)
        if args.tta:
            workload += ' (--tta: configs[2])'
    else:
        from vocal_remover_amd import train as vtrain       # noqa: E402
        trainer = vtrain.Trainer(net, lr=1e-3, world_size=world, rank=rank)
        g = torch.Generator().manual_seed(rank)
        B = args.train_batch
        X = torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)
        y = (X * torch.rand((B, 2, N_FFT // 2 + 1, CROP), generator=g)).to(dev)
        X = X.to(dev)

        def step():
            return trainer.step(X, y)
        frames_per_step = B * CROP
        crops = B
        workload = 'configs[3]: train.py step, batch %d x [2,1025,256] per GPU, fwd + L1 + bwd + Adam' % B

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # roofline of the dominant kernel family, measured live with HIP events on the library's stream
    nat = vr.native
    import ctypes
    nat.check(nat.lib().vr_profile_begin(net._handle.h))
    step()
    cms, cfl, cn, cby = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    nat.check(nat.lib().vr_profile_end(net._handle.h, ctypes.byref(cms), ctypes.byref(cfl), ctypes.byref(cn),
                                       ctypes.byref(cby)))
    achieved = cfl.value / (cms.value * 1e-3) / 1e12 if cms.value > 0 else 0.0

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command (collected
    # separately -- counters cannot be read from inside the process), summary committed in profiles/
    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'r01_infer_pmc.json')
    if args.mode == 'infer' and not args.tta and args.seconds == 30.0 and os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get('bytes_per_launch')

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        out = {
            'metric': 'spectrogram-frames/sec (%s, CascadedNet n_fft=2048)' % ('inference' if args.mode == 'infer' else 'train-step'),
            'value': world * frames_per_step * args.steps / dt,
            'unit': 'spectrogram-frames/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic (seeded noise + sines; seeded random weights, no baseline.pth exists)',
            'config': {'workload': workload, 'n_fft': N_FFT, 'hop': HOP, 'cropsize': CROP,
                       'frames_per_step_per_gpu': frames_per_step, 'crops_per_step_per_gpu': crops,
                       'computed_frames_per_sec': world * crops * CROP * args.steps / dt,
                       'parallelism': 'replicas x%d (songs shard, no collective)' % world if args.mode == 'infer'
                       else 'dp%d (RCCL all-reduce of one flat fp32 gradient bucket)' % world},
            'roofline': {'bound': 'mfma',
                         'kernel': CONV_FAMILY_INFER if args.mode == 'infer' else CONV_FAMILY_TRAIN,
                         'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                         'traffic_unit': 'HBM bytes per launch (mean over the conv launches of a step; rocprofv3 '
                                         'FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/r01_infer_pmc.json)',
                         'algorithmic_bytes_per_launch': cby.value / max(cn.value, 1),
                         'achieved_note': 'algorithmic FLOPs = 2 x multiply-adds of the direct convolutions / summed '
                                          'launch durations (HIP events); the Winograd launches execute 2.25x fewer',
                         'launches_per_step': cn.value, 'kernel_ms_per_step': cms.value,
                         'algorithmic_gflop_per_step': cfl.value / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == 'infer':
            out['cpu_baseline'] = cpu_baseline_infer(sd)
        elif world == 1 and not args.no_cpu_baseline:
            from vocal_remover_amd import train as vtrain
            out['cpu_baseline'] = vtrain.cpu_baseline_train(sd)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
