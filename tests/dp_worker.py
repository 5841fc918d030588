"""Worker of tests/test_gpu_dp.py: one data-parallel rank.  Launched by torch.distributed.run with a gloo process
group (host channel: the ncclUniqueId / the staged bucket).  argv: out_dir [backend [wire]]
  backend 'staged' (default): every rank opens its handle on cuda:0 (ranks share the one GPU of the test box), the gradient
                    bucket is staged through host memory for the all-reduce;
  backend 'rccl':   rank r opens cuda:r and the library's own RCCL communicator carries the bucket over xGMI (fp32 or bf16
                    wire) -- needs as many GPUs as ranks.
Rank r trains on shard r of the batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
from oracle import train_step, weights  # noqa: E402  (seeded inputs only)


def main():
    out_dir = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else 'staged'
    wire = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    __graft_entry__.load_package()
    from vocal_remover_amd import nets, train as vtrain
    n_fft, nout, nl, per = 512, 8, 32, 2
    # rank 0 holds the real weights, the others start from different ones: the constructor's broadcast must fix that
    sd = weights.make_state_dict(11 if rank == 0 else 100 + rank, n_fft=n_fft, nout=nout, nout_lstm=nl)
    model = nets.CascadedNet(n_fft, n_fft // 2, nout, nl)
    model.load_state_dict(sd)
    dev = 'cuda:%d' % (rank if backend == 'rccl' else 0)
    model.to(torch.device(dev))
    tr = vtrain.Trainer(model, lr=1e-3, world_size=world, rank=rank, dropout=False, backend=backend, wire=wire)
    X, y = train_step.synth_batch(per * world, T=64, n_fft=n_fft, seed=7)
    Xs, ys = X[rank * per:(rank + 1) * per].to(dev), y[rank * per:(rank + 1) * per].to(dev)
    loss = model.train_step(Xs, ys, 1)
    tr.reduce()
    grads = {k: v.numpy() * tr.opt.grad_scale for k, v in model.grads().items()}
    tr.opt.step()
    model.zero_grad()
    state = {k: v.numpy() for k, v in model.state_dict().items()}
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), loss=np.float64(loss),
             **{'g::' + k: v for k, v in grads.items()}, **{'p::' + k: v for k, v in state.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
