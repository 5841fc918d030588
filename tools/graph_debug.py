import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__
vr = __graft_entry__.load_package()
import bench
net, sd = bench.seeded_state(vr)
net.to(torch.device('cuda:0')); net.eval()
wave = torch.from_numpy(bench.synth_wave(float(sys.argv[1]) if len(sys.argv) > 1 else 5.0, 0)).to('cuda:0')
sp = vr.inference.Separator(net, torch.device('cuda:0'), batchsize=0, cropsize=256)
for i in range(4):
    y, v = sp.separate_wave(wave)
    torch.cuda.synchronize()
    print('call', i, float(y.abs().sum()), flush=True)
