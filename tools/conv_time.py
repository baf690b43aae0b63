import sys, numpy as np, torch
sys.path.insert(0, '.')
from nele_gan_amd import model as M, ops
torch.manual_seed(0)
B, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 251
D = M.Discriminator(nout=2).cuda()
D.precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
din = torch.rand(B, 64, T, 4, device='cuda', requires_grad=False)
for _ in range(3): D.forward_packed(din)
tags = ['D.conv%d.fwd' % i for i in range(1, 6)]
ops.PROFILE = {t: [] for t in tags}
for _ in range(10): D.forward_packed(din)
torch.cuda.synchronize()
for t in tags:
    ev = ops.PROFILE[t]
    ms = sorted(e0.elapsed_time(e1) for e0, e1, _ in ev)[len(ev)//2]
    fl = ev[0][2]
    print(t, 'median %.3f ms  %.1f TFLOP/s' % (ms, fl / ms / 1e9))
