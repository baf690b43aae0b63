"""D's conv weight gradients alone at the bench shape: per-layer time (HIP events) of nele_conv_wgrad on the bf16 buffers of a real
forward / backward pass, and the result against the one-kernel-row tile kernel (NELE_WGRAD_DMA=0 in a second process would be the
A/B; here: both libraries' results are compared through the D.backward gradients).  python tools/wgrad_check.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import model, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 251
torch.manual_seed(0)
D = model.Discriminator().cuda()
D.precision = 'bf16'
D.overlap_wgrad = False
D.train()
din = torch.randn(B, 64, T, 4, device='cuda').abs().requires_grad_(True)
tags = ['D.conv%d.wgrad' % l for l in range(1, 6)]
def step():
    D.flat_parameters().grad.zero_()
    s = D.forward_packed(din)
    s.pow(2).sum().backward()
for _ in range(2):
    step()
ops.PROFILE = {t: [] for t in tags}
for _ in range(3):
    step()
torch.cuda.synchronize()
tot = 0.0
for t in tags:
    ms = [e0.elapsed_time(e1) for e0, e1, _ in ops.PROFILE[t]]
    tot += sum(ms) / len(ms)
    print('%s: %.3f ms (incl. its partial reduction)' % (t, sum(ms) / len(ms)))
print('sum %.3f ms; NELE_WGRAD_DMA=%s' % (tot, os.environ.get('NELE_WGRAD_DMA', '1')))
ops.PROFILE = None
g = D.flat_parameters().grad.clone()
print('grad checksum %.9e  absmax %.6e  finite %s' % (float(g.double().sum()), float(g.abs().max()), bool(torch.isfinite(g).all())))
torch.save(g.cpu(), '/tmp/wgrad_%s.pt' % os.environ.get('NELE_WGRAD_DMA', '1'))
if os.path.exists('/tmp/wgrad_0.pt') and os.path.exists('/tmp/wgrad_1.pt'):
    a, b = torch.load('/tmp/wgrad_0.pt'), torch.load('/tmp/wgrad_1.pt')
    print('DMA vs tile kernel: max abs diff %.3e (of max %.3e), rel L2 %.3e' % (float((a - b).abs().max()), float(a.abs().max()), float((a - b).norm() / a.norm())))
