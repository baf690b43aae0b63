"""D forward + backward alone at the bench shape (kernel times / PMC of the weight-gradient kernels): python tools/d_check.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 251
torch.manual_seed(0)
D = model.Discriminator().cuda()
D.precision = 'bf16'
D.overlap_wgrad = os.environ.get('D_OVERLAP', '0') == '1'
D.train()
din = torch.randn(B, 64, T, 4, device='cuda').abs().requires_grad_(True)
def step():
    D.zero_grad(set_to_none=False) if hasattr(D, 'zero_grad') else None
    s = D.forward_packed(din)
    s.pow(2).sum().backward()
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print('D forward + backward B=%d T=%d: %.2f ms' % (B, T, (time.perf_counter() - t0) / 3 * 1e3))

from nele_gan_amd._lib import lib as _l
if hasattr(_l, 'nele_wgrad_tile_prof_read'):           # a -DWT_PROF build: phase clocks of the weight-gradient tile kernels, per tile visit (all four layers pooled)
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    _l.nele_wgrad_tile_prof_read(buf, 1)
    step(); torch.cuda.synchronize()
    _l.nele_wgrad_tile_prof_read(buf, 1)
    n = max(buf[5], 1)
    print('weight-gradient tile kernels, shader clocks per tile visit: top barrier %.0f  loads + LDS writes %.0f  barrier %.0f  MFMA loop %.0f  (%d visits)' % (buf[4] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[5]))
