"""G forward + backward alone at the bench shape (kernel times / PMC of the generator's kernels): python tools/g_check.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 251
torch.manual_seed(0)
G = model.Generator_Conv1D_cLN().cuda()
G.precision = os.environ.get('G_PRECISION', 'bf16')
x = torch.randn(B, T, 64, device='cuda').abs()
y = torch.randn(B, T, 64, device='cuda').abs()
gw = torch.randn(B, T, 64, device='cuda')
def step():
    m = G(x, y)
    (m * gw).sum().backward()
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print('G forward + backward B=%d T=%d (%s): %.2f ms' % (B, T, G.precision, (time.perf_counter() - t0) / 3 * 1e3))

from nele_gan_amd._lib import lib as _l
if hasattr(_l, 'nele_conv1d_prof_read'):               # a -DC1_PROF build: phase clocks of conv1d_tile16_kernel per workgroup (all layers pooled)
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    _l.nele_conv1d_prof_read(buf, 1)
    step(); torch.cuda.synchronize()
    _l.nele_conv1d_prof_read(buf, 1)
    n = max(buf[5], 1)
    print('conv1d_tile16_kernel, shader clocks per workgroup: strip staging %.0f  first weights %.0f  MFMA chunks %.0f  chunk-end store + barrier %.0f  epilogues %.0f  (%d workgroups)' % (
        buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[4] / n, buf[5]))

if hasattr(_l, 'nele_wgrad_tile_prof_read'):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    _l.nele_wgrad_tile_prof_read(buf, 1)
    step(); torch.cuda.synchronize()
    _l.nele_wgrad_tile_prof_read(buf, 1)
    n = max(buf[5], 1)
    print('weight-gradient tile kernel, shader clocks per tile visit: top barrier %.0f  loads + LDS writes %.0f  barrier %.0f  MFMA loop %.0f  (%d visits)' % (buf[4] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[5]))
