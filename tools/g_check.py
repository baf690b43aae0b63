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
