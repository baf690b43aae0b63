"""Generator forward + backward alone at the bench shape: python tools/g_time.py [B] [T] [precision]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import model as M
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 251
prec = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
torch.manual_seed(0)
G = M.Generator_Conv1D_cLN().cuda()
G.precision = prec
x, y, gw = torch.rand(B, T, 64, device='cuda'), torch.rand(B, T, 64, device='cuda'), torch.randn(B, T, 64, device='cuda')
def step():
    G.flat_parameters().grad.zero_()
    m = G(x, y)
    (m * gw).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
K = 10
for _ in range(K):
    G.flat_parameters().grad.zero_()
    ev[0].record(); m = G(x, y); ev[1].record(); (m * gw).sum().backward(); ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print('G %s B=%d T=%d: forward %.3f ms, backward %.3f ms, total %.3f ms' % (prec, B, T, tf / K, tb / K, (tf + tb) / K))
with torch.no_grad():
    G.eval()
    for _ in range(3): G(x, y)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): G(x, y)
    e1.record(); torch.cuda.synchronize()
    print('  eval forward %.3f ms' % (e0.elapsed_time(e1) / K))
