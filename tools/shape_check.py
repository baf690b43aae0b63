import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from nele_gan_amd import model
torch.manual_seed(1)
def grads(G, x, y, gw):
    for p in G.parameters():
        if p.grad is not None: p.grad.zero_()
    m = G(x, y); (m * gw).sum().backward()
    return m.detach().clone(), {k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None}
for B, T in [(1, 33), (3, 70), (5, 251), (2, 129), (7, 40), (4, 64)]:
    G = model.Generator_Conv1D_cLN().cuda()
    x = torch.randn(B, T, 64, device='cuda').abs(); y = torch.randn(B, T, 64, device='cuda').abs(); gw = torch.randn(B, T, 64, device='cuda')
    G.precision = 'f32'; m0, g0 = grads(G, x, y, gw)
    G.precision = 'bf16'; m1, g1 = grads(G, x, y, gw)
    worst = 0.0
    for k in g0:
        d = (g0[k] - g1[k]).abs().max().item(); s = g0[k].abs().max().item() + 1e-12
        worst = max(worst, d / s)
    print('G B=%d T=%d: mask rel diff %.2e, worst grad rel diff %.2e' % (B, T, ((m0 - m1).abs().max() / m0.abs().max()).item(), worst))
for B, T in [(1, 52), (3, 70), (2, 251), (2, 129), (5, 60)]:
    D = model.Discriminator().cuda(); D.train()
    din = torch.randn(B, 64, T, 4, device='cuda').abs().requires_grad_(True)
    res = []
    for prec in ('f32', 'bf16'):
        D.precision = prec
        for p in D.parameters():
            if p.grad is not None: p.grad.zero_()
        if din.grad is not None: din.grad.zero_()
        s = D.forward_packed(din); s.pow(2).sum().backward()
        res.append((s.detach().clone(), din.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in D.named_parameters() if p.grad is not None}))
    (s0, d0, g0), (s1, d1, g1) = res
    worst = max(((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-12)).item() for k in g0)
    print('D B=%d T=%d: score rel diff %.2e, din grad rel diff %.2e, worst weight grad rel diff %.2e' % (B, T, ((s0 - s1).abs().max() / s0.abs().max()).item(), ((d0 - d1).abs().max() / d0.abs().max()).item(), worst))
