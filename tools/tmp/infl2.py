import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
from nele_gan_amd import synth, ops
from nele_gan_amd.inference import Enhancer
from nele_gan_amd.train_nele import GanTrainer
tr = GanTrainer('siib&haspi&estoi'); tr.G.precision = tr.D.precision = 'bf16'
mode = os.environ.get('MODE', 'step')
if mode == 'step':
    c, v = synth.batch(64, 64000, start=0)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    for _ in range(2): tr.canonical_step(cw, nw)
    torch.cuda.synchronize()
elif mode == 'streams':          # only create as many idle side streams as the trainer would
    keep = [torch.cuda.Stream() for _ in range(7)]
B, L, K = 128, 128000, 60
c, v = synth.batch(64, L, start=5000)
c = np.tile(c, (2, 1)); v = np.tile(v, (2, 1))
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
enh = Enhancer(G=tr.G); enh.G.precision = 'bf16'
for o in enh.enhance_stream([(cw, nw)] * 5, inflight=4): pass
torch.cuda.synchronize()
dev = enh.device
cur = torch.cuda.current_stream(dev)
print(mode, 'caller shares a queue with slot:', [ops.shares_queue(cur, s, dev) for s in enh._slots], 'slots among themselves:',
      [ops.shares_queue(enh._slots[i], enh._slots[j], dev) for i in range(3) for j in range(i + 1, 3)])
if mode == 'step':
    side = [s for s in tr._all_side_streams()]
    print('trainer side streams vs caller:', [ops.shares_queue(cur, s, dev) for s in side])
for n in (3, 4, 3, 4):
    for o in enh.enhance_stream([(cw, nw)] * (n + 1), inflight=n): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for o in enh.enhance_stream([(cw, nw)] * K, inflight=n): pass
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print('  inflight %d: %.3f ms/batch -> %.0f utt/s' % (n, dt * 1e3, B / dt))
