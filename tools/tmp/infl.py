import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
from nele_gan_amd import synth
from nele_gan_amd.inference import Enhancer
from nele_gan_amd.train_nele import GanTrainer
tr = GanTrainer('siib&haspi&estoi'); tr.G.precision = tr.D.precision = 'bf16'
if os.environ.get('STEP', '1') == '1':
    c, v = synth.batch(64, 64000, start=0)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    for _ in range(2): tr.canonical_step(cw, nw)
    torch.cuda.synchronize()
B, L, K = 128, 128000, 60
c, v = synth.batch(64, L, start=5000)
c = np.tile(c, (2, 1)); v = np.tile(v, (2, 1))
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
enh = Enhancer(G=tr.G); enh.G.precision = 'bf16'
for n in (3, 4, 3, 4, 4, 3):
    for o in enh.enhance_stream([(cw, nw)] * (n + 1), inflight=n): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for o in enh.enhance_stream([(cw, nw)] * K, inflight=n): pass
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print('inflight %d: %.3f ms/batch -> %.0f utt/s' % (n, dt * 1e3, B / dt))
