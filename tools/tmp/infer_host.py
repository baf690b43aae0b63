import sys, time, os, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
from nele_gan_amd import synth
from nele_gan_amd.inference import Enhancer
B, L, K = 128, 128000, 40
c, v = synth.batch(64, L, start=5000)
c = np.tile(c, (2, 1)); v = np.tile(v, (2, 1))
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
enh = Enhancer(); enh.G.precision = 'bf16'
batches = [(cw, nw)] * K
for o in enh.enhance_stream(batches[:6], inflight=3): pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for o in enh.enhance_stream(batches, inflight=3): pass
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('stream: wall %.3f ms/batch, host loop %.3f ms/batch' % (dt / K * 1e3, th / K * 1e3))
pr = cProfile.Profile(); pr.enable()
for o in enh.enhance_stream(batches, inflight=3): pass
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
