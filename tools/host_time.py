import sys, time, torch
sys.path.insert(0, '.')
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
import nele_gan_amd._lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = GanTrainer(target_metric='siib&estoi')
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.synchronize()
# count launches
n = [0]
orig = L.call
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.canonical_step(cw, nw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.2f ms, total %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(4): tr.canonical_step(cw, nw)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
