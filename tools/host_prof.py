"""Host-side cost of one canonical step (launch-bound regime, B = 32): cProfile by own time.  usage: python tools/host_prof.py [B] [metrics]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = sys.argv[2] if len(sys.argv) > 2 else 'siib&estoi'
tr = GanTrainer(target_metric=M)
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.synchronize()
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
N = 8
for _ in range(N): tr.canonical_step(cw, nw)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
print('--- per %d steps, by own time' % N)
st.sort_stats('tottime').print_stats(45)
print('--- by cumulative time')
st.sort_stats('cumulative').print_stats(60)
