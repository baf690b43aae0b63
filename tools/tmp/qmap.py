import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from nele_gan_amd import synth, ops
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = GanTrainer('siib&haspi&estoi'); tr.G.precision = tr.D.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.synchronize()
dev = tr.device
main = torch.cuda.current_stream(dev)
names = ['_side(SIIB)', '_side2(HASPI)', '_fside(feat/ESTOI)', 'D.w0', 'D.w1', 'G.w']
st = [tr._side, tr._side2, tr._fside, tr.D._wstream[0], tr.D._wstream[1], tr.G._wstream]
print('shares the MAIN stream queue:', {n: ops.shares_queue(main, s, dev) for n, s in zip(names, st)})
for i in range(6):
    print(names[i], 'shares with', [names[j] for j in range(6) if j != i and ops.shares_queue(st[i], st[j], dev)])
t0 = time.perf_counter()
for _ in range(10): tr.canonical_step(cw, nw)
torch.cuda.synchronize()
print('step %.2f ms' % ((time.perf_counter() - t0) * 100))
