"""Which of the trainer's side streams share a hardware queue (GanTrainer._shares_queue)?  python tools/queue_map.py [B=64] [metrics]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = sys.argv[2] if len(sys.argv) > 2 else 'siib&haspi&estoi'
tr = GanTrainer(target_metric=M)
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.synchronize()
st = {'main': torch.cuda.current_stream(), 'side': tr._side, 'side2': tr._side2, 'fside': tr._fside, 'Gw': tr.G._wstream, 'Dw0': tr.D._wstream[0], 'Dw1': tr.D._wstream[1]}
names = list(st)
print('shares a queue with:')
for a in names:
    print('%-6s' % a, ' '.join(b for b in names if b != a and tr._shares_queue(st[a], st[b])))
