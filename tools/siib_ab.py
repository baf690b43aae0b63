"""SIIB raw scores and call time: python tools/siib_ab.py B [L]  (run once with NELE_SIIB_LAG=0 and once without; compare the printed scores)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nele_gan_amd import metrics as mt, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
c, v = synth.batch(min(B, 16), L, start=40)
reps = (B + len(c) - 1) // len(c)
c = np.tile(c, (reps, 1))[:B]; v = np.tile(v, (reps, 1))[:B]
x = torch.from_numpy(c).cuda(); y = torch.from_numpy(c * 0.7 + v).cuda()
raw, mapped, info = mt.batch_siib(x, y, return_info=True)
torch.cuda.synchronize()
print('scores', ' '.join('%.10f' % t for t in raw[:8].double().cpu().numpy()))
print('info', info[:2].cpu().numpy().tolist())
t0 = time.perf_counter()
for _ in range(5):
    mt.batch_siib(x, y)
torch.cuda.synchronize()
print('B=%d L=%d: %.2f ms per call (NELE_SIIB_LAG=%s)' % (B, L, (time.perf_counter() - t0) / 5 * 1e3, os.environ.get('NELE_SIIB_LAG', '1')))
