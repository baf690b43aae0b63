"""HASPI raw scores (float64 print) and call time: python tools/haspi_ab.py B [L]; run under different NELE_HASPI_* switches and compare"""
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
raw, mapped, info = mt.batch_haspi(x, y, return_info=True)
torch.cuda.synchronize()
print('scores', ' '.join('%.9f' % t for t in raw[:8].double().cpu().numpy()))
t0 = time.perf_counter()
for _ in range(5):
    mt.batch_haspi(x, y)
torch.cuda.synchronize()
print('B=%d L=%d: %.2f ms per call (%s)' % (B, L, (time.perf_counter() - t0) / 5 * 1e3, ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('NELE_HASPI'))))
