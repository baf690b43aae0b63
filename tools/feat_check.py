"""Feature / resynthesis kernels alone at the bench shape: python tools/feat_check.py [B] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nele_gan_amd import audio_util as au, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
c, v = synth.batch(min(B, 16), L, start=40)
reps = (B + len(c) - 1) // len(c)
x = torch.from_numpy(np.tile(c, (reps, 1))[:B]).cuda(); n = torch.from_numpy(np.tile(v, (reps, 1))[:B]).cuda()
p_power = 0.3 if not hasattr(au, 'P_POWER') else au.P_POWER
alpha = torch.ones(B, 1 + L // 256, 64, device='cuda') * 0.7
def step():
    cs, cb = au.stft_band(x, p_power)
    ns, _ = au.stft_band(n, p_power, want_band=False)
    _, nb = au.imcra_band(ns, p_power)
    w = au.gain_istft(alpha, cs)
    return cs, cb, nb
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print('features B=%d L=%d: %.2f ms' % (B, L, (time.perf_counter() - t0) / 5 * 1e3))
