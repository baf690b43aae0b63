"""Inference path (inference.py:79-117, BASELINE configs[4]) alone: python tools/infer_time.py [B] [K] [mode] [L]
mode: plain (Enhancer.enhance back to back), stream (Enhancer.enhance_stream), stages (HIP events per stage of one batch)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import synth, audio_util as au, model as M
from nele_gan_amd.inference import Enhancer, p_power, inv_p

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else 'plain'
L = int(sys.argv[4]) if len(sys.argv) > 4 else 128000
c, v = synth.batch(min(B, 64), L, start=5000)
import numpy as np
reps = (B + c.shape[0] - 1) // c.shape[0]
c = np.tile(c, (reps, 1))[:B]; v = np.tile(v, (reps, 1))[:B]
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
enh = Enhancer()
enh.G.precision = os.environ.get('PREC', 'bf16')
for _ in range(3):
    out = enh.enhance(cw, nw)
torch.cuda.synchronize()
if mode == 'plain':
    t0 = time.perf_counter()
    for _ in range(K):
        out = enh.enhance(cw, nw)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('plain B=%d L=%d: %.3f ms/batch (host enqueue %.3f ms) -> %.0f utt/s' % (B, L, dt * 1e3, th / K * 1e3, B / dt))
elif mode == 'stream':
    nfl = int(os.environ.get('INFLIGHT', '3'))
    batches = [(cw, nw)] * K
    for o in enh.enhance_stream(batches[:4], inflight=nfl):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for o in enh.enhance_stream(batches, inflight=nfl):
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('stream(inflight %d) B=%d L=%d: %.3f ms/batch -> %.0f utt/s' % (nfl, B, L, dt * 1e3, B / dt))
    ref = enh.enhance(cw, nw)
    print('bit-identical to enhance():', bool(torch.equal(ref, o)))
else:
    lengths = None
    names = ['stft_clean', 'stft_noise', 'imcra', 'G', 'alpha2', 'gain_istft+post']
    tot = [0.0] * len(names)
    for _ in range(K):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        with torch.no_grad():
            ev[0].record(); clean_spec, clean_band = au.stft_band(cw, p_power)
            ev[1].record(); noise_spec, _ = au.stft_band(nw, p_power, want_band=False)
            ev[2].record(); _, noise_band = au.imcra_band(noise_spec, p_power)
            ev[3].record(); mask = enh.G(clean_band, noise_band)
            ev[4].record(); alpha2 = M.normed_alpha2(mask, clean_band, inv_p)
            ev[5].record(); w = au.gain_istft(alpha2, clean_spec, rms_target=0.030, pcm16=True)
            ev[6].record()
        torch.cuda.synchronize()
        for i in range(len(names)):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
    print('stages B=%d L=%d (ms):' % (B, L), {n: round(t / K, 3) for n, t in zip(names, tot)}, 'sum %.3f' % (sum(tot) / K))
