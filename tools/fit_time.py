#!/usr/bin/env python
"""GanTrainer.fit - the reference script's outer loop (train_nele.py:110-428) - on a synthetic corpus laid out like the reference's folders:
N training triples (Clean / Noise / MultiEnh) of 3 - 4 s, V validation pairs; per epoch a shuffled draw of S training files, G-steps,
validation scoring, checkpoint, sample generation, true targets of the generated and the pre-enhanced examples, three D passes + replay.
Seconds per epoch with and without the clean-signal cache (enable_clean_cache: epochs after the first reuse every clean-file-only result).
usage: python tools/fit_time.py [N=360] [S=150] [V=96] [batch=32] [epochs=8]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nele_gan_amd import dataio, synth, dist as nd
from nele_gan_amd.train_nele import GanTrainer

N, S, V, batch, epochs = [int(sys.argv[k]) if len(sys.argv) > k else d for k, d in ((1, 360), (2, 150), (3, 96), (4, 32), (5, 8))]
nd.bind_to_gpu_numa_node(0)
root = tempfile.mkdtemp(prefix='nele_fit_', dir='/dev/shm')
try:
    rs = np.random.RandomState(0)
    for sub in ('Train/Clean', 'Train/Noise', 'Train/MultiEnh', 'Test/Clean', 'Test/Noise'):
        os.makedirs(os.path.join(root, sub))
    c, v = synth.batch(64, 64000, start=40000)
    for i in range(N + V):
        k, L = i % 64, int(rs.randint(48000, 64001))
        part = 'Train' if i < N else 'Test'
        dataio.write_wav_pcm16('%s/%s/Clean/u%04d.wav' % (root, part, i), np.roll(c[k], 37 * (i // 64))[:L])
        dataio.write_wav_pcm16('%s/%s/Noise/u%04d.wav' % (root, part, i), np.roll(v[k], 91 * (i // 64))[:L])
        if i < N:
            dataio.write_wav_pcm16('%s/Train/MultiEnh/u%04d.wav' % (root, i), (1.4 * np.roll(c[k], 37 * (i // 64))[:L]).astype(np.float32))
    train = sorted(dataio.get_filepaths(root + '/Train/Clean/'))
    test = sorted(dataio.get_filepaths(root + '/Test/Clean/'))
    for cache in (False, True):
        tr = GanTrainer('siib&haspi&estoi', seed=666)
        tr.G.precision = tr.D.precision = 'bf16'
        ts = []

        def tick(res):
            torch.cuda.synchronize()
            ts.append(time.perf_counter())
        t0 = time.perf_counter()
        tr.fit(train, root + '/Train/Noise/', test, root + '/Test/Noise/', train_enh_path=root + '/Train/MultiEnh/', epochs=epochs, sampling=S, valid_samples=V,
               batch=batch, output_path=root + '/out%d' % cache, pt_dir=root + '/ck%d' % cache, log_path=root + '/log%d.txt' % cache, clean_cache=cache, on_epoch=tick)
        per = np.diff([t0] + ts) * 1e3
        print('clean cache %-5s: ms per epoch %s; last %d epochs %.0f ms = %.0f drawn utterances/s (+ %d validation utterances scored per epoch)%s' % (
            cache, ' '.join('%.0f' % p for p in per), len(per) // 2, per[len(per) // 2:].mean(), S / per[len(per) // 2:].mean() * 1e3, V,
            '' if not cache else '; cache %s' % {k_: v_ for k_, v_ in tr.clean_cache.stats().items() if k_ in ('hits', 'misses', 'partial', 'stored', 'bytes')}))
finally:
    shutil.rmtree(root, ignore_errors=True)
