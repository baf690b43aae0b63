"""Epoch soak: python tools/soak_epoch.py [epochs=40] [cache=0|1] - GanTrainer.run_epoch from wav files (192 utterances of 2.5 - 4 s, batches of 64,
samples written; cache=1: with enable_clean_cache()): time per epoch, status counters, device memory, buffer-shape counts every 5 epochs."""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nele_gan_amd import dataio, synth
from nele_gan_amd.train_nele import GanTrainer
root = tempfile.mkdtemp(prefix='nele_se_', dir='/dev/shm')
try:
    n_utt, batch = 192, 64
    c, v = synth.batch(n_utt, 64000, start=40000)
    rs = np.random.RandomState(0)
    os.makedirs(root + '/Clean'); os.makedirs(root + '/Noise')
    files = []
    for i in range(n_utt):
        L = int(rs.randint(40000, 64001))
        dataio.write_wav_pcm16('%s/Clean/u%04d.wav' % (root, i), c[i, :L]); dataio.write_wav_pcm16('%s/Noise/u%04d.wav' % (root, i), v[i, :L])
        files.append('%s/Clean/u%04d.wav' % (root, i))
    tr = GanTrainer('siib&haspi&estoi'); tr.D.precision = tr.G.precision = 'bf16'
    n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cache = tr.enable_clean_cache() if (len(sys.argv) > 2 and sys.argv[2] == '1') else None
    fb = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=8, ahead=2, keep=2)
    t0 = time.perf_counter()
    for ep in range(1, n_ep + 1):
        res = tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
        if ep % 5 == 0:
            torch.cuda.synchronize()
            print('epoch %2d  %.1f ms/epoch  g_loss %s  d_steps %d  history %d  status %s  mem %.2f GB (reserved %.2f)  shapes G %d D %d' % (
                ep, (time.perf_counter() - t0) / ep * 1e3, None if res['g_loss'] is None else round(float(res['g_loss']), 4), res['d_steps'], len(tr.history),
                {k: v for k, v in res['status'].items() if v}, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, len(tr.G._bufs), len(tr.D._bufs)), flush=True)
    fb.close()
    print('files written', sum(len(f) for _, _, f in os.walk(root + '/out')), 'cache', None if cache is None else cache.stats())
finally:
    shutil.rmtree(root, ignore_errors=True)
