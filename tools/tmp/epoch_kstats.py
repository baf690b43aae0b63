import os, sys, time, tempfile, shutil
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from nele_gan_amd import dataio, synth, dist as nd
from nele_gan_amd.train_nele import GanTrainer
root = tempfile.mkdtemp(prefix='nele_ep_', dir='/dev/shm')
cached = len(sys.argv) > 1 and sys.argv[1] == '1'
try:
    n_utt, batch = 256, 64
    c, v = synth.batch(n_utt, 64000, start=40000)
    rs = np.random.RandomState(0)
    os.makedirs(root + '/Clean'); os.makedirs(root + '/Noise')
    files = []
    for i in range(n_utt):
        L = int(rs.randint(48000, 64001))
        dataio.write_wav_pcm16('%s/Clean/u%04d.wav' % (root, i), c[i, :L]); dataio.write_wav_pcm16('%s/Noise/u%04d.wav' % (root, i), v[i, :L])
        files.append('%s/Clean/u%04d.wav' % (root, i))
    tr = GanTrainer('siib&haspi&estoi'); tr.D.precision = tr.G.precision = 'bf16'
    fb = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=8, ahead=2, keep=2)
    if cached:
        tr.enable_clean_cache()
    t0 = time.perf_counter()
    for ep in range(2, 12):
        tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
    torch.cuda.synchronize()
    print('10 epochs (cached=%s): %.1f ms each incl. warm-up' % (cached, (time.perf_counter() - t0) * 100))
finally:
    shutil.rmtree(root, ignore_errors=True)
