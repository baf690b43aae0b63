import os, sys, time, tempfile, shutil, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from nele_gan_amd import dataio, synth, dist as nd
from nele_gan_amd.train_nele import GanTrainer
nd.bind_to_gpu_numa_node(0)
root = tempfile.mkdtemp(prefix='nele_ep_', dir='/dev/shm')
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
    tr.enable_clean_cache()
    for ep in (2, 3, 4, 5):
        tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ep in (6, 7, 8, 9):
        tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
    torch.cuda.synchronize()
    print('cached epoch %.2f ms' % ((time.perf_counter() - t0) / 4 * 1e3))
    # host time only: enqueue without waiting
    pr = cProfile.Profile(); pr.enable()
    for ep in (10, 11):
        tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
finally:
    shutil.rmtree(root, ignore_errors=True)
