import os, sys, time, tempfile, shutil
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from nele_gan_amd import dataio, synth, dist as nd
if os.environ.get('BIND', '0') == '1':
    print('bound', len(nd.bind_to_gpu_numa_node(0) or []))
root = tempfile.mkdtemp(prefix='nele_lp_', dir='/dev/shm')
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
    for keep in (1, 2, 4):
        fb2 = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=8, ahead=2, keep=keep)
        for b in fb2: pass
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for b in fb2: pass
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print('keep', keep, 'ms per pass', [round(t * 1e3, 2) for t in ts], 'decoded', fb2.decoded_files)
        fb2.close()
    import cProfile, pstats
    fb2 = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=8, ahead=2, keep=1)
    for b in fb2: pass
    pr = cProfile.Profile(); pr.enable()
    for b in fb2: pass
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(12)
finally:
    shutil.rmtree(root, ignore_errors=True)
