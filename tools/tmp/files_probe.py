import os, shutil, sys, tempfile, time
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from nele_gan_amd import dataio, synth, inference
from nele_gan_amd.inference import Enhancer, enhance_files
from nele_gan_amd.train_nele import GanTrainer
n_utt, batch = 2048, 128
root = tempfile.mkdtemp(prefix='nele_fs_', dir='/dev/shm')
acc = {'stage_wait': 0.0, 'stage_rest': 0.0, 'getitem': 0.0, 'read': 0.0, 'nread': 0}
orig_stage = dataio.FileBatches._stage
def stage(self, g):
    t0 = time.perf_counter()
    kind, futs = self._pending[g][0], self._pending[g][1]
    if kind == 'i16':
        futs.result()
    t1 = time.perf_counter()
    orig_stage(self, g)
    acc['stage_wait'] += t1 - t0; acc['stage_rest'] += time.perf_counter() - t1
dataio.FileBatches._stage = stage
orig_get = dataio.FileBatches.__getitem__
def getitem(self, g):
    t0 = time.perf_counter(); r = orig_get(self, g); acc['getitem'] += time.perf_counter() - t0; return r
dataio.FileBatches.__getitem__ = getitem
orig_read = dataio.read_wav_batch_pcm16
def rd(paths, arr, threads):
    t0 = time.perf_counter(); r = orig_read(paths, arr, threads); acc['read'] += time.perf_counter() - t0; acc['nread'] += 1; return r
dataio.read_wav_batch_pcm16 = rd
try:
    c, v = synth.batch(64, 128000, start=70000)
    os.makedirs(root + '/Clean'); os.makedirs(root + '/Noise')
    rs = np.random.RandomState(1)
    files = []
    for i in range(n_utt):
        k, L = i % 64, int(rs.randint(112000, 128001))
        dataio.write_wav_pcm16('%s/Clean/u%05d.wav' % (root, i), c[k, :L]); dataio.write_wav_pcm16('%s/Noise/u%05d.wav' % (root, i), v[(k + i // 64) % 64, :L])
        files.append('%s/Clean/u%05d.wav' % (root, i))
    tr = GanTrainer('siib&estoi'); tr.G.precision = 'bf16'
    enh = Enhancer(G=tr.G); enh.G.precision = 'bf16'
    enhance_files(enh, files, root + '/Noise/', root + '/Warm', batch=batch, workers=8)
    for k in acc: acc[k] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enhance_files(enh, files, root + '/Noise/', root + '/Enh', batch=batch, workers=8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nb = n_utt // batch
    print('total %.1f ms = %.2f ms per batch (%d batches); main thread: getitem %.2f (waiting for the reader %.2f, staging %.2f) ms per batch; a read call (128 files) %.2f ms, %d calls' % (
        dt * 1e3, dt * 1e3 / nb, nb, acc['getitem'] * 1e3 / nb, acc['stage_wait'] * 1e3 / nb, acc['stage_rest'] * 1e3 / nb, acc['read'] * 1e3 / max(1, acc['nread']), acc['nread']))
    # the GPU part alone on resident batches, same shapes
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    enhance_files(enh, files, root + '/Noise/', root + '/Enh2', batch=batch, workers=8)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
finally:
    shutil.rmtree(root, ignore_errors=True)
