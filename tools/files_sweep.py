#!/usr/bin/env python
"""inference.enhance_files on a synthetic corpus of 8 s wav files on tmpfs (bench.py: inference_from_files) under different host-side
settings: library threads per batch call, batches read ahead, with / without writing the output files.  Which stage bounds the file path?"""
import os, shutil, sys, tempfile, time
if os.environ.get('NUMA_LOCAL', '0') == '1':          # A/B: pin the process to the CPUs of the GPU's NUMA node before anything is allocated
    import glob
    import torch as _t
    _p = _t.cuda.get_device_properties(0)
    _node = int(open('/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node' % (_p.pci_domain_id, _p.pci_bus_id, _p.pci_device_id)).read())
    _cpus = set()
    for _part in open('/sys/devices/system/node/node%d/cpulist' % _node).read().strip().split(','):
        _a, _, _b = _part.partition('-')
        _cpus |= set(range(int(_a), int(_b or _a) + 1))
    os.sched_setaffinity(0, _cpus)
    print('pinned to NUMA node', _node, len(_cpus), 'cpus')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nele_gan_amd import dataio, synth
from nele_gan_amd.inference import Enhancer, enhance_files
from nele_gan_amd.train_nele import GanTrainer

n_utt, batch = 2048, 128
root = tempfile.mkdtemp(prefix='nele_fs_', dir='/dev/shm')
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
    shutil.rmtree(root + '/Warm', ignore_errors=True)
    for workers, ahead, inflight, write in ((8, 2, 3, True), (8, 2, 4, True), (8, 2, 2, True), (8, 2, 3, True), (8, 2, 4, True), (8, 2, 4, False)):
        best = 0.0
        for rep in range(3):
            shutil.rmtree(root + '/Enh', ignore_errors=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enhance_files(enh, files, root + '/Noise/', root + '/Enh', batch=batch, workers=workers, ahead=ahead, inflight=inflight, write=write)
            torch.cuda.synchronize()
            best = max(best, n_utt / (time.perf_counter() - t0))
        print('workers %2d ahead %d inflight %d write %-5s: %.0f utterances/s' % (workers, ahead, inflight, write, best), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
