"""The two from-files figures of bench.py on their own: python tools/files_time.py [n_infer=4096] [n_epoch=256]
(inference_from_files: wav -> enhanced wav at 8 s per utterance; epoch_from_files: run_epoch fed by dataio.FileBatches)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
import torch
from nele_gan_amd.train_nele import GanTrainer
n_inf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tr = GanTrainer('siib&haspi&estoi')
tr.D.precision = tr.G.precision = 'bf16'
if n_inf:
    for k in range(2):
        print('inference_from_files', json.dumps(bench.inference_from_files(tr, n_utt=n_inf)), flush=True)
if n_ep:
    a = argparse.Namespace(metrics='siib&haspi&estoi', precision='bf16')
    print('epoch_from_files', json.dumps(bench.epoch_from_files(a, n_utt=n_ep)), flush=True)
