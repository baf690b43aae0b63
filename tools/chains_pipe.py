"""Event timeline of the pipelined small-batch step (canonical_step(next_batch=...), early mode).  usage: python tools/chains_pipe.py [B=32] [mode=early|late|plain]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import synth, metrics as mt
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else 'early'
tr = GanTrainer(target_metric='siib&estoi')
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
def wrap(obj, attr, before=None, after=None):
    orig = getattr(obj, attr)
    def f(*a, **k):
        if before: mark(before)
        r = orig(*a, **k)
        if after: mark(after)
        return r
    setattr(obj, attr, f)
wrap(mt.SiibSplit, 'clean_part', 'clean0', 'clean_done')
wrap(mt.SiibSplit, 'degraded_part', 'deg0', 'deg_done')
wrap(tr, 'g_step', 'g0', 'g_done')
wrap(tr, 'generate', None, 'y_ready')
wrap(tr, '_d_finish', 'dfwd_done', 'end')
wrap(tr, 'features', 'feat0', 'feat_done')
import nele_gan_amd.audio_util as au
wrap(au, 'imcra_band', None, 'imcra_done')
pre = None
def one():
    global pre
    mark('start')
    if mode == 'plain':
        return tr.canonical_step(cw, nw)
    r = tr.canonical_step(cw, nw, pre=pre, next_batch=(cw, nw), early=(mode == 'early'))
    pre = tr.prefetched
    return r
for _ in range(4): one()
torch.cuda.synchronize(); ev.clear()
N = 8
t0 = time.perf_counter()
for _ in range(N): one()
torch.cuda.synchronize(); print(mode, 'ms/step', (time.perf_counter() - t0) / N * 1e3)
base = ev['start'][0]
for i in range(2, 5):
    row = []
    for n, lst in ev.items():
        for e in lst:
            t = ev['start'][i].elapsed_time(e)
            if -0.5 < t < 9.0:
                row.append((t, n))
    print('step %d: ' % i + '  '.join('%s %.2f' % (n, t) for t, n in sorted(row)))
