"""Small-batch step, plain against pipelined (canonical_step(next_batch=...): the next batch's input-only work at the start of the step on the
second stream / workspace set).  usage: python tools/pipe_time.py [B=32] [metrics=siib&estoi] [steps=20]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = sys.argv[2] if len(sys.argv) > 2 else 'siib&estoi'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tr = GanTrainer(target_metric=M)
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
LEN = int(os.environ.get('PIPE_LEN', '64000'))
c, v = synth.batch(B, LEN, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
for mode in (sys.argv[4].split(',') if len(sys.argv) > 4 else ('plain', 'late', 'early', 'plain', 'early')):
    pre = None
    def one():
        global pre
        if mode == 'plain':
            return tr.canonical_step(cw, nw)
        tr.late_prefetch = 'features' if mode == 'feats' else 'all'
        r = tr.canonical_step(cw, nw, pre=pre, next_batch=(cw, nw), early=(mode == 'early'))
        pre = tr.prefetched
        return r
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): one()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('B=%d L=%d %s %-6s %.3f ms/step (host enqueue %.3f)' % (B, LEN, M, mode, (t2 - t0) / N * 1e3, (t1 - t0) / N * 1e3))
tr.check_status()
