"""The canonical step with its main chain on a high-priority stream (metric side streams stay at normal priority): python tools/prio_time.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 'siib&haspi&estoi' if B > 64 else 'siib&estoi'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
def run(stream, label):
    tr = GanTrainer(target_metric=M); tr.D.precision = tr.G.precision = 'bf16'
    ctx = torch.cuda.stream(stream) if stream is not None else None
    if ctx: ctx.__enter__()
    for _ in range(3): tr.canonical_step(cw, nw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): tr.canonical_step(cw, nw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
    if ctx: ctx.__exit__(None, None, None)
    print('B=%d %-28s %.3f ms/step' % (B, label, dt))
    del tr; torch.cuda.empty_cache()
for rep in range(2):
    run(None, 'default stream')
    run(torch.cuda.Stream(priority=-1), 'high-priority main stream')
    run(torch.cuda.Stream(priority=0), 'normal-priority main stream')
