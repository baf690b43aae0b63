"""Feasibility probe: capture one canonical step in a HIP graph (torch.cuda.graph) and replay it.  python tools/graph_probe.py [B] [metrics]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
metrics = sys.argv[2] if len(sys.argv) > 2 else 'siib&estoi'
tr = GanTrainer(metrics)
tr.D.precision = tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): tr.canonical_step(cw, nw)
print('eager: %.3f ms/step' % timeit(lambda: tr.canonical_step(cw, nw)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = tr.canonical_step(cw, nw)
    print('captured')
    g.replay(); torch.cuda.synchronize()
    print('replayed: losses', float(out[0]), float(out[1]), 'targets finite', bool(torch.isfinite(out[2]).all()))
    print('graph: %.3f ms/step' % timeit(g.replay))
except Exception as e:
    import traceback; traceback.print_exc()
    print('CAPTURE FAILED:', type(e).__name__, str(e)[:500])
