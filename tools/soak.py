"""Soak run: python tools/soak.py [steps=300] [batch=64] - canonical steps on a different synthetic batch each time (pipelined hand-over of
the next batch), status counters / losses / device memory every 50 steps.  Looks for rare events: eigensolver give-ups, masked optimiser
steps, non-finite values, memory growth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import synth
from nele_gan_amd.train_nele import GanTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tr = GanTrainer('siib&haspi&estoi')
tr.D.precision = tr.G.precision = 'bf16'
pool = []
for k in range(8):
    c, v = synth.batch(B, 64000 - 37 * k, start=1000 * k)          # eight batches of different lengths (buffer sets cycle)
    pool.append((torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()))
t0 = time.perf_counter()
pre = None
for s in range(steps):
    cw, nw = pool[s % 8]
    lg, ld, tgt = tr.canonical_step(cw, nw, pre=pre, next_batch=pool[(s + 1) % 8])
    pre = tr.prefetched                                             # the next batch's input-only work, already in flight
    if (s + 1) % 50 == 0:
        torch.cuda.synchronize()
        st = tr.check_status(raise_on_error=False)
        print('step %4d  %.1f ms/step  loss_g %.4f loss_d %.4f  targets %s  status %s  mem %.2f GB (reserved %.2f)' % (
            s + 1, (time.perf_counter() - t0) / (s + 1) * 1e3, float(lg), float(ld), [round(float(x), 3) for x in tgt.mean(0)],
            {k: v for k, v in st.items() if v}, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
        assert torch.isfinite(lg) and torch.isfinite(ld) and torch.isfinite(tgt).all()
print('done: %d steps, final status %s' % (steps, tr.check_status(raise_on_error=False)))
