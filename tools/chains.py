import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) if False else sys.path.insert(0, '.')
from nele_gan_amd import synth, metrics as mt
from nele_gan_amd.train_nele import GanTrainer
import os
MET = os.environ.get('CH_METRICS', 'siib&estoi'); BB = int(os.environ.get('CH_B', '32'))
tr = GanTrainer(target_metric=MET)
tr.D.precision = 'bf16'; tr.G.precision = 'bf16'
c, v = synth.batch(BB, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
orig_clean = mt.SiibSplit.clean_part
def clean_part(self):
    mark('B0'); r = orig_clean(self); mark('B_done'); return r
mt.SiibSplit.clean_part = clean_part
orig_deg = mt.SiibSplit.degraded_part
def degraded_part(self, y):
    mark('ypart0'); r = orig_deg(self, y); mark('ypart_done'); return r
mt.SiibSplit.degraded_part = degraded_part
if 'haspi' in MET:
    oc = mt.HaspiSplit.clean_part
    def hclean(self, *a, **k):
        mark('H0'); r = oc(self, *a, **k); mark('Hx_done'); return r
    mt.HaspiSplit.clean_part = hclean
    od = mt.HaspiSplit.degraded_part
    def hdeg(self, *a, **k):
        mark('Hy0'); r = od(self, *a, **k); mark('Hy_done'); return r
    mt.HaspiSplit.degraded_part = hdeg
orig_gen = tr.generate
def generate(*a, **k):
    r = orig_gen(*a, **k); mark('y_ready'); return r
tr.generate = generate
orig_feat = tr.features
def features(*a, **k):
    mark('start'); r = orig_feat(*a, **k); mark('feat_done'); return r
tr.features = features
orig_g = tr.g_step
def g_step(*a, **k):
    r = orig_g(*a, **k); mark('gstep_done'); return r
tr.g_step = g_step
orig_fin = tr._d_finish
def d_finish(score, tgt):
    mark('dfwd_done'); r = orig_fin(score, tgt); mark('end'); return r
tr._d_finish = d_finish
for _ in range(3): tr.canonical_step(cw, nw)
torch.cuda.synchronize(); ev.clear()
t0 = time.perf_counter()
N = 8
for _ in range(N): tr.canonical_step(cw, nw)
torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t0) / N * 1e3)
names = ['B0', 'feat_done', 'B_done', 'gstep_done', 'y_ready', 'ypart0', 'ypart_done', 'dfwd_done', 'end'] + (['H0', 'Hx_done', 'Hy0', 'Hy_done'] if 'haspi' in MET else [])
for n in names:
    ts = [ev['start'][i].elapsed_time(ev[n][i]) for i in range(2, N)]
    print('%-12s %.3f ms (min %.3f max %.3f)' % (n, sum(ts) / len(ts), min(ts), max(ts)))
