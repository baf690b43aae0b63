"""Which stage of the canonical step can ROCm's stream capture take?  Each stage is captured in its OWN process (a crash inside
hipStreamEndCapture takes the process down).  python tools/graph_probe2.py <stage> [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nele_gan_amd import synth, ops, metrics as mt
from nele_gan_amd.train_nele import GanTrainer
stage = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tr = GanTrainer('siib&estoi')
tr.D.precision = tr.G.precision = 'bf16'
c, v = synth.batch(B, 64000, start=0)
cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
f = tr.features(cw, nw)
enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
tgt = tr.true_metrics(cw, enh, nw)
din = tr.d_inputs(enh, f['noise_band'], f['clean_band'])
def run():
    if stage == 'features': return tr.features(cw, nw)['clean_band']
    if stage == 'stft': 
        from nele_gan_amd import audio_util as au
        return au.stft_band(cw, 1 / 6)[1]
    if stage == 'g_step': return tr.g_step(f['clean_band'], f['noise_band'])
    if stage == 'g_fwd':
        with torch.no_grad(): return tr.G(f['clean_band'], f['noise_band'])
    if stage == 'generate': return tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
    if stage == 'estoi': return mt.batch_estoi(cw[:, :enh.shape[1]].contiguous(), enh)[1]
    if stage == 'siib': return mt.batch_siib(cw[:, :enh.shape[1]].contiguous(), enh)[1]
    if stage == 'd_step': return tr.d_step(din, tgt)
    if stage == 'd_fwd':
        with torch.no_grad(): return tr.D.forward_packed(din)
    if stage == 'adam': tr.optimizer_d.step(); return tr.D.flat_parameters().flat
    if stage == 'torch': return (cw * 2 + nw).sum()
    if stage.startswith('step'): return tr.canonical_step(cw, nw)[2]
    if stage.startswith('v'):
        lvl = int(stage[1:])
        main = torch.cuda.current_stream()
        if tr._side is None: tr._side = ops.side_stream(tr.device)
        if tr._side2 is None: tr._side2 = ops.side_stream(tr.device)
        side, side2 = tr._side, tr._side2
        ff = tr.features(cw, nw)
        lg = tr.g_step(ff['clean_band'], ff['noise_band'])
        out = lg
        if lvl >= 2:
            e2 = tr.generate(ff['clean_band'], ff['noise_band'], ff['clean_spec'])
            out = e2
        if lvl >= 3:
            L = e2.shape[1]
            x = cw[:, :L].contiguous()
            ready = torch.cuda.Event(); ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                y = (e2 + nw[:, :L]).contiguous()
                y_ready = torch.cuda.Event(); y_ready.record(side)
            with torch.cuda.stream(side2):
                side2.wait_event(y_ready)
                col = mt.batch_estoi(x, y)[1]
                others = torch.cuda.Event(); others.record(side2)
            if lvl in (3, 4):
                with torch.cuda.stream(side):
                    side.wait_event(others)
                    tg = torch.stack([col, col], dim=1)
                    done = torch.cuda.Event(); done.record(side)
            else:
                done = others
                main.wait_event(others)
                tg = torch.stack([col, col], dim=1)
            out = tg
            if lvl == 3:
                main.wait_event(done)
        if lvl >= 4:
            dn = tr.d_inputs(e2, ff['noise_band'], ff['clean_band'])
            tr.optimizer_d.zero_grad()
            score = tr.D.forward_packed(dn)
            main.wait_event(done)
            out = tr._d_finish(score, tg)
        tr._join_side_streams(main)
        return out
    if stage == 'input_only':
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        w = tr._input_only_work(cw, nw, None, ev, with_features=False)
        tr._join_side_streams(torch.cuda.current_stream())
        return w['x']
    raise SystemExit('unknown stage')
if 'norecord' in stage:
    torch.Tensor.record_stream = lambda self, st: None
if 'nostatus' in stage:
    tr._note_status = lambda *a, **k: None
if 'noprep' in stage:
    tr.D.prepare = lambda *a, **k: None
if 'nosplit' in stage:
    tr.metrics = ['estoi']
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): run()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(stage, 'warm', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
print(stage, 'captured', flush=True)
g.replay(); torch.cuda.synchronize()
print(stage, 'replayed ok, finite', bool(torch.isfinite(out.float()).all()), flush=True)
