#!/usr/bin/env python
"""Does the loop learn?  The reference's only health signal is its learning curve (train_nele.py:159-225: mean raw SIIB / HASPI / ESTOI of
the validation set per GAN epoch, written to log.txt and plotted).  This tool runs ``GanTrainer.run_epoch`` for E epochs on a fixed
synthetic corpus (nele_gan_amd.synth: seeded, RMS 0.03, SNR -11 .. -1 dB), same seed, once per precision, and records per epoch

  * ``valid``       mean raw (unmapped) SIIB [bit/s] / HASPI / ESTOI of the validation utterances enhanced by the current G (:200-225);
  * ``valid_mapped`` the same scores through the reference's logistic maps (intel.py:102-139: the targets D predicts and G pushes to 1)
                    and ``objective`` = mean over metrics of (1 - mapped mean)^2 - what the G-step minimises, measured with the TRUE metrics;
  * ``g_loss``      mean G loss of the epoch's G-steps, MSE(D(G(x)), 1) (:149);
  * ``d_mse_fresh`` D's mean squared error on this epoch's newly generated samples BEFORE it trains on them (does D predict the true
                    scores of samples it has not seen?), ``d_mse_fit`` the same after the epoch's three passes (:342-426);
  * ``target_mean`` mean mapped true targets of the epoch's generated training samples.

usage: python tools/learn_curve.py [--epochs 30] [--utts 256] [--valid 64] [--batch 8] [--length 63871] [--precisions f32,bf16]
                                   [--metrics 'siib&haspi&estoi'] [--out profiles/r06/learn_curve.json]
``unprocessed`` in the output = the validation metrics of clean + noise with no enhancement (the curve's reference level)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nele_gan_amd import synth                                   # noqa: E402
from nele_gan_amd.train_nele import GanTrainer                   # noqa: E402


def spearman(y):
    """Spearman rank correlation of a series with its index (trend: > 0 = rising)."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    if n < 3 or np.all(y == y[0]):
        return 0.0
    ry = np.argsort(np.argsort(y)).astype(np.float64)
    rx = np.arange(n, dtype=np.float64)
    return float(np.corrcoef(rx, ry)[0, 1])


def batches_of(c, v, batch, first_id):
    out = []
    for k in range(0, c.shape[0], batch):
        out.append({'clean': c[k:k + batch].contiguous(), 'noise': v[k:k + batch].contiguous(),
                    'ids': torch.arange(first_id + k, first_id + k + c[k:k + batch].shape[0], dtype=torch.int64, device=c.device)})
    return out


MAPS = {'siib': (0.06, 32.0), 'haspi': (0.95, 2.8), 'estoi': (8.0, 0.25)}


def objective(mapped):
    return float(np.mean([(1.0 - v) ** 2 for v in mapped.values()]))


def run(precision, args, train, valid, seed, log=print):
    tr = GanTrainer(args.metrics, seed=seed)
    tr.G.precision = tr.D.precision = precision
    L = valid[0]['clean'].shape[1]
    Lr = 256 * (L // 256)
    base_all = torch.cat([tr.true_metrics(b['clean'], b['clean'][:, :Lr].contiguous(), b['noise'], norm=False, utt_ids=b['ids']) for b in valid]).double()
    base = base_all.mean(dim=0).tolist()
    base_mapped = {m: float((1.0 / (1.0 + torch.exp(-MAPS[m][0] * (base_all[:, i] - MAPS[m][1])))).mean()) for i, m in enumerate(tr.metrics)}
    curve = []
    t0 = time.perf_counter()
    for ep in range(1, args.epochs + 1):
        res = tr.run_epoch(ep, train, valid, d_batch=args.batch, d_eval=True)
        row = {'epoch': ep, 'valid': res['valid'], 'valid_mapped': res['valid_mapped'], 'objective': objective(res['valid_mapped']), 'g_loss': None if res['g_loss'] is None else float(res['g_loss']),
               'd_mse_fresh': res['d_mse_fresh'], 'd_mse_fit': res['d_mse_fit'], 'target_mean': res['target_mean'],
               'g_steps': res['g_steps'], 'd_steps': res['d_steps'], 'status': {k: int(v) for k, v in res['status'].items() if v}}
        curve.append(row)
        log('%s epoch %2d  valid %s  objective %.4f  g_loss %s  d_mse fresh %.5f fit %.5f  targets %s' % (
            precision, ep, {k: round(v, 4) for k, v in row['valid'].items()}, row['objective'], None if row['g_loss'] is None else round(row['g_loss'], 5),
            row['d_mse_fresh'], row['d_mse_fit'], [round(t, 4) for t in row['target_mean']]))
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    ms = tr.metrics
    tail = curve[-max(1, len(curve) // 6):]
    summ = {'seconds': secs, 'seed': seed, 'unprocessed': dict(zip(ms, base)), 'unprocessed_mapped': base_mapped, 'unprocessed_objective': objective(base_mapped),
            'objective_first': curve[0]['objective'], 'objective_tail_mean': float(np.mean([r['objective'] for r in tail])),
            'objective_spearman': spearman([r['objective'] for r in curve]),
            'd_mse_fresh_first': curve[0]['d_mse_fresh'], 'd_mse_fresh_last': curve[-1]['d_mse_fresh'],
            'd_mse_fresh_ratio': curve[-1]['d_mse_fresh'] / curve[0]['d_mse_fresh'],
            'valid_first': curve[0]['valid'], 'valid_last': curve[-1]['valid'],
            'valid_best': {m: max(r['valid'][m] for r in curve) for m in ms},
            'spearman': {m: spearman([r['valid'][m] for r in curve]) for m in ms}}
    return {'precision': precision, 'seed': seed, 'summary': summ, 'curve': curve}


def compare(a, b, metrics):
    """Largest |difference| between two runs' validation curves (per metric, and of the objective)."""
    oa, ob = np.array([r['objective'] for r in a['curve']]), np.array([r['objective'] for r in b['curve']])
    out = {'objective': {'max_abs_diff': float(np.abs(oa - ob).max()), 'tail_mean_diff': float(abs(a['summary']['objective_tail_mean'] - b['summary']['objective_tail_mean']))}}
    for m in metrics:
        ya = np.array([r['valid'][m] for r in a['curve']])
        yb = np.array([r['valid'][m] for r in b['curve']])
        rng = max(float(ya.max() - ya.min()), 1e-12)
        out[m] = {'max_abs_diff': float(np.abs(ya - yb).max()), 'f32_range': rng, 'max_rel_to_level': float((np.abs(ya - yb) / np.abs(ya)).max()),
                  'last_diff_rel': float(abs(ya[-1] - yb[-1]) / abs(ya[-1]))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--utts', type=int, default=256)
    ap.add_argument('--valid', type=int, default=64)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--length', type=int, default=63871)
    ap.add_argument('--precisions', default='f32,bf16')
    ap.add_argument('--metrics', default='siib&haspi&estoi')
    ap.add_argument('--seed', type=int, default=666)
    ap.add_argument('--f32-second-seed', type=int, default=None, help='also run f32 with this seed: the seed-to-seed band the bf16 curve is compared with')
    ap.add_argument('--noise-tilt', type=float, default=0.5, help='noise spectrum 1/f^tilt (0.5 = the bench recipe, same long-term spectrum as the speech; 1.5 = low-pass noise)')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    c, v = synth.batch(args.utts, args.length, start=0, noise_tilt=args.noise_tilt)
    cv, vv = synth.batch(args.valid, args.length, start=100000, noise_tilt=args.noise_tilt)
    dev = 'cuda'
    train = batches_of(torch.from_numpy(c).to(dev), torch.from_numpy(v).to(dev), args.batch, 0)
    valid = batches_of(torch.from_numpy(cv).to(dev), torch.from_numpy(vv).to(dev), max(args.batch, 64), 100000)
    runs = [run(p, args, train, valid, args.seed) for p in args.precisions.split(',')]
    other = run('f32', args, train, valid, args.f32_second_seed) if args.f32_second_seed is not None else None
    doc = {'tool': 'tools/learn_curve.py', 'args': vars(args), 'device': torch.cuda.get_device_name(0),
           'note': 'synthetic corpus (nele_gan_amd.synth), fixed seed; valid = mean RAW metrics of the validation set per epoch (train_nele.py:200-225); '
                   'SIIB / ESTOI parity is of the build\'s oracle (unpinned), see DESIGN.md section 2',
           'runs': runs}
    by = {r['precision']: r for r in runs}
    if 'f32' in by and 'bf16' in by:
        doc['bf16_vs_f32'] = compare(by['f32'], by['bf16'], list(by['f32']['curve'][0]['valid'].keys()))
    if other is not None and 'f32' in by:
        doc['f32_other_seed'] = other
        doc['f32_seed_vs_seed'] = compare(by['f32'], other, list(by['f32']['curve'][0]['valid'].keys()))
        print('f32_seed_vs_seed', json.dumps(doc['f32_seed_vs_seed']))
    for r in runs:
        print(r['precision'], json.dumps(r['summary']))
    if 'bf16_vs_f32' in doc:
        print('bf16_vs_f32', json.dumps(doc['bf16_vs_f32']))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as fh:
            json.dump(doc, fh, indent=1)
        print('wrote', args.out)


if __name__ == '__main__':
    main()
