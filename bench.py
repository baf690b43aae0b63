#!/usr/bin/env python
"""Headline benchmark: utterances/s per GAN_epoch step (G + D + metric loss) on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one canonical GAN_epoch step over one batch of synthetic utterances already
resident in HBM: features(clean, noise) -> G-step (G fwd+bwd through D, Adam-G) -> generate (G eval,
gain, iSTFT, PCM_16) -> true metrics (SIIB + HASPI + ESTOI, logistic maps) -> D-step (D fwd+bwd, Adam-D).
Workload at N = 1 = BASELINE.json configs[2] (the largest single-GPU configuration): batch 256 synthetic
4 s @ 16 kHz utterances, SIIB+HASPI+ESTOI multi-metric targets, bf16 MFMA operands.  The same JSON line
carries two companions measured after the timed region: "configs1" (BASELINE configs[1]: batch 32,
SIIB+ESTOI), "nonperiodic" (the headline workload at L = 63 871, no multiple of SIIB's 200-sample hop, so
neither SIIB's frame-periodic shortcut nor its rank-deficient-component cut applies), "headline_f32" (float32 MFMA
operands), "shard128" / "global1024" (the two ends of configs[3]'s strong-scaling ratio measured on one GPU).
Multi-GPU: utterances shard across ranks, one flat RCCL all-reduce of the G and of the D gradients
per step.  `python bench.py --gpus N` with N > 1 and no RANK in the environment starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
(never an exec) before anything touches the GPU, relays its JSON line and exits with its code; under torchrun the
script is a rank.  Default at N > 1 = BASELINE configs[3]: global batch 1024 split over the ranks (strong scaling,
north_star's ">= 6x at 8 GPUs"); `--batch B` with N > 1 = B utterances per GPU (weak scaling); `--global-batch G`
any other total.  `--dry-run` runs the launcher, the sharding and the collectives on CPU (gloo) without kernels.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F64_PEAK_TFLOPS = 78.6            # MI355X_MICROARCH.md: float64, vector FMA and v_mfma_f64_16x16x4 alike
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 at the f32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E, about 8 TB/s
PROFILE_ROUND = 'r06'            # profiles/<round>/traffic.json: PMC passes of this command (tools/prof_round.sh + tools/make_traffic_json.py)


def csrc_sha():
    """sha256 over the kernel sources: PMC counters of another tree's kernels must not be printed next to this tree's timings."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'nele_gan_amd', 'csrc')
    for n in sorted(os.listdir(d)):
        if n.endswith(('.hip', '.h')):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), 'rb').read())
    return h.hexdigest()[:16]


def launch_ranks(n, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as a fresh child process tree and relay its output.  Nothing in this process
    has touched the GPU (torch is not even imported), and the child is a subprocess, not an exec."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if line is not None:
        print(line)
    return r.returncode if (r.returncode != 0 or line is not None) else 1


def dry_run(a, world, rank):
    """Launcher + sharding + collectives on CPU (gloo), no kernels: what `--gpus N` does around the step, checkable without GPUs."""
    import torch
    import torch.distributed as dist
    from nele_gan_amd import dist as ndist
    if world > 1:
        dist.init_process_group('gloo')
    lo, hi = ndist.shard_range(a.global_batch, rank, world)
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
    g = torch.full((2093120,), float(rank + 1))            # G's gradient bucket (8.37 MB)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ndist.allreduce_mean_(g)
        time.sleep(0.001 * (hi - lo) / 64.0)
    dt = time.perf_counter() - t0
    shard = torch.tensor([float(hi - lo)])
    # BASELINE configs[4]: 10 000 utterances of 8 s, contiguous file shards (inference.enhance_files: dist.shard_range), batches of 128
    ilo, ihi = ndist.shard_range(10000, rank, world)
    ish = torch.tensor([float(ihi - ilo), float((ihi - ilo + 127) // 128)])
    if world > 1:
        dist.all_reduce(shard)
        dist.all_reduce(ish)
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'utterances/sec per GAN_epoch step (G+D+metric loss)', 'value': a.global_batch * a.steps / dt, 'unit': 'utterances/s',
                          'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
                          'scaling': a.scaling, 'vs_baseline': None, 'dtype': a.precision, 'data': 'synthetic', 'dry_run': True,
                          'config': {'workload': 'DRY RUN (CPU, gloo, no kernels): launcher + sharding + collectives only',
                                     'global_batch': a.global_batch, 'parallelism': 'dp%d' % world},
                          'ranks_seen': int(ones.item()), 'shard_sum': int(shard.item()), 'shard_rank0': [lo, hi],
                          'configs4': {'files': 10000, 'files_sharded': int(ish[0].item()), 'shard_rank0': [ilo, ihi], 'batches_of_128_all_ranks': int(ish[1].item())}}))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(metrics, length, n_utt, seed_start=9000):
    """The oracle's canonical step (CPU port of the reference path) on a bounded sample, parallelised the way the reference is
    (SURVEY 8d): features by 8 loader workers (dataloader.py:91), metrics fanned out over min(32, cores) joblib processes
    (audio_util.py:146), G / D at batch 1 on torch-CPU with a bounded intra-op thread count."""
    import joblib
    import torch
    from nele_gan_amd import synth
    from nele_gan_amd.model import Discriminator, Generator_Conv1D_cLN
    from oracle.step import CpuStep, _noop
    ncpu = os.cpu_count() or 1
    n_metric = min(32, ncpu)
    n_feat = min(8, ncpu)
    n_torch = min(16, ncpu)
    torch.set_num_threads(n_torch)
    torch.manual_seed(666)
    G, D = Generator_Conv1D_cLN(), Discriminator(nout=len(metrics))
    step = CpuStep(G.state_dict(), D.state_dict(), metrics=metrics)
    c, v = synth.batch(n_utt, length, start=seed_start)
    joblib.Parallel(n_jobs=n_metric)(joblib.delayed(_noop)(i) for i in range(4 * n_metric))     # worker start-up is not part of the sample
    t0 = time.perf_counter()
    step.epoch_slice(c, v, feature_workers=n_feat, metric_workers=n_metric)
    dt = time.perf_counter() - t0
    return {'value': n_utt / dt, 'unit': 'utterances/s', 'cores': max(n_metric, n_torch), 'kind': 'port',
            'sample': '%d synthetic %.1f s utterances, %s: features on %d worker processes, G-step and D-step at batch 1 (reference semantics) with %d torch '
                      'threads, metrics on %d joblib processes; %.1f s wall; host %d logical cores; stage seconds %s'
                      % (n_utt, length / 16000.0, '+'.join(metrics), n_feat, n_torch, n_metric, dt, ncpu, {k: round(t, 2) for k, t in step.times.items()})}


def inference_rate(tr, batch, K, rank, sweep=(64, 256, 512)):
    """inference.py:79-117 on batches of 8 s utterances (BASELINE configs[4] per GPU: pure replicas, no collective).
    `value` = Enhancer.enhance_stream with 3 batches in flight (each batch on its own stream: the IMCRA scan and the per-utterance tails of
    one batch run under the generator of another; results bit-identical to the single-stream path, tests/test_epoch_gpu.py); the
    single-stream rate, the IMCRA scan alone and other batch sizes are reported beside it.  Two roofline fractions for the whole path:
    SURVEY 8d's algorithmic work per utterance at T = 501 - 2.093 GFLOP in the generator (bf16 MFMA) and 3.5 MB of HBM traffic."""
    import torch
    from nele_gan_amd import _lib, synth
    from nele_gan_amd import audio_util as au
    from nele_gan_amd.inference import Enhancer, p_power
    enh = Enhancer(G=tr.G)
    enh.G.precision = tr.G.precision

    def rate(B, steps, inflight):
        c, v = synth.batch(min(B, 128), 128000, start=5000 + rank * 128)
        if B > c.shape[0]:
            import numpy as np
            c, v = np.tile(c, (B // c.shape[0] + 1, 1))[:B], np.tile(v, (B // v.shape[0] + 1, 1))[:B]
        cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
        if inflight == 0:
            enh.enhance(cw, nw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = enh.enhance(cw, nw)
        else:
            for out in enh.enhance_stream([(cw, nw)] * (inflight + 1), inflight=inflight):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for out in enh.enhance_stream([(cw, nw)] * steps, inflight=inflight):
                pass
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, cw, nw

    dt, cw, nw = rate(batch, 4 * K, 4)               # four batches in flight: three side streams + the caller's own (four hardware queues)
    dt3, _, _ = rate(batch, 4 * K, 3)
    dt1, _, _ = rate(batch, K, 0)
    # the noise branch alone (STFT -> |Y|^2, then IMCRA's recursions - serial over the 501 frames - as one thread per utterance and bin)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    au.noise_band(nw, p_power)
    ev[0].record()
    for _ in range(5):
        au.noise_band(nw, p_power)
    ev[1].record()
    torch.cuda.synchronize()
    imcra_ms = ev[0].elapsed_time(ev[1]) / 5
    out = {'value': batch / dt, 'unit': 'utterances/s', 'ms_per_batch': dt * 1e3, 'utterance_seconds': 8.0, 'batch': batch, 'batches_in_flight': 4,
           'three_in_flight': {'value': batch / dt3, 'ms_per_batch': dt3 * 1e3}, 'realtime_factor': batch * 8.0 / dt, 'single_stream': {'value': batch / dt1, 'ms_per_batch': dt1 * 1e3}, 'noise_branch_ms': imcra_ms,
           'roofline': {'mfma': {'achieved': 2.093e9 * batch / dt / 1e12, 'peak': BF16_MFMA_PEAK_TFLOPS if tr.G.precision == 'bf16' else F32_MFMA_PEAK_TFLOPS,
                                 'unit': 'TFLOP/s', 'flops_per_utterance': 2.093e9},
                        'hbm': {'achieved': 3.5e6 * batch / dt / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'bytes_per_utterance': 3.5e6}},
           'by_batch': {}}
    for k in ('mfma', 'hbm'):
        out['roofline'][k]['frac'] = out['roofline'][k]['achieved'] / out['roofline'][k]['peak']
    for B in sweep:
        try:
            d, _, _ = rate(B, max(4, 2 * K * 128 // B), 4)
            out['by_batch'][str(B)] = {'value': B / d, 'ms_per_batch': d * 1e3}
        except Exception as e:                               # a sweep point must not take the line down
            out['by_batch'][str(B)] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        torch.cuda.empty_cache()
    tr.G.train()
    return out


def companion(a, metric_str, batch, length, steps, main_tr=None, precision=None, pipe=None, serial=False):
    """The canonical step of another workload on this GPU (fresh trainer, 2 warm-up steps, `steps` timed steps)."""
    import gc
    import torch
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    try:
        return _companion(a, metric_str, batch, length, steps, main_tr, precision, pipe, serial)
    except Exception as e:                                  # a companion must not take the headline line down with it
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300]), 'batch': batch, 'samples_per_utterance': length, 'metrics': metric_str}
    finally:
        gc.collect()
        torch.cuda.empty_cache()                            # multi-GB metric workspaces of the companion's trainer


def _companion(a, metric_str, batch, length, steps, main_tr=None, precision=None, pipe=None, serial=False):
    import torch
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    if serial:
        # every "side stream" of this trainer is the current stream (ops.side_stream): the whole step on ONE stream, kernels strictly in
        # sequence - what the multi-stream schedule is worth, as a measured number
        os.environ['NELE_SERIAL'] = '1'
        main_tr = None
    try:
        return _companion_run(a, metric_str, batch, length, steps, main_tr, precision, pipe)
    finally:
        if serial:
            os.environ.pop('NELE_SERIAL', None)


def _companion_run(a, metric_str, batch, length, steps, main_tr, precision, pipe):
    import torch
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    tr = GanTrainer(metric_str)
    tr.D.precision = tr.G.precision = precision or a.precision
    if main_tr is not None:
        # same logical streams as the headline trainer: the runtime multiplexes streams onto 4 hardware queues, and a second set of
        # seven streams would time-slice with the (idle) first set's queues ("hardware queues")
        tr._side, tr._side2, tr._fside = main_tr._side, main_tr._side2, main_tr._fside
        tr.D._wstream, tr.G._wstream = main_tr.D._wstream, main_tr.G._wstream
    c, v = synth.batch(batch, length, start=20000)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    # A batch of at most GanTrainer.early_prefetch_max_batch utterances leaves most of the GPU idle and its step is as long as its longest
    # dependent chain (SIIB's clean-signal half with the eigen-decomposition: 4.2 of 5.8 ms at B = 32).  A loop that knows its next batch
    # - every loop fed by a DataLoader does - runs that chain a step ahead: canonical_step(next_batch=...) enqueues the next batch's
    # input-only work at the start of the step on a second set of streams / workspaces.  Same kernels, same work per step (every timed
    # step carries one batch's input-only work), bit-identical results; the plain figure (every step strictly on its own, as the
    # headline step runs) is reported beside it.  Larger batches saturate the GPU by themselves: pipelining costs them 2 - 4 ms.
    def timed(pipelined):
        pre = None
        def one_step():
            nonlocal pre
            if pipelined:
                r = tr.canonical_step(cw, nw, pre=pre, next_batch=(cw, nw))
                pre = tr.prefetched
                return r
            return tr.canonical_step(cw, nw)
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lg, ld, tgt = one_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert bool(torch.isfinite(tgt).all()) and bool(torch.isfinite(ld))
        tr.check_status()
        return dt
    if pipe is None:
        pipe = batch <= tr.early_prefetch_max_batch or os.environ.get('NELE_PREFETCH', '0') == '1'
    dt_plain = timed(False)
    dt = timed(True) if pipe else dt_plain
    out = {'value': batch / dt, 'unit': 'utterances/s', 'ms_per_step': dt * 1e3, 'batch': batch, 'samples_per_utterance': length,
           'metrics': metric_str, 'steps': steps, 'dtype': precision or a.precision}
    if pipe:
        out['pipeline'] = ("next batch's input-only work (features, clean-signal halves of the metrics) enqueued at the start of the step (canonical_step(next_batch=...))"
                           if batch <= tr.early_prefetch_max_batch else
                           "next batch's features enqueued behind this step's targets (canonical_step(next_batch=...), late_prefetch = 'features')")
        out['ms_per_step_plain'] = dt_plain * 1e3
    return out


def epoch_equivalent(tr, cw, nw, K, utts):
    """The reference's per-utterance work mix of one GAN epoch (train_nele.py:110-429): one G-step, one generated sample, true targets of
    the generated and of the pre-enhanced 'DRC' example, and 2 x 3 D-steps (both examples in each of the three passes; the 1/30 history
    replay of pass B is left out).  The stages follow each other on the current stream - no cross-stage overlap, so this is a lower bound;
    inside the target stage the three metrics run on their own streams (GanTrainer.true_metrics_pair, as run_epoch calls it)."""
    import torch
    drc = (cw * 1.5).contiguous()                       # stands for the pre-enhanced example of the same utterances
    def unit():
        f = tr.features(cw, nw)
        tr.g_step(f['clean_band'], f['noise_band'])
        enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
        L = enh.shape[1]
        # both examples of the same clean batch: the clean-signal halves of SIIB (KLT basis) and HASPI (reference chain) run once
        t_gen, t_drc = tr.true_metrics_pair(cw, enh, drc[:, :L].contiguous(), nw)
        d_gen = tr.d_inputs(enh, f['noise_band'], f['clean_band'])
        d_drc = tr.d_inputs(drc[:, :L].contiguous(), f['noise_band'], f['clean_band'])
        for _ in range(3):
            tr.d_step(d_gen, t_gen)
            tr.d_step(d_drc, t_drc)
    unit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        unit()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    return {'value': utts / dt, 'unit': 'utterances/s', 'ms_per_unit': dt * 1e3,
            'mix': '1 G-step + generate + targets of 2 examples (clean-signal work shared; SIIB / HASPI / ESTOI on three streams) + 6 D-steps per batch, stages in sequence'}


def epoch_with_quality(a, cw, nw):
    """GanTrainer.run_epoch itself - the reference's whole epoch body (train_nele.py:110-429) on the headline batch with the quality
    discriminator ON (train_nele.py:150-152, 362-365: the reference always trains D_Qua; PESQ / ViSQOL are not built, so the targets are
    synthetic constants), a pre-enhanced ('DRC') example per utterance and the 1/30 history replay of pass B: G-step with the
    0.5 MSE(D_Qua) term, checkpoint-free, sample generation, targets of both examples (clean-signal work shared), three D passes over
    2 x B items + replay, D and D_Qua stepped.  Timed over two epochs after a warm-up epoch that fills the history."""
    import gc
    import torch
    from nele_gan_amd.train_nele import GanTrainer
    try:
        tr = GanTrainer(a.metrics, use_quality=True)
        tr.D.precision = tr.G.precision = tr.D_Qua.precision = a.precision
        B = cw.shape[0]
        drc = (cw * 1.5).contiguous()
        qua = torch.full((B, 2), 0.6, device=cw.device)
        batches = [{'clean': cw, 'noise': nw, 'drc': drc, 'qua': qua, 'drc_qua': qua * 0.5}]
        tr.run_epoch(2, batches, (), d_batch=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = None
        for ep in (3, 4):
            res = tr.run_epoch(ep, batches, (), d_batch=B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        return {'value': B / dt, 'unit': 'utterances/s', 'ms_per_epoch': dt * 1e3, 'utterances': B, 'd_steps_per_epoch': res['d_steps'], 'g_steps_per_epoch': res['g_steps'],
                'history_items': len(tr.history),
                'mix': 'GanTrainer.run_epoch: G-step incl. 0.5 MSE(D_Qua), generate, targets of generated + pre-enhanced example, 3 D passes over 2 B items + 1/30 replay, '
                       'D and D_Qua stepped; quality targets synthetic (PESQ / ViSQOL not built)'}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        gc.collect()
        torch.cuda.empty_cache()


def epoch_from_files(a, n_utt=256, batch=64):
    """End to end from wav files: a synthetic corpus (PCM_16, lengths 3 .. 4 s, as the reference's folders of clean / noise files) is
    written to tmpfs, then GanTrainer.run_epoch is fed from it through dataio.FileBatches (a batch of files per library call into pinned
    int16 rows, asynchronous copies, int16 -> float32 on the device, two batches ahead) - the reference feeds its loop from 8 DataLoader workers (dataloader.py:86-98) and re-reads the files in
    every stage.  Reported: utterances/s of the epoch fed from files, of the same epoch on batches already resident in HBM, and of the
    loader alone (decode + upload)."""
    import gc
    import shutil
    import tempfile
    import numpy as np
    import torch
    from nele_gan_amd import dataio, synth
    from nele_gan_amd.train_nele import GanTrainer
    root = tempfile.mkdtemp(prefix='nele_corpus_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        c, v = synth.batch(n_utt, 64000, start=40000)
        rs = np.random.RandomState(0)
        os.makedirs(root + '/Clean'); os.makedirs(root + '/Noise')
        files = []
        for i in range(n_utt):
            L = int(rs.randint(48000, 64001))
            dataio.write_wav_pcm16('%s/Clean/u%04d.wav' % (root, i), c[i, :L])
            dataio.write_wav_pcm16('%s/Noise/u%04d.wav' % (root, i), v[i, :L])
            files.append('%s/Clean/u%04d.wav' % (root, i))
        tr = GanTrainer(a.metrics)
        tr.D.precision = tr.G.precision = a.precision
        nthreads = min(8, os.cpu_count() or 8)        # library threads per batch call (measured: 8 - 16 flat, 32 slower: the calls only move bytes)
        fb = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=nthreads, ahead=2, keep=2)   # keep < batches: every pass over the list decodes again
        for ep in (2, 3, 4):                                                  # warm-up: buffer sets / plans of the padded D shapes (shuffled chunks + replay)
            tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
        torch.cuda.synchronize()
        n0 = fb.decoded_files
        t0 = time.perf_counter()
        for ep in (5, 6):                                                     # the generated samples are written as name@epoch.wav like train_nele.py:309-313
            res = tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
        torch.cuda.synchronize()
        dt_files = (time.perf_counter() - t0) / 2
        decoded = (fb.decoded_files - n0) // 2
        fbm = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=nthreads, ahead=0, keep=len(fb) + 1)
        mem = [dict(fbm[i]) for i in range(len(fbm))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in (7, 8):
            tr.run_epoch(ep, mem, (), d_batch=batch)
        torch.cuda.synchronize()
        dt_mem = (time.perf_counter() - t0) / 2
        fb2 = dataio.FileBatches(files, root + '/Noise/', batch=batch, workers=nthreads, ahead=2, keep=1)
        for _ in range(2):                                                  # two passes: the pinned staging buffers are allocated (tens of milliseconds
            for b in fb2:                                                   # each; the wrap-around from the last batch to the first needs one set more)
                pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in fb2:                                                       # steady state: every file decoded again (keep = 1), buffers from the pool
            pass
        torch.cuda.synchronize()
        dt_load = time.perf_counter() - t0
        # the same epoch with the clean-signal halves of SIIB / HASPI kept across epochs (GanTrainer.enable_clean_cache: the reference draws the
        # same training files every epoch, train_nele.py:35-38,119): epoch 9 fills the cache, epochs 10 and 11 are timed
        cache = tr.enable_clean_cache()
        tr.run_epoch(9, fb, (), d_batch=batch, sample_dir=root + '/out')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in (10, 11):
            tr.run_epoch(ep, fb, (), d_batch=batch, sample_dir=root + '/out')
        torch.cuda.synchronize()
        dt_cached = (time.perf_counter() - t0) / 2
        cached = {'value': n_utt / dt_cached, 'unit': 'utterances/s', 'ms_per_epoch': dt_cached * 1e3, 'speedup_vs_uncached': dt_files / dt_cached,
                  'cache': cache.stats(), 'note': 'epochs >= 2 of a corpus whose clean files do not change; bit-identical targets (tests/test_clean_cache_gpu.py)'}
        tr.clean_cache = None
        fb.close(); fb2.close(); fbm.close()
        return {'value': n_utt / dt_files, 'cached': cached, 'unit': 'utterances/s', 'ms_per_epoch': dt_files * 1e3, 'utterances': n_utt, 'batch': batch,
                'files_decoded_per_epoch': decoded, 'files_written_per_epoch': len(res.get('sample_files', ())), 'd_steps': res['d_steps'], 'g_steps': res['g_steps'],
                'resident_batches': {'value': n_utt / dt_mem, 'ms_per_epoch': dt_mem * 1e3},
                'loader_alone': {'value': n_utt / dt_load, 'unit': 'utterances/s (clean + noise wav decoded, padded, uploaded)', 'host_threads': nthreads,
                                 'wav_MB_per_s': 2 * sum(os.path.getsize(f) for f in files) / dt_load / 1e6},
                'mix': 'GanTrainer.run_epoch(epoch >= 2) over %d files of 3 .. 4 s in batches of %d: G-steps, generate, targets, 3 D passes + replay' % (n_utt, batch)}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        gc.collect()
        torch.cuda.empty_cache()


def inference_from_files(tr, n_utt=4096, batch=128):
    """BASELINE configs[4] end to end on one GPU: inference.py:79-117 from wav files to wav files - a synthetic corpus of 8 s utterances
    (clean + noise, PCM_16) on tmpfs, `inference.enhance_files` (a batch of files per library call into pinned int16 rows, int16 over
    PCIe both ways, conversions on the device, three batches in flight on the GPU) - the second pass over the whole list is timed (the
    first allocates the pinned staging buffers: tens of milliseconds each)."""
    import gc
    import shutil
    import tempfile
    import numpy as np
    import torch
    from nele_gan_amd import dataio, synth
    from nele_gan_amd.inference import Enhancer, enhance_files
    root = tempfile.mkdtemp(prefix='nele_infer_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        c, v = synth.batch(64, 128000, start=70000)
        os.makedirs(root + '/Clean'); os.makedirs(root + '/Noise')
        rs = np.random.RandomState(1)
        files = []
        for i in range(n_utt):
            k, L = i % 64, int(rs.randint(112000, 128001))
            dataio.write_wav_pcm16('%s/Clean/u%05d.wav' % (root, i), c[k, :L])
            dataio.write_wav_pcm16('%s/Noise/u%05d.wav' % (root, i), v[(k + i // 64) % 64, :L])
            files.append('%s/Clean/u%05d.wav' % (root, i))
        enh = Enhancer(G=tr.G)
        enh.G.precision = tr.G.precision
        nthreads = min(8, os.cpu_count() or 8)        # library threads per batch call (measured: 8 - 16 flat, 32 slower: the calls only move bytes)
        enhance_files(enh, files, root + '/Noise/', root + '/Warm', batch=batch, workers=nthreads)    # first pass: buffer sets, plans, pinned staging buffers
        shutil.rmtree(root + '/Warm', ignore_errors=True)
        passes = []
        for k in range(2):                                                  # two timed passes, the better one is reported (a pass is ~0.1 s: one page-locked allocation shows)
            shutil.rmtree(root + '/Enh', ignore_errors=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = enhance_files(enh, files, root + '/Noise/', root + '/Enh', batch=batch, workers=nthreads)
            torch.cuda.synchronize()
            passes.append(time.perf_counter() - t0)
        dt = min(passes)
        tr.G.train()
        nbytes = sum(os.path.getsize(f) for f in files) * 2 + sum(os.path.getsize(f) for f in out)
        return {'value': n_utt / dt, 'unit': 'utterances/s', 'files': n_utt, 'utterance_seconds': '7 .. 8', 'batch': batch, 'host_threads': nthreads,
                'wav_MB_per_s': nbytes / dt / 1e6, 'seconds': dt, 'passes_seconds': passes,
                'path': 'wav files on tmpfs -> int16 rows of a pinned buffer (one library call per batch) -> HBM -> float32 on the device -> 3 batches in flight -> int16 on the device -> pinned -> PCM_16 files (one library call per batch)'}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        gc.collect()
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=0, help='utterances per GPU (N = 1 default: 256 = BASELINE configs[2]; with N > 1: weak scaling)')
    ap.add_argument('--global-batch', type=int, default=0, metavar='G',
                    help='total utterances, split over the ranks (strong scaling; N > 1 default: 1024 = BASELINE configs[3]; overrides --batch)')
    ap.add_argument('--dry-run', action='store_true', help='CPU / gloo: launcher, sharding and collectives only, no kernels')
    ap.add_argument('--no-isolated', action='store_true', help='skip the isolated D.conv5 forward launches after the timed region (profiling runs: '
                                                                 'the kernel statistics then hold the step\'s own launches only)')
    ap.add_argument('--length', type=int, default=64000)
    ap.add_argument('--metrics', default='siib&haspi&estoi')
    ap.add_argument('--companions', type=int, default=1, help='also run the configs[1] and non-periodic-length companions (N = 1 only; 0 = skip)')
    ap.add_argument('--cpu-utts', type=int, default=32, help='utterances in the CPU-baseline sample (0 = skip)')
    ap.add_argument('--breakdown', action='store_true', help='per-stage timing to stderr')
    ap.add_argument('--inference', type=int, default=-1, metavar='K',
                    help='also time K batches of the inference path (BASELINE configs[4]: 8 s utterances, features -> G -> resynthesis -> RMS 0.03 -> PCM_16); reported as "inference"')
    ap.add_argument('--epoch-equivalent', type=int, default=-1, metavar='K',
                    help='also time K units of the reference epoch mix per batch (SURVEY 8d): 1 G-step, generate, targets of the generated and of the\n'
                         'pre-enhanced (DRC) example, 6 D-steps (2 examples x 3 passes, train_nele.py:342-426); reported as "epoch_equivalent"')
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16'],
                    help='MFMA operand type of the discriminator conv forward / data-gradient passes (f32 accumulate; BASELINE configs[1] names bf16)')
    a = ap.parse_args()
    if a.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))              # before torch is imported: this process never touches a GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # workload defaults: N = 1 -> configs[2] (256 utterances on the GPU); N > 1 -> configs[3] (1024 utterances over the ranks)
    a.scaling = 'weak'
    if a.global_batch == 0 and a.batch == 0:
        if world > 1:
            a.global_batch = 1024
        else:
            a.batch = 256
    if a.global_batch > 0:
        a.scaling = 'strong'
    companions_default = world == 1 and a.batch == 256 and a.length == 64000 and a.metrics == 'siib&haspi&estoi'
    if a.inference < 0:
        a.inference = 10 if (companions_default and a.companions) else 0
    if a.epoch_equivalent < 0:
        a.epoch_equivalent = 2 if (companions_default and a.companions) else 0
    if a.dry_run:
        if a.global_batch == 0:
            a.global_batch = a.batch * world
        return dry_run(a, world, rank)

    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # NELE_BENCH_ONE_DEVICE=1 (test harness for 1-GPU boxes, never a measurement): every rank on device 0, buckets over gloo - the same
    # sharding, deferred D update, barriers and max-over-ranks timing with the real kernels; the line carries "one_device_test": true
    one_device = world > 1 and os.environ.get('NELE_BENCH_ONE_DEVICE', '0') == '1'
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    # one process per GPU, bound to the CPUs of that GPU's NUMA node before anything is allocated (pinned staging buffers, loader threads):
    # the file-fed companions are host-bound, and an unbound process on a two-socket box copies across the socket link
    cpus_before = os.sched_getaffinity(0)
    from nele_gan_amd import dist as _nd
    numa_cpus = _nd.bind_to_gpu_numa_node(local)
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from nele_gan_amd import dist as ndist
    from nele_gan_amd import ops, synth
    from nele_gan_amd.train_nele import GanTrainer, parse_metrics
    metrics = parse_metrics(a.metrics)
    tr = GanTrainer(a.metrics, device='cuda:%d' % local)
    tr.D.precision = a.precision
    tr.G.precision = a.precision
    scaling = a.scaling
    if a.global_batch > 0:
        assert a.global_batch % world == 0, "--global-batch must be a multiple of the number of GPUs (equal shards: one plain mean all-reduce)"
        lo, hi = ndist.shard_range(a.global_batch, rank, world)      # contiguous utterance shard of this rank
        a.batch = hi - lo
    else:
        lo = rank * a.batch
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, device='cuda')
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    c, v = synth.batch(a.batch, a.length, start=lo)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # NELE_PREFETCH=1 (A/B switch, default off): the input-only work of the NEXT batch (features, SIIB / HASPI clean-signal halves) is enqueued
    # behind the current step's targets, like the reference's DataLoader workers prepare upcoming items (dataloader.py:86-92).  Measured at
    # B = 256: 78.3 against 77.3 ms per step - the GPU is saturated by the step's own kernels, moving work into the D backward pass only
    # slows that critical chain down - so every step runs strictly on its own.
    use_pre = os.environ.get('NELE_PREFETCH', '0') == '1' and not a.breakdown
    pre = None
    def one_step():
        nonlocal pre
        if use_pre:
            r = tr.canonical_step(cw, nw, pre=pre, next_batch=(cw, nw))
            pre = tr.prefetched
            return r
        return tr.canonical_step(cw, nw)
    for _ in range(a.warmup):
        one_step()
    tag = 'D.conv5.fwd'           # D's 5th conv forward: every launch of the timed region (one in the G-step, one in the D-step per step)
    wtag = 'D.conv5.wgrad'        # the memory-side companion figure: D's 5th conv weight gradient (events on its own stream; includes its partial reduction)
    from nele_gan_amd import _lib
    # bf16 mode: the two dense figures come from the library's own HIP-event hook (shape-specific tags of nele_conv16 / nele_conv_wgrad_*), so
    # that the timed steps run exactly as a training loop runs them - G and D passes replayed from their recorded job tables (nele_gen_fwd /
    # nele_disc_bwd ..., one foreign call per pass).  float32 mode (other kernels, no tags): torch events around the per-layer calls.
    T_ = 1 + a.length // 256
    lib_tags = a.precision == 'bf16'
    ctag, wltag = 'conv16_N64_K3888_e2', 'wgrad_N64_K3888'
    ops.PROFILE = None if lib_tags else {'gstep.' + tag: [], tag: [], wtag: []}
    htag = 'haspi_bank_gain_kernel'    # the HBM-side figure: HASPI's signal filter bank + compression-gain pass, timed by the library's HIP-event hook
    etag = 'eigh_tridiag_cluster'      # the float64 figure: the cluster tridiagonalisation of SIIB's 420 x 420 covariance (largest kernel of the r03 step)
    armed = ([htag] if 'haspi' in metrics else []) + ([etag] if 'siib' in metrics else []) + ([ctag, wltag] if lib_tags else [])
    if armed:
        _lib.profile_begin(','.join(armed))
    stage_ev = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if a.breakdown:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            ev[0].record(); f = tr.features(cw, nw)
            ev[1].record(); tr.g_step(f['clean_band'], f['noise_band'])
            ev[2].record(); enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
            ev[3].record(); tgt = tr.true_metrics(cw, enh, nw)
            ev[4].record(); tr.d_step(tr.d_inputs(enh, f['noise_band'], f['clean_band']), tgt)
            ev[5].record(); stage_ev.append(ev)
        else:
            last = one_step()
    tr._flush_d()                       # (several ranks: the last step's D update is deferred under the NEXT step's features; there is none)
    barrier()
    dt = time.perf_counter() - t0
    if not a.breakdown:
        # the timed steps must have produced numbers: finite losses / targets / weights and no optimiser step masked on the device
        lg_, ld_, tgt_ = last
        assert bool(torch.isfinite(lg_)) and bool(torch.isfinite(ld_)) and bool(torch.isfinite(tgt_).all()), "non-finite loss / targets in the timed region"
        assert bool(torch.isfinite(tr.G.flat_parameters().flat).all()) and bool(torch.isfinite(tr.D.flat_parameters().flat).all()), "non-finite weights"
        status = tr.check_status(raise_on_error=True)
    rank_ms = None
    if world > 1:
        # per-rank clocks of the same timed region (the line's ms_per_step is their maximum): a slow rank, a rank that waited in the
        # collectives and the eigensolver's repair count of every rank are visible in the first real multi-GPU line
        mine = dt / a.steps * 1e3
        lo_ = torch.tensor([mine], device='cuda', dtype=torch.float64)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        rep_ = torch.tensor([float(status.get('eigh_repaired', 0)) if not a.breakdown else 0.0], device='cuda', dtype=torch.float64)
        dist.all_reduce(rep_, op=dist.ReduceOp.MAX)
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rank_ms = {'min': float(lo_.item()), 'max': dt / a.steps * 1e3, 'rank0': mine, 'eigh_repaired_max_over_ranks': int(rep_.item())}
    hbm_ms = _lib.profile_collect_tag(htag) if 'haspi' in metrics else []
    eig_ms = _lib.profile_collect_tag(etag) if 'siib' in metrics else []

    class _Ms:                                               # a measured duration with torch.cuda.Event's interface
        def __init__(self, ms):
            self.ms = ms

        def elapsed_time(self, other):
            return other.ms

    def as_events(ms_list, work):
        return [(_Ms(0.0), _Ms(m), work) for m in ms_list]

    if lib_tags:
        conv_flops = 2.0 * a.batch * 44 * (T_ - 20) * 64 * 3888
        cms = _lib.profile_collect_tag(ctag)                 # launches alternate: G-step's D forward, D-step's D forward
        prof_g, prof_d = as_events(cms[0::2], conv_flops), as_events(cms[1::2], conv_flops)
        wbytes = 4.0 * (a.batch * 52 * (T_ - 12) * 48 + a.batch * 44 * (T_ - 20) * 64 + 64 * 3888)
        prof_w = as_events(_lib.profile_collect_tag(wltag), wbytes)
    else:
        prof_g, prof_d = ops.PROFILE['gstep.' + tag], ops.PROFILE[tag]
        prof_w = ops.PROFILE[wtag]
    _lib.profile_begin(None)
    prof = prof_g + prof_d
    # the same launch on an otherwise idle GPU (after the timed region): inside the step the kernel shares the CUs with the metric
    # stream (SIIB's clean-signal part, incl. the all-CU tridiagonalisation, runs beside the G-step), which inflates its duration
    iso_tag = 'iso.D.conv5.fwd'
    ops.PROFILE = None if lib_tags else {iso_tag: []}
    torch.cuda.synchronize()
    iso = []
    if not a.no_isolated:            # profiling runs switch this off: the per-step kernel sums must hold the step's own launches only
        tr.D.eval()
        tr.D.profile_prefix = 'iso.'
        if lib_tags:
            _lib.profile_begin(ctag)
        with torch.no_grad():
            for _ in range(6):
                tr.D.forward_packed(tr._last_din)
        torch.cuda.synchronize()
        iso = as_events(_lib.profile_collect_tag(ctag), conv_flops)[1:] if lib_tags else ops.PROFILE[iso_tag][1:]
        _lib.profile_begin(None)
        tr.D.profile_prefix = ''
        tr.D.train()
    ops.PROFILE = None
    # the step's two gradient all-reduces (G 8.37 MB, D 1.37 MB) alone, after the timed region: what a step pays for them at most
    allreduce_ms = None
    if world > 1:
        gg, gd = tr.G.flat_parameters().grad, tr.D.flat_parameters().grad
        for _ in range(3):
            ndist.allreduce_mean_(gg); ndist.allreduce_mean_(gd)
        barrier()
        ta = time.perf_counter()
        for _ in range(10):
            ndist.allreduce_mean_(gg); ndist.allreduce_mean_(gd)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) / 10 * 1e3
    if rank == 0:
        T = 1 + a.length // 256
        # HBM bytes per launch (PMC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, gfx950-corrected) and MFMA-busy figures of the
        # SAME command on this tree: profiles/<PROFILE_ROUND>/traffic.json (tools/prof_round.sh + tools/make_traffic_json.py); counters
        # cannot be read from inside the process, so the line carries them only for the workload the committed passes were taken on
        pmc = {}
        pmc_note = 'no committed PMC passes for this workload'
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND, 'traffic.json')))
            if tj.get('workload') == {'batch': a.batch, 'length': a.length, 'metrics': a.metrics, 'precision': a.precision}:
                if tj.get('csrc_sha') == csrc_sha():
                    pmc = tj['kernels']
                    pmc_note = 'profiles/%s/traffic.json (csrc %s)' % (PROFILE_ROUND, tj.get('csrc_sha'))
                else:
                    pmc_note = 'profiles/%s/traffic.json was taken on other kernel sources (csrc %s, this tree %s): counters dropped' % (
                        PROFILE_ROUND, tj.get('csrc_sha'), csrc_sha())
        except Exception:
            pmc = {}
        ckey = 'conv16_kernel<4, 4, true>'                     # D.conv5 forward: 48 -> 64 channels (TN = 4), 4-row tiles, bf16 output + pooled partial sums (nele_conv16_gap)
        traffic = pmc.get(ckey, {}).get('hbm_bytes_corrected')
        kernel_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof) / max(1, len(prof))
        flops = prof[0][2] if prof else 0.0
        iso_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in iso) / max(1, len(iso))
        peak = BF16_MFMA_PEAK_TFLOPS if a.precision == 'bf16' else F32_MFMA_PEAK_TFLOPS
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        out = {
            'metric': 'utterances/sec per GAN_epoch step (G+D+metric loss)',
            'value': a.batch * world * a.steps / dt,
            'unit': 'utterances/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': a.precision, 'data': 'synthetic',
            'config': {'workload': (('BASELINE configs[3]' if world > 1 else 'BASELINE configs[2]') if 'haspi' in a.metrics.lower() else 'BASELINE configs[1]') + ': batch=%d/GPU synthetic %.0f s@16 kHz RMS 0.03 utterances, %s targets, '
                                   'canonical GAN_epoch step (features, G-step, generate, metrics, D-step)' % (a.batch, a.length / 16000.0, '+'.join(metrics).upper()),
                       'global_batch': a.batch * world, 'samples_per_utterance': a.length, 'frames': T, 'parallelism': 'dp%d' % world},
            'roofline': {'bound': 'mfma',
                         'kernel': '%s (%s: implicit-GEMM Conv2d 48->64 9x9, %s MFMA operands, f32 accumulate, M=%d N=64 K=3888)' % (
                             'conv16_kernel<4,4,true> via nele_conv16_gap' if a.precision == 'bf16' else 'conv_span_kernel<4>', tag, a.precision, a.batch * 44 * (T - 20)),
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic, 'mfma_busy_frac': pmc.get(ckey, {}).get('mfma_busy_frac'), 'launch_ms': kernel_ms,
                         'launch_ms_gstep': sum(e0.elapsed_time(e1) for e0, e1, _ in prof_g) / max(1, len(prof_g)),   # beside the half-GPU tridiagonalisation
                         'launch_ms_dstep': sum(e0.elapsed_time(e1) for e0, e1, _ in prof_d) / max(1, len(prof_d)),
                         'isolated_launch_ms': iso_ms, 'achieved_isolated': (flops / (iso_ms * 1e-3) / 1e12 if iso_ms > 0 else 0.0),
                         'frac_isolated': (flops / (iso_ms * 1e-3) / 1e12 / peak if iso_ms > 0 else 0.0), 'launches_timed': len(prof), 'flops_per_launch': flops,
                         'pmc_source': pmc_note},
            'ranks_seen': ranks_seen,
            'host_cpus_bound': (len(numa_cpus) if numa_cpus else None),     # CPUs of the GPU's NUMA node this process was restricted to (None: unbound)
            **({'per_rank_ms_per_step': rank_ms} if rank_ms is not None else {}),
            **({'one_device_test': True} if one_device else {}),
            'step_status': status if not a.breakdown else None,      # device-side counters of the timed region + warm-up (GanTrainer.check_status): eigh_repaired must be 0
        }
        if allreduce_ms is not None:
            out['allreduce_ms_per_step'] = allreduce_ms
        if prof_w:
            # D.conv5 weight gradient: dW[64][3888] = dY^T im2col(X), an MFMA-bound GEMM (arithmetic intensity ~1000 FLOP/B)
            w_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in prof_w) / len(prof_w)
            w_flops = flops                                            # same M, N, K as the forward pass
            out['roofline_wgrad'] = {'bound': 'mfma', 'kernel': 'conv_wgrad_dma_kernel<4,11,3> + wgrad_reduce_kernel (%s: Conv2d 48->64 9x9 weight gradient)' % wtag,
                                     'achieved': w_flops / (w_ms * 1e-3) / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                                     'frac': w_flops / (w_ms * 1e-3) / 1e12 / peak,
                                     'traffic': (pmc.get('conv_wgrad_dma_kernel<4, 11, 3>') or pmc.get('conv_wgrad_dma_kernel<4, 11, 2>') or pmc.get('conv_wgrad_tile16_kernel<4, 7, true, true>') or {}).get('hbm_bytes_corrected'),
                                     'mfma_busy_frac': (pmc.get('conv_wgrad_dma_kernel<4, 11, 3>') or pmc.get('conv_wgrad_dma_kernel<4, 11, 2>') or {}).get('mfma_busy_frac'),
                                     'algorithmic_bytes': prof_w[0][2], 'launch_ms': w_ms, 'launches_timed': len(prof_w)}
        if hbm_ms:
            # HASPI signal filter bank (pass 2) + compression gain + gain low-pass + dB SL + IHC pass 1, one launch per signal per step - the
            # step's largest single HBM consumer since the gain pass moved into it: reads the middle-ear signal (float64, once per sample)
            # and the control envelope (float32 |u|^2), writes the dB-SL envelope (float32): 8 bytes per (sample, channel) + 8 per sample.
            # It is bound by the float64 issue rate of its recurrences, not by HBM: the fraction says how far from the memory roof it runs.
            n24p = (int(a.length * 1.5) + 31) // 32 * 32
            h_bytes = float(a.batch) * n24p * (32 * 8 + 8)
            h_ms = sum(hbm_ms) / len(hbm_ms)
            out['roofline_hbm'] = {'bound': 'hbm', 'kernel': 'haspi_bank_scan_kernel<true,true,false,true> (pyhaspi2.py:863-915, 982-997, 1080-1088 fused; %d rows x %d samples x 32 channels)' % (a.batch, n24p),
                                   'achieved': h_bytes / (h_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': h_bytes / (h_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   'traffic': pmc.get('haspi_bank_scan_kernel<true, true, false, true>', {}).get('hbm_bytes_corrected'), 'launch_ms': h_ms,
                                   'algorithmic_bytes': h_bytes, 'launches_timed': len(hbm_ms), 'limited_by': 'float64 issue rate (recurrences), not HBM'}
        if eig_ms:
            # Householder tridiagonalisation of the 420 x 420 clean covariance, the steps the two cluster stages run (trailing size 419 .. 256:
            # two workgroups per matrix hold the lower triangle of the trailing matrix in registers + LDS down to 320 rows, the full
            # matrix from there; both launch sites carry the tag): step k costs 4 m^2 flops (symmetric matrix-vector product + rank-2 update, m = n - 1 - k).  Inside a training step
            # SIIB's clean part asks for 32 / 64 matrices per launch (half of the chip).  achieved = flops of one step's launches / their time.
            n_e, m_hand = 420, 256
            f_mat = float(sum(4 * (n_e - 1 - k) ** 2 for k in range(n_e - m_hand)))
            e_step_ms = sum(eig_ms) / a.steps
            out['roofline_f64'] = {'bound': 'mfma', 'kernel': 'eigh_tridiag_clusters_kernel + eigh_tridiag_cluster2_kernel (oracle/siib.py:97 np.linalg.eigh -> Householder steps 0..%d of %d on '
                                                               '%d matrices per step; float64 FMA, latency-bound cross-workgroup chain)' % (n_e - m_hand - 1, n_e - 2, a.batch),
                                   'achieved': f_mat * a.batch / (e_step_ms * 1e-3) / 1e12, 'peak': F64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': f_mat * a.batch / (e_step_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                                   'traffic': pmc.get('eigh_tridiag_clusters_kernel', {}).get('hbm_bytes_corrected'), 'launch_ms': sum(eig_ms) / len(eig_ms),
                                   'launches_timed': len(eig_ms), 'launches_per_step': len(eig_ms) / a.steps, 'ms_per_step': e_step_ms,
                                   'flops_per_step': f_mat * a.batch}
        # `roofline` = the kernel with the largest time per step among the timed ones (the r03 verdict: the line's first figure must not be
        # the step's best kernel); the dense-conv figure stays as `roofline_mfma`
        out['roofline_mfma'] = out['roofline']
        cands = {'roofline_mfma': kernel_ms * len(prof) / max(1, a.steps)}
        if prof_w:
            cands['roofline_wgrad'] = out['roofline_wgrad']['launch_ms'] * len(prof_w) / a.steps
        if hbm_ms:
            cands['roofline_hbm'] = sum(hbm_ms) / a.steps
        if eig_ms:
            cands['roofline_f64'] = sum(eig_ms) / a.steps
        top = max(cands, key=cands.get)
        out['roofline'] = dict(out[top], nominated='largest time per step among the timed kernels',
                               ms_per_step_by_kernel={k: round(v, 3) for k, v in cands.items()}, same_as=top)
        if a.breakdown and stage_ev:
            names = ['features', 'g_step', 'generate', 'metrics', 'd_step']
            br = {n: sum(ev[i].elapsed_time(ev[i + 1]) for ev in stage_ev) / len(stage_ev) for i, n in enumerate(names)}
            sys.stderr.write('stage ms: %s\n' % json.dumps({k: round(x, 3) for k, x in br.items()}))
        if a.inference > 0:
            out['inference'] = inference_rate(tr, min(a.batch, 128), a.inference, rank)
            if world == 1 and a.companions:
                out['inference_from_files'] = inference_from_files(tr)
        if world == 1 and a.companions:
            out['configs1'] = companion(a, 'siib&estoi', 32, 64000, 12, tr)
            # L = 63 871: no multiple of SIIB's 200-sample hop (nor of 100), like any real file - nothing repeats in the replicated signal,
            # so neither the frame-periodic shortcut nor the rank-deficient-component cut (deviation (i)) applies: this is the
            # workload the 1e-4 SIIB claim against the build's oracle (parity with pysiib: unpinned) holds on
            out['nonperiodic'] = companion(a, a.metrics, a.batch, 63871, 4, tr)
            # the parity mode (float32 MFMA operands: what every golden-vector test runs in) on BASELINE configs[1]'s shape and on the headline's
            out['configs1_f32'] = companion(a, 'siib&estoi', 32, 64000, 8, tr, precision='f32')
            out['headline_f32'] = companion(a, a.metrics, a.batch, a.length, 3, tr, precision='f32')
            # the two ends of BASELINE configs[3]'s strong-scaling ratio on ONE GPU: the per-GPU shard (1024 / 8 = 128 utterances) and the
            # whole global batch on one GPU.  predicted ratio = t(1024) / (t(128) + the step's two gradient all-reduces); the all-reduce
            # figure is an ESTIMATE (8.37 MB + 1.37 MB float32 over an 8-rank RCCL ring on xGMI, latency-bound) until an 8-GPU
            # node measures `allreduce_ms_per_step`
            # the headline workload with the next batch's FEATURES prefetched behind the targets (what a DataLoader-fed loop can do at any batch
            # size; `value` itself stays the plain step: every step strictly on its own)
            out['headline_pipelined'] = companion(a, a.metrics, a.batch, a.length, 6, tr, pipe=True)
            # ... and with every kernel of the step on ONE stream (no side streams at all): what the seven-stream schedule buys
            out['headline_one_stream'] = companion(a, a.metrics, a.batch, a.length, 4, None, pipe=False, serial=True)
            out['shard128'] = companion(a, a.metrics, 128, a.length, 6, tr)
            out['global1024'] = companion(a, a.metrics, 1024, a.length, 3, tr)
            if 'ms_per_step' in out['shard128'] and 'ms_per_step' in out['global1024']:
                ar = 0.3
                out['predicted_strong_scaling_8'] = {'value': out['global1024']['ms_per_step'] / (out['shard128']['ms_per_step'] + ar),
                                                     't_1024_on_1_gpu_ms': out['global1024']['ms_per_step'], 't_128_shard_ms': out['shard128']['ms_per_step'],
                                                     'allreduce_ms_assumed': ar, 'note': 'predicted from two 1-GPU measurements; not a multi-GPU run'}
            out['siib_parity_note'] = ('at L % 200 == 0 (the headline length 64 000) the replicated signal is exactly frame-periodic and the clean '
                                       'covariance is rank deficient: this build (oracle and kernels) drops components with eigenvalue <= 1e-10 max, '
                                       'the reference (pysiib) scores them from rounding noise - 1 % .. 23 % higher raw SIIB on the bench utterances '
                                       '(tests/test_oracle_metrics.py, DESIGN 2); the 1e-4 SIIB claim - against the build\'s oracle, parity with pysiib being unpinned - holds at lengths that are no multiple of 100 '
                                       '(`nonperiodic`, L = 63 871)')
    ee = None
    if a.epoch_equivalent > 0:           # every rank takes part (g_step / d_step all-reduce when world > 1)
        ee = epoch_equivalent(tr, cw, nw, a.epoch_equivalent, a.batch * world)
    if rank == 0:
        if ee is not None:
            out['epoch_equivalent'] = ee
            if world == 1 and a.companions:
                out['epoch_equivalent_qua'] = epoch_with_quality(a, cw, nw)
                out['epoch_from_files'] = epoch_from_files(a)
                if 'cached' in out['epoch_from_files']:
                    out['epoch_from_files_cached'] = out['epoch_from_files'].pop('cached')   # a companion of its own: never the headline `value`
        if world == 1 and a.cpu_utts > 0:
            os.sched_setaffinity(0, cpus_before)         # the CPU leg parallelises like the reference over ALL host cores
            out['cpu_baseline'] = cpu_baseline(metrics, a.length, a.cpu_utts)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
