"""The GAN_epoch loop of the reference ``train_nele.py`` on the MI355X, batched over utterances.

Loop surface kept from the reference (train_nele.py:110-429): per epoch
  G-step (from epoch 2)  :122-156   -> GanTrainer.g_step
  checkpoint             :272-277   -> GanTrainer.save_checkpoint ('enhance-model' / 'intel-model')
  generate D samples     :279-316   -> GanTrainer.generate (G.eval, no grad, mask*beta2, resynthesis, PCM_16)
  true metric targets    :318-340   -> GanTrainer.true_metrics (batched SIIB / HASPI / ESTOI kernels, logistic maps)
  D training, 3 passes + 1/30 history replay :342-426 -> GanTrainer.d_epoch / d_step
What changed on purpose: utterances are processed as batches resident in HBM (the reference is
batch 1 with wav files on disk as the hand-off between G, the metrics and D), the energy
normalisation stays per utterance, and gradients are averaged across ranks with one flat RCCL
all-reduce per model per optimiser step when torch.distributed is initialised.
"""
import ctypes
import os
import random

import numpy as np
import torch
import torch.nn as nn

from . import audio_util as au
from . import dist as ndist
from . import metrics as mt
from . import model as M
from . import ops
from .optim import Adam

# train_nele.py:30-43
TargetMetric = 'siib&haspi&estoi'
GAN_epoch = 500
num_of_sampling = 300
num_of_valid_sample = 480
batch_size = 1
fs = 16000
p_power = (1 / 6)
inv_p = 6
weight_qua = 0.5

_METRIC_FN = {'siib': 'batch_siib', 'estoi': 'batch_estoi', 'haspi': 'batch_haspi'}


def parse_metrics(target_metric):
    names = [m.strip().lower() for m in target_metric.replace(',', '&').split('&') if m.strip()]
    for m in names:
        if m not in _METRIC_FN:
            raise ValueError("unknown metric %r (supported: siib, haspi, estoi)" % m)
    return names


# raw score -> target in (0, 1): 1 / (1 + exp(-k (x - x0))) (intel.py mapping_*_harvard)
_MAPS = {'siib': (0.06, 32.0), 'haspi': (0.95, 2.8), 'estoi': (8.0, 0.25)}


class _MetricFork:
    """See GanTrainer._metric_fork."""

    def __init__(self, tr, inputs):
        self.tr, self.inputs, self.outs, self.used = tr, [t for t in inputs if t is not None], [], []
        self.marks = None
        self.multi = (tr.device.type == 'cuda' and len(tr.metrics) > 1 and tr.metric_streams and not torch.cuda.is_current_stream_capturing())
        if self.multi:
            tr._check_queues()
            self.main = torch.cuda.current_stream()
            self.ev0 = torch.cuda.Event()
            self.ev0.record(self.main)

    def on(self, m):
        return _MetricForkCtx(self, m)

    def out(self, t):
        self.outs.append(t)
        return t

    def mark(self):
        """Record the end of this batch's metric work on its streams (the events join() / wait() makes the consumer wait for)."""
        if self.multi and self.marks is None:
            self.marks = []
            for st in self.used:
                ev = torch.cuda.Event()
                ev.record(st)
                self.marks.append((st, ev))

    def wait(self):
        """The CURRENT stream waits for the marked metric work (the deferred half of join(): run_epoch lets the metrics of a batch run
        under the next batch's generator and resolves all targets at the end of the pass)."""
        if not self.multi:
            return
        self.mark()
        cur = torch.cuda.current_stream()
        for st, ev in self.marks:
            cur.wait_event(ev)
            for t in self.inputs:
                t.record_stream(st)
        for t in self.outs:
            t.record_stream(cur)

    def join(self):
        self.mark()
        self.wait()


class _PendingTargets:
    """Targets whose metric kernels are enqueued on the metric streams; result() makes the current stream wait for them and assembles
    the [B, n_metrics] tensor(s).  (true_metrics(..., defer=True))"""

    def __init__(self, forks, build):
        self.forks, self.build, self.value = forks, build, None
        for f in forks:
            f.mark()

    def result(self):
        if self.build is not None:
            for f in self.forks:
                f.wait()
            self.value, self.build = self.build(), None
        return self.value


class _MetricForkCtx:
    def __init__(self, fork, m):
        self.fork, self.m, self.ctx = fork, m, None

    def __enter__(self):
        f = self.fork
        if not f.multi:
            return 'main'
        name = {'siib': '_side', 'haspi': '_side2'}.get(self.m, '_fside')
        if getattr(f.tr, name) is None:
            setattr(f.tr, name, ops.side_stream(f.tr.device))
        st = getattr(f.tr, name)
        if st not in f.used:
            f.used.append(st)
            st.wait_event(f.ev0)
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        return name[1:]

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class GanTrainer:
    def __init__(self, target_metric=TargetMetric, device='cuda', lr_g=5e-4, lr_d=2.5e-4, use_quality=False, pcm16=True, seed=666,
                 haspi_dither=None, dither_seed=0, quality_scorer=None):
        """quality_scorer: a ``quality.Scorer`` (PESQ / ViSQOL are external host programs, registered with quality.set_backends): with
        ``use_quality`` run_epoch then scores the generated and the pre-enhanced examples for D_Qua's targets where the batches bring none
        ('qua' / 'drc_qua'), and the validation utterances for the learning curve (train_nele.py:216-222, 323-324, 336-337).
        haspi_dither: None = HASPI targets without the IHC firing jitter (deterministic); 'utterance' = the reference's semantics
        (pyhaspi2.py:362-365 dithers every call): standard-normal rows drawn per UTTERANCE ID from a counter-based generator
        (``dither_seed``, utterance id) - not per rank or per batch position, so a sharded run scores an utterance exactly
        as a single-GPU run does.  The ids come with the batch (``utt_ids``; default: position in the batch)."""
        if haspi_dither not in (None, 'utterance'):
            raise ValueError("haspi_dither must be None or 'utterance'")
        self.haspi_dither = haspi_dither
        self.dither_seed = int(dither_seed)
        self.metrics = parse_metrics(target_metric)
        self.device = M._norm_dev(device)            # indexed ('cuda' -> 'cuda:0'): torch.device('cuda') != torch.device('cuda:0')
        torch.manual_seed(seed)                      # same initial weights on every rank
        random.seed(seed)                            # train_nele.py:28
        self.G = M.Generator_Conv1D_cLN().to(self.device)
        self.D = M.Discriminator(nout=len(self.metrics)).to(self.device)
        self.D_Qua = M.Discriminator_Quality().to(self.device) if use_quality else None
        self.optimizer_g = Adam(self.G, lr=lr_g)     # train_nele.py:89-91
        self.optimizer_d = Adam(self.D, lr=lr_d)
        self.optimizer_dqua = Adam(self.D_Qua, lr=lr_d) if use_quality else None
        # flat parameter / gradient buffers now, on the current stream (not lazily inside D.prepare on a side stream: see FlatParams.ensure)
        for m in (self.G, self.D, self.D_Qua):
            if m is not None:
                m.flat_parameters(self.device)
        self.MSELoss = nn.MSELoss()
        self.pcm16 = pcm16
        self.quality_scorer = quality_scorer
        self._quality_pool = None
        self.step_g = 0
        self.step_d = 0
        self.history = []                            # Previous_Discriminator_training_list (train_nele.py:373-403)
        # D inputs of past epochs kept in HBM for the replay; beyond it items move to host memory (_history_trim).  A quarter of the device's
        # memory (72 GB of an MI355X's 288), not a constant: a smaller GPU must spill before it runs out
        self.history_hbm_bytes = (torch.cuda.get_device_properties(self.device).total_memory // 4) if self.device.type == 'cuda' else (64 << 30)
        self._side = None
        self._side2 = None
        self._fside = None
        self._last_enh = None                        # enhanced batch of the last canonical_step (parity tests read it)
        self.prefetched = None                       # input-only work of the next batch (canonical_step(next_batch=...))
        # Early prefetch (small batches): the next batch's input-only work is enqueued at the START of a step, so that its latency chain
        # (SIIB's eigen-decomposition: 3.5 ms at B = 32 whatever else runs) overlaps the whole step instead of heading the next one.
        # Two sets of metric workspaces alternate (batch k's degraded-signal half still needs batch k's eigenvectors while batch
        # k + 1's are computed); _sets[i] is the parked workspace dict while set 1 - i is the active one (self._ws).
        self.early_prefetch_max_batch = 64
        # what a LARGE batch's canonical_step(next_batch=...) enqueues behind its targets: 'features' (STFT / band energies / IMCRA of the
        # next batch fill the D backward pass: 38.1 -> 37.7 ms at B = 256) or 'all' (the metrics' clean-signal halves too: 40.1 ms - the
        # GPU is saturated, they only slow the D-step's critical chain down)
        self.late_prefetch = 'features'
        self.metric_streams = True                   # true_metrics / true_metrics_pair: one stream per metric (False: all on the current stream)
        self._sets = [None, {}]
        self._cur_set = 0
        self._wstreams_plain = None                  # D's weight-gradient streams outside the pipelined step (see _pipeline_queues)
        self._queues_checked = False                 # _pipeline_queues has verified that the three pipelined streams sit on three hardware queues
        self.split_haspi = os.environ.get('NELE_HASPI_SPLIT', '1') != '0'   # HASPI's clean-signal half beside the G-step (A/B switch)
        # metric status, accumulated on the device without a host synchronisation and read by check_status():
        # [SIIB undefined (too few active frames: pysiib raises), SIIB clamped (M / frame caps hit: truncated score),
        #  HASPI below threshold (pyhaspi2.py:357-358 raises)] - one accumulator per stream that folds into it
        self._status = {}
        self._ws = {}                                # metric workspaces of the split objects (multi-GB at B = 256): owned here, freed with the trainer
        self.world = ndist.world_size()
        # Data parallelism ("overlap with the next batch's feature kernels"): the canonical step's LAST action is the all-reduce
        # of D's gradients + Adam-D, and the next step's first work - the features of its batch - needs neither.  With overlap_allreduce the
        # step ends by STARTING that all-reduce (async: RCCL's own stream) and the update is completed (_flush_d: wait, scale, Adam-D) by
        # whatever touches D next - the following canonical_step does so right after enqueuing its features, so the collective's latency
        # hides under them.  One rank: no collective, nothing is deferred.
        self.overlap_allreduce = True
        self._pending_d = None
        self.target_lag = 3                          # run_epoch: batches whose metric targets may be in flight behind the sample-generation loop
        self.clean_cache = None                      # metrics.CleanStateCache (enable_clean_cache): clean-signal halves of SIIB / HASPI kept across epochs
        for m in (self.G, self.D, self.D_Qua):
            if m is not None:
                ndist.broadcast_module_(m, 0)

    # ---------------------------------------------------------------- data-parallel glue
    def _allreduce_grads(self, module, weight=None):
        """One flat all-reduce of the module's gradient bucket.  weight=None: mean over ranks (equal shards).  weight=n: this rank's
        gradient is the mean over its n items (0 = empty step); the result is the mean over all items of all ranks."""
        if self.world > 1:
            g = module.flat_parameters().grad
            if weight is None:
                ndist.allreduce_mean_(g)
            else:
                ndist.allreduce_weighted_mean_(g, float(weight))

    def _flush_d(self):
        """Complete a deferred D update (see overlap_allreduce): wait for the gradient all-reduce, scale, Adam-D."""
        pend, self._pending_d = self._pending_d, None
        if pend is not None:
            pend.wait()
            self.optimizer_d.step()
            self.step_d += 1

    @staticmethod
    def _advance_sn(module):
        """Empty data-parallel step: the power iteration a real step's forward pass would have run (replicas keep identical u, v)."""
        if module is not None and hasattr(module, 'advance_power_iteration'):
            module.advance_power_iteration()

    def _dither(self, x, utt_ids):
        """HASPI dither rows [B, 2, nsub, 32] of this batch (haspi_dither='utterance'), or None."""
        if self.haspi_dither is None:
            return None
        B, L = x.shape
        if utt_ids is None:
            utt_ids = torch.arange(B, dtype=torch.int64, device=x.device)
        return mt.haspi_dither_rows(utt_ids, self.dither_seed, L, device=x.device)

    # ---------------------------------------------------------------- features (dataloader.py:30-42)
    def features(self, clean_wav, noise_wav, lengths=None):
        """wav [B,L] x2 -> dict(clean_band, noise_band [B,T,64], clean_spec [B,T,257] complex64, frames).
        lengths [B] (optional): samples of each utterance inside the padded batch - the reference handles one file of any length at a
        time (dataloader.py:30-42); 'frames' = 1 + lengths // 256 travels with the features to the stages that need it.
        The noise branch (STFT -> IMCRA: recursions that are serial over frames - 0.5 ms at B = 256, T = 251 as one thread per utterance
        and bin, audio_util.noise_band) heads the step's critical path, so it is issued first; the clean STFT + band energies run beside
        it on their own stream."""
        main = torch.cuda.current_stream()
        if self._fside is None:
            self._fside = ops.side_stream(self.device)
        fs_ = self._fside
        lengths = au._i32(lengths, self.device)
        frames = au.frames_of(lengths)
        ev0 = torch.cuda.Event()
        ev0.record(main)
        with torch.cuda.stream(fs_):
            fs_.wait_event(ev0)
            clean_spec, clean_band = au.stft_band(clean_wav, p_power, lengths=lengths)
            evc = torch.cuda.Event()
            evc.record(fs_)
        noise_band = au.noise_band(noise_wav, p_power, lengths=lengths, frames=frames)
        main.wait_event(evc)
        clean_spec.record_stream(main)
        clean_band.record_stream(main)
        if clean_wav.is_cuda:
            clean_wav.record_stream(fs_)
        return {'clean_band': clean_band, 'noise_band': noise_band, 'clean_spec': clean_spec, 'frames': frames, 'lengths': lengths}

    # ---------------------------------------------------------------- G-step (train_nele.py:122-156)
    def g_step(self, clean_band, noise_band, frames=None, weight=None):
        """One optimiser step of G (train_nele.py:130-155).  ``weight``: number of utterances this rank contributes (run_epoch under data
        parallelism: ranks may hold different batch sizes / batch counts; None = plain mean over ranks); ``clean_band=None`` = an empty
        step that only joins the collective (this rank has run out of batches)."""
        self._flush_d()
        if clean_band is None:
            # a real G-step runs D's (and D_Qua's) training-mode forward pass, which advances their spectral-norm u / v once
            self._advance_sn(self.D)
            self._advance_sn(self.D_Qua)
            self.optimizer_g.zero_grad()
            self._allreduce_grads(self.G, 0 if weight is None else weight)
            self.optimizer_g.step()
            self.step_g += 1
            return None
        B = clean_band.shape[0]
        self.D.weight_grad_enabled = False           # D / D_Qua gradients of this step are never applied (train_nele.py:153-155)
        if self.D_Qua is not None:
            self.D_Qua.weight_grad_enabled = False
        self.optimizer_g.zero_grad()
        mask = self.G(clean_band, noise_band)
        din, _ = M.energy_norm_pack(mask, clean_band, noise_band, p_power, inv_p)
        self._last_din = din                         # bench.py re-launches D's forward on it for the isolated roofline figure
        self.D.profile_prefix = 'gstep.'             # bench.py times these launches
        score = self.D.forward_packed(din, frames)
        self.D.profile_prefix = ''
        loss = self.MSELoss(score, torch.ones_like(score))
        if self.D_Qua is not None:
            score_q = self.D_Qua.forward_packed(self.quality_inputs(din), frames)
            loss = loss + weight_qua * self.MSELoss(score_q, torch.ones_like(score_q))
        loss.backward()
        self._allreduce_grads(self.G, weight)
        self.optimizer_g.step()
        self.step_g += 1
        self.D.weight_grad_enabled = True
        if self.D_Qua is not None:
            self.D_Qua.weight_grad_enabled = True
        return loss.detach()

    # ---------------------------------------------------------------- sample generation (train_nele.py:279-316)
    @torch.no_grad()
    def generate(self, clean_band, noise_band, clean_spec, rms_target=0.0, frames=None):
        self.G.eval()
        mask = self.G(clean_band, noise_band)
        alpha2 = M.normed_alpha2(mask, clean_band, inv_p, frames=frames)
        enh_wav = au.gain_istft(alpha2, clean_spec, rms_target=rms_target, pcm16=self.pcm16, frames=frames)
        self.G.train()
        return enh_wav

    # ---------------------------------------------------------------- metric status (intel.py / pysiib / pyhaspi2.py raise; here: count)
    def _note_status(self, which, siib_info=None, haspi_info=None):
        """Fold a metric call's per-utterance status into the device-side counters of accumulator ``which`` on the current stream."""
        acc = self._status.get(which)
        if acc is None:
            acc = self._status[which] = torch.zeros(4, dtype=torch.int64, device=self.device)
        if siib_info is not None:
            st = siib_info[:, 3]
            acc[0:2] += torch.stack((((st & 24) != 0).sum(), ((st & 7) != 0).sum()))
            acc[3:4] += ((st & 32) != 0).sum()       # covariances that went through the eigensolver's repair path (slower, not wrong)
        if haspi_info is not None:
            acc[2:3] += (haspi_info[:, 1] != 0).sum()

    def check_status(self, raise_on_error=True):
        """Synchronising read of everything the step masked or counted on the device: metric scores the reference would have raised on
        (SIIB with too few active frames, HASPI below threshold), truncated SIIB scores (replication / frame caps), and optimiser steps
        skipped because their gradient was not finite (a NaN target or a poisoned eigen-decomposition must not reach the weights).
        Call once per epoch (run_epoch does)."""
        self._flush_d()
        cur = torch.cuda.current_stream() if self.device.type == 'cuda' else None
        for st_ in self._all_side_streams():         # the 'side' accumulators are updated on a side stream BEHIND the event the main stream waits on
            if cur is not None:
                cur.wait_stream(st_)
        tot = torch.zeros(4, dtype=torch.int64)
        for acc in self._status.values():
            tot += acc.cpu()
        # eigh_repaired: SIIB covariances whose cluster tridiagonalisation gave up (its workgroups were not co-resident within the spin
        # limit: another process on the GPU, resident communication kernels) and were redone by one workgroup each - correct scores, but
        # ~3 ms per matrix instead of 25 us: a non-zero count explains a slow step, it is not an error
        st = {'siib_undefined': int(tot[0]), 'siib_clamped': int(tot[1]), 'haspi_below_threshold': int(tot[2]), 'eigh_repaired': int(tot[3]),
              'skipped_g_steps': self.optimizer_g.skipped_steps(), 'skipped_d_steps': self.optimizer_d.skipped_steps(),
              'skipped_dqua_steps': self.optimizer_dqua.skipped_steps() if self.optimizer_dqua is not None else 0}
        if raise_on_error and any(st[k] for k in ('siib_undefined', 'haspi_below_threshold', 'skipped_g_steps', 'skipped_d_steps',
                                                  'skipped_dqua_steps')):
            raise RuntimeError("NELE-GAN step status: %s (undefined metric targets; the reference raises from pysiib / "
                               "pyhaspi2.py:357-358 on such utterances; the affected optimiser steps were skipped)" % st)
        return st

    # ---------------------------------------------------------------- true metric targets (train_nele.py:318-340)
    def enable_clean_cache(self, budget_bytes=None):
        """Keep the clean-signal halves of SIIB (VAD .. KLT eigen-decomposition) and HASPI (the reference-signal chain) of every scored
        utterance in device memory and reuse them whenever the utterance is scored again: the reference draws the same <= 720 training
        files for 500 epochs (train_nele.py:35-38,119,318-340) and recomputes them each time.  Batches must carry ``keys`` (or ``names``):
        one hashable per utterance that stands for its clean waveform (with haspi_dither='utterance': also for its utterance id).
        ~2.1 MB (SIIB) + ~6.5 MB (HASPI) per 4 s utterance.  budget_bytes: device-memory cap (default: a quarter of what is free now);
        beyond it new utterances are recomputed every time.  Scores are bit-identical to recomputation (copies only)."""
        if budget_bytes is None:
            budget_bytes = torch.cuda.mem_get_info(self.device)[0] // 4
        self.clean_cache = mt.CleanStateCache(budget_bytes)
        self.cache_len_quantum = 16384               # metric inputs are padded to multiples of this many samples while the cache is on (_canon)
        import collections
        self._feat_cache = collections.OrderedDict() # run_epoch: features of a batch of (clean, noise) files, keyed by the batch's keys (LRU)
        self._feat_bytes = 0
        self._feat_seen = collections.OrderedDict()  # batch keys seen once (no tensors): a batch's features are kept from its second sighting on
        self._drc_cache = {}                         # run_epoch: D item (input, targets[, quality targets]) of an utterance's pre-enhanced example
        return self.clean_cache

    def _dither_tag(self):
        return None if self.haspi_dither is None else (self.haspi_dither, self.dither_seed)

    def _canon(self, x, ys, lengths):
        """Metric inputs of a batch in the cache's canonical geometry: rows zero-padded to the next multiple of ``cache_len_quantum`` samples
        and explicit per-row lengths.  A cached clean-signal state is laid out for one padded length; a loop that re-draws its batches every
        epoch (fit) pads every batch to its own longest file, so the same utterance would otherwise be looked up under ever new lengths.  The
        scores do not depend on the padding (everything behind a row's own end is zeros and never read; tests/test_varlen_gpu.py)."""
        q = int(getattr(self, 'cache_len_quantum', 16384))
        B, L = x.shape
        Lc = (L + q - 1) // q * q
        if lengths is None:
            lengths = torch.full((B,), L, dtype=torch.int32, device=x.device)
        if Lc == L:
            return x, ys, lengths
        def pad(t):
            o = t.new_zeros((B, Lc))
            o[:, :L] = t
            return o
        return pad(x), [pad(y_) for y_ in ys], lengths

    def _metric(self, m, x, y, which, lengths=None, utt_ids=None, keys=None):
        if self.clean_cache is not None and keys is not None and m in ('siib', 'haspi'):
            # split form: the clean-signal half comes from the cache when every utterance of the batch is in it (bit-identical to the
            # one-call form: phase 0 runs exactly phases 3 and 4 back to back)
            if m == 'siib':
                sp = mt.SiibSplit(x, lengths=lengths, owner=self._ws)
                sp.clean_part(cache=self.clean_cache, keys=keys)
                raw, mapped = sp.degraded_part(y)
                self._note_status(which, siib_info=sp.info)
            else:
                hp = mt.HaspiSplit(x, lengths=lengths, owner=self._ws)
                dz = self._dither(x, utt_ids)
                hp.clean_part(dither=dz, cache=self.clean_cache, keys=keys, dither_tag=self._dither_tag())
                raw, mapped = hp.degraded_part(y, dither=dz)
                self._note_status(which, haspi_info=hp.info)
            return raw, mapped
        if m == 'siib':
            raw, mapped, info = mt.batch_siib(x, y, return_info=True, lengths=lengths)
            self._note_status(which, siib_info=info)
        elif m == 'haspi':
            raw, mapped, info = mt.batch_haspi(x, y, return_info=True, lengths=lengths, dither=self._dither(x, utt_ids))
            self._note_status(which, haspi_info=info)
        else:
            raw, mapped = mt.batch_estoi(x, y, lengths=lengths)
        return raw, mapped

    @staticmethod
    def enhanced_lengths(lengths):
        """Samples of the resynthesised signal of an utterance with ``lengths`` samples: 256 * (L // 256) (audio_util.py:60-65), which is
        also what the metrics see (min of the clean and the enhanced length, audio_util.py:134-141)."""
        return None if lengths is None else (torch.div(lengths, 256, rounding_mode='floor') * 256).to(torch.int32)

    def _metric_fork(self, inputs):
        """Fork / join of the metric calls of one batch: SIIB on the metric stream, HASPI on the second one, everything else (ESTOI) on the
        feature stream - three dependency chains that share nothing but their inputs (canonical_step runs them the same way).  One
        metric, a CPU trainer or a stream capture in progress: everything stays on the current stream."""
        return _MetricFork(self, inputs)

    @torch.no_grad()
    def true_metrics(self, clean_wav, enh_wav, noise_wav, norm=True, lengths=None, resynth=True, utt_ids=None, defer=False, keys=None):
        """[B, n_metrics] targets of (clean, enhanced + noise) (audio_util.py:120-203).  lengths [B]: samples of each utterance inside the
        padded batch; resynth=True: ``enh_wav`` came out of ``generate`` (each row holds 256 * (L // 256) samples); False: ``lengths``
        already are min(clean, enhanced) per utterance (the pre-enhanced 'DRC' examples, audio_util.py:267-321).
        utt_ids [B] int64: utterance ids for the per-utterance HASPI dither (haspi_dither='utterance').
        defer: return a _PendingTargets instead - the kernels are enqueued on the metric streams, the caller's stream is not made to
        wait for them until .result() (run_epoch: a batch's metrics run under the next batch's generator).
        keys [B] (host list of hashables, one per utterance): with enable_clean_cache() the clean-signal halves are reused across calls."""
        L = min(clean_wav.shape[1], enh_wav.shape[1])          # audio_util.py:134-141
        x = clean_wav[:, :L].contiguous()
        y = (enh_wav[:, :L] + noise_wav[:, :L]).contiguous()
        lengths = au._i32(lengths, self.device)
        if resynth:
            lengths = self.enhanced_lengths(lengths)
        elif lengths is not None:
            lengths = torch.clamp(lengths, max=L)
        if self.clean_cache is not None and keys is not None:
            x, (y,), lengths = self._canon(x, [y], lengths)
        cols = []
        fork = self._metric_fork((x, y, lengths))             # (everything allocated on THIS stream that the metric streams read: kept alive and
                                                               #  marked as used there until the targets are waited for)
        for m in self.metrics:
            with fork.on(m) as which:
                raw, mapped = self._metric(m, x, y, which, lengths, utt_ids, keys)
                cols.append(fork.out(mapped if norm else raw))
        if defer:
            return _PendingTargets([fork], lambda: torch.stack(cols, dim=1))
        fork.join()
        return torch.stack(cols, dim=1)

    @torch.no_grad()
    def true_metrics_pair(self, clean_wav, enh_wav, drc_wav, noise_wav, norm=True, lengths=None, drc_lengths=None, utt_ids=None, defer=False,
                          keys=None, lengths_host=None, drc_lengths_host=None):
        """Targets of TWO degraded versions of one clean batch - the generated example and the pre-enhanced ('DRC') one, which the loop
        scores back to back (train_nele.py:318-340) - with the clean-signal work done once: SIIB's VAD / clean spectra / covariance /
        eigen-decomposition (its KLT basis) and HASPI's whole reference-signal chain depend on the clean signal only
        (metrics.SiibSplit / HaspiSplit: clean_part() once, degraded_part() twice).  -> (targets_enh, targets_drc), each bit-identical
        to a true_metrics() call of its own.  Falls back to two full calls when the two comparisons do not see the same clean samples
        (different truncation lengths, audio_util.py:134-137)."""
        lengths = au._i32(lengths, self.device)
        L = min(clean_wav.shape[1], enh_wav.shape[1])
        Ld = min(clean_wav.shape[1], drc_wav.shape[1])
        ml_e = self.enhanced_lengths(lengths)
        ml_d = None
        if drc_lengths is not None or lengths is not None:
            full = lambda t, w: torch.full((w.shape[0],), w.shape[1], dtype=torch.int32, device=self.device) if t is None else au._i32(t, self.device)
            ml_d = torch.clamp(torch.minimum(full(drc_lengths, drc_wav), full(lengths, clean_wav)), max=Ld)
        if ml_e is not None and ml_d is not None and lengths_host is not None and (drc_lengths_host is not None or drc_lengths is None):
            # the same comparison on the host copies of the lengths a loader hands over (dataio.FileBatches): no device read-back, which
            # would stall the thread that enqueues the GPU work once per batch
            lh = np.asarray(lengths_host, dtype=np.int64)
            dh = np.full_like(lh, drc_wav.shape[1]) if drc_lengths_host is None else np.asarray(drc_lengths_host, dtype=np.int64)
            eq = bool(np.array_equal(256 * (lh // 256), np.minimum(np.minimum(dh, lh), Ld)))
        else:
            eq = (ml_e is None and ml_d is None) or (ml_e is not None and ml_d is not None and bool(torch.equal(ml_e, ml_d)))
        same = (L == Ld) and eq
        if not same:
            # (the pre-enhanced comparison sees the clean file cut to another length: its clean-signal state is cached under its own key)
            kd = None if keys is None else [('drc', k_) for k_ in keys]
            if defer:
                pa = self.true_metrics(clean_wav, enh_wav, noise_wav, norm=norm, lengths=lengths, utt_ids=utt_ids, defer=True, keys=keys)
                pb = self.true_metrics(clean_wav, drc_wav, noise_wav, norm=norm, lengths=ml_d, resynth=False, utt_ids=utt_ids, defer=True, keys=kd)
                return _PendingTargets([], lambda: (pa.result(), pb.result()))
            return (self.true_metrics(clean_wav, enh_wav, noise_wav, norm=norm, lengths=lengths, utt_ids=utt_ids, keys=keys),
                    self.true_metrics(clean_wav, drc_wav, noise_wav, norm=norm, lengths=ml_d, resynth=False, utt_ids=utt_ids, keys=kd))
        x = clean_wav[:, :L].contiguous()
        ys = [(enh_wav[:, :L] + noise_wav[:, :L]).contiguous(), (drc_wav[:, :L] + noise_wav[:, :L]).contiguous()]
        if self.clean_cache is not None and keys is not None:
            x, ys, ml_e = self._canon(x, ys, ml_e)
            ml_d = ml_e
        pick = (lambda r, m_: m_) if norm else (lambda r, m_: r)
        cols = [{}, {}]
        fork = self._metric_fork([x] + ys + [ml_e, ml_d])
        for m in self.metrics:
            with fork.on(m) as which:
                if m == 'siib':
                    sp = mt.SiibSplit(x, lengths=ml_e, owner=self._ws)
                    sp.clean_part(cache=self.clean_cache, keys=keys)
                    for k, y in enumerate(ys):
                        raw, mapped = sp.degraded_part(y)
                        cols[k][m] = fork.out(pick(raw, mapped).clone())
                        self._note_status(which, siib_info=sp.info)
                elif m == 'haspi':
                    hp = mt.HaspiSplit(x, lengths=ml_e, owner=self._ws)
                    dz = self._dither(x, utt_ids)
                    hp.clean_part(dither=dz, cache=self.clean_cache, keys=keys, dither_tag=self._dither_tag())
                    for k, y in enumerate(ys):
                        raw, mapped = hp.degraded_part(y, dither=dz)
                        cols[k][m] = fork.out(pick(raw, mapped).clone())
                        self._note_status(which, haspi_info=hp.info)
                else:
                    for k, y in enumerate(ys):
                        raw, mapped = mt.batch_estoi(x, y, lengths=ml_e)
                        cols[k][m] = fork.out(pick(raw, mapped))
        if defer:
            return _PendingTargets([fork], lambda: tuple(torch.stack([c[m] for m in self.metrics], dim=1) for c in cols))
        fork.join()
        return tuple(torch.stack([c[m] for m in self.metrics], dim=1) for c in cols)

    # ---------------------------------------------------------------- D-step (train_nele.py:349-367)
    def d_inputs(self, enh_wav, noise_band, clean_band, lengths=None, resynth=True):
        """dataloader.py:54-84: features of the enhanced wav, stacked (enhanced, noise, clean).  lengths: of the ORIGINAL utterances
        (resynth=True: the enhanced ones hold 256 * (L // 256) samples and have the same frame count; False: ``lengths`` are the
        enhanced files' own, e.g. the pre-enhanced 'DRC' examples)."""
        lengths = au._i32(lengths, self.device)
        _, enh_band = au.stft_band(enh_wav, p_power, want_spec=False, lengths=self.enhanced_lengths(lengths) if resynth else lengths)
        T = clean_band.shape[1]
        if enh_band.shape[1] != T:                   # a pre-enhanced file padded / cut to another frame count than the clean batch
            eb = enh_band.new_zeros((enh_band.shape[0], T, 64))
            n = min(T, enh_band.shape[1])
            eb[:, :n] = enh_band[:, :n]
            enh_band = eb
        return ops.d_pack(enh_band, noise_band, clean_band)

    @staticmethod
    def quality_inputs(din):
        """[enhanced, clean] of the packed D input (dataloader.py:83: the D_Qua item drops the noise channel)."""
        din_q = torch.zeros_like(din)
        din_q[..., 0] = din[..., 0]
        din_q[..., 1] = din[..., 2]
        return din_q

    def d_step(self, din, target, target_qua=None, weight=None, frames=None, has_qua=None, items=None):
        """One optimiser step of D on (din, target) - and of D_Qua on ([enh, clean], target_qua) when the quality discriminator is
        enabled and quality targets are given (train_nele.py:356-365).  ``weight``: number of items this rank contributes (d_epoch
        under data parallelism; None = plain mean over ranks); ``din=None`` = an empty step that only joins the collectives.
        ``items``: the loss runs over the first ``items`` rows only (the rest are fill rows of a padded batch, _padded_chunks).
        ``has_qua``: whether THIS optimiser step includes D_Qua - under data parallelism it must be the same on every rank (the D_Qua
        all-reduce is a collective), so d_epoch decides it once per pass for all ranks; None = decide from ``target_qua`` (single rank)."""
        self._flush_d()
        if has_qua is None:
            has_qua = self.D_Qua is not None and target_qua is not None
        if has_qua and self.D_Qua is None:
            raise ValueError("d_step: quality targets given but the trainer has no D_Qua (use_quality=False)")
        if has_qua and din is not None and target_qua is None:
            raise ValueError("d_step: this step trains D_Qua (has_qua) but the batch carries no quality targets")
        self.optimizer_d.zero_grad()
        if din is None:
            self._advance_sn(self.D)                 # empty step: the power iteration the other ranks' forward passes run
        score = self.D.forward_packed(din, frames) if din is not None else None
        score_qua = None
        if has_qua:
            self.optimizer_dqua.zero_grad()
            if din is not None:                      # both forward passes first, as the reference (train_nele.py:356-357)
                score_qua = self.D_Qua.forward_packed(self.quality_inputs(din), frames)
            else:
                self._advance_sn(self.D_Qua)
        if items is not None and score is not None and items < score.shape[0]:
            score, target = score[:items], target[:items]
            if score_qua is not None:
                score_qua, target_qua = score_qua[:items], target_qua[:items]
        loss = self._d_finish(score, target, weight)
        if has_qua:
            if score_qua is not None:
                loss_qua = self.MSELoss(score_qua, target_qua)
                loss_qua.backward()
                self.last_loss_qua = loss_qua.detach()
            self._allreduce_grads(self.D_Qua, weight)
            self.optimizer_dqua.step()
        return loss

    def _d_finish(self, score, target, weight=None, defer=False):
        """loss, backward, gradient all-reduce, Adam-D.  defer (canonical_step with overlap_allreduce on several ranks, equal shards): the
        all-reduce is started asynchronously and the update is left to _flush_d()."""
        loss = None
        if score is not None:
            loss = self.MSELoss(score, target)
            loss.backward()
        if defer and self.world > 1 and weight is None and self.overlap_allreduce:
            self._pending_d = ndist.PendingMean(self.D.flat_parameters().grad)
            return loss.detach() if loss is not None else None
        self._allreduce_grads(self.D, weight)
        self.optimizer_d.step()
        self.step_d += 1
        return loss.detach() if loss is not None else None

    # ---------------------------------------------------------------- one canonical step
    def _use_set(self, i):
        """Make metric-workspace set i (0 or 1) the active one (self._ws); the other set is parked."""
        if i == self._cur_set:
            return
        self._sets[self._cur_set] = self._ws
        self._ws = self._sets[i]
        self._sets[i] = None
        self._cur_set = i

    def _all_side_streams(self):
        return [st_ for st_ in (self._side, self._side2, self._fside) if st_ is not None]

    def _shares_queue(self, a, b, spin_us=300.0):
        """ops.shares_queue on this trainer's device."""
        return ops.shares_queue(a, b, self.device, spin_us)

    def _check_queues(self):
        """Once per trainer: the second metric stream and the feature stream must not share the METRIC stream's hardware queue (SIIB's
        latency chain through the eigensolver lives there: whatever else is multiplexed onto that queue runs in submission order with
        it, and the chain waits behind every kernel in between - measured on true_metrics at B = 64: SIIB + HASPI 7.4 ms, + ESTOI on a
        third stream that happened to share the queue 10.6 ms).  A stream that does is replaced by a fresh one that does not."""
        dev = self.device
        if self._queues_checked or dev.type != 'cuda' or torch.cuda.is_current_stream_capturing():
            return
        for name in ('_side', '_side2', '_fside'):
            if getattr(self, name) is None:
                setattr(self, name, ops.side_stream(dev))
        if self._side == torch.cuda.current_stream(dev):
            return
        self._queues_checked = True
        for name in ('_side2', '_fside'):
            tries = 0
            while tries < 8 and (self._shares_queue(self._side, getattr(self, name)) or
                                 (name == '_fside' and self._shares_queue(self._side2, self._fside))):
                setattr(self, name, torch.cuda.Stream(device=dev))      # (three hardware queues serve the side streams: one each)
                tries += 1

    def _pipeline_queues(self, on):
        """The HIP runtime multiplexes all streams onto four hardware queues: the default stream has its own; side streams get
        the other three, in an order the runtime decides, and share them beyond the third.  In a plain step the trainer's six side streams
        pair up in phases that never overlap.  In the pipelined step the next batch's eigen-decomposition occupies the metric stream's
        queue with 1 ms kernels for the WHOLE step and whatever shares that queue waits behind them (kernel trace at B = 32: G's or D's
        weight gradients, + 0.9 ms per step, whichever of their streams the runtime happened to put there).  So while pipelining only
        THREE side streams are in use, one queue each: the metric stream carries the long chain alone; the weight gradients of G and D
        ride on the feature stream and the second metric stream, which are idle in those phases.  That the three do sit on three
        queues is MEASURED once per trainer (_shares_queue: an idle wave parked on the metric stream, a trivial kernel timed on the other);
        a stream that shares the metric stream's queue is replaced by a fresh one that does not."""
        dev = self.device
        if self._side is None:
            self._side = ops.side_stream(dev)
        if self._side2 is None:
            self._side2 = ops.side_stream(dev)
        if self._fside is None:
            self._fside = ops.side_stream(dev)
        self._check_queues()                            # (plain steps too: SIIB's and HASPI's streams on one queue run one after the other)
        if on and self._wstreams_plain is None:
            self._wstreams_plain = (self.D._wstream, self.G._wstream)
            self.D._wstream = (self._fside, self._side2)
            self.G._wstream = self._fside
        elif not on and self._wstreams_plain is not None:
            self.D._wstream, self.G._wstream = self._wstreams_plain
            self._wstreams_plain = None

    def _input_only_work(self, clean_wav, noise_wav, lengths, after, with_features, utt_ids=None, with_metrics=True):
        """Everything of a step that needs only its INPUTS, enqueued on the side streams behind event ``after``: the clean-signal
        halves of SIIB (VAD .. eigen-decomposition .. clean projections) and HASPI (the whole reference-signal chain) and, with
        ``with_features``, the features of both waveforms.  -> dict."""
        if self._side is None:
            self._side = ops.side_stream(self.device)
        if self._side2 is None:
            self._side2 = ops.side_stream(self.device)
        side, side2 = self._side, self._side2
        L = 256 * (clean_wav.shape[1] // 256)              # length of the resynthesised signal (audio_util.py:76-110)
        lengths = au._i32(lengths, self.device)
        mlens = self.enhanced_lengths(lengths)             # what the metrics see of each utterance (audio_util.py:134-141)
        w = {'clean': clean_wav, 'noise': noise_wav, 'lengths': lengths, 'mlens': mlens, 'split': None, 'hsplit': None, 'feats': None,
             'dither': None, 'utt_ids': utt_ids, 'set': self._cur_set, 'clean_done': None, 'clean_stream': None, 'metrics': bool(with_metrics)}
        with torch.cuda.stream(side):
            side.wait_event(after)
            w['x'] = clean_wav[:, :L].contiguous()
            x_ready = torch.cuda.Event()
            x_ready.record(side)
            if 'siib' in self.metrics and with_metrics:
                w['split'] = mt.SiibSplit(w['x'], lengths=mlens, owner=self._ws)
                w['split'].clean_part()
            w['clean_done'] = torch.cuda.Event()
            w['clean_done'].record(side)
            w['clean_stream'] = side
        if 'haspi' in self.metrics and self.split_haspi and with_metrics:
            # HASPI's reference-signal half (ear model .. modulation filters of the CLEAN signal) needs no enhanced signal either
            with torch.cuda.stream(side2):
                side2.wait_event(after)
                side2.wait_event(x_ready)
                w['hsplit'] = mt.HaspiSplit(w['x'], lengths=mlens, owner=self._ws)
                w['dither'] = self._dither(w['x'], utt_ids)
                w['hsplit'].clean_part(dither=w['dither'])
        if with_features:
            if self._fside is None:
                self._fside = ops.side_stream(self.device)
            fs_ = self._fside
            with torch.cuda.stream(fs_):
                fs_.wait_event(after)
                frames = au.frames_of(lengths)
                clean_spec, clean_band = au.stft_band(clean_wav, p_power, lengths=lengths)
                noise_spec, _ = au.stft_band(noise_wav, p_power, want_band=False, lengths=lengths)
                _, noise_band = au.imcra_band(noise_spec, p_power, frames=frames)
                w['feats'] = {'clean_band': clean_band, 'noise_band': noise_band, 'clean_spec': clean_spec, 'frames': frames, 'lengths': lengths}
                w['feats_ready'] = torch.cuda.Event()
                w['feats_ready'].record(fs_)
        return w

    def prefetch(self, clean_wav, noise_wav, lengths=None, after=None, utt_ids=None, with_metrics=True):
        """The input-only work of the NEXT batch, as the reference's DataLoader workers prepare the features of upcoming items
        while the current one trains (dataloader.py:86-92, 8 workers).  canonical_step(..., next_batch=...) calls this at the point where
        the current step's targets are done, so that the work fills the D backward pass (the one phase of a step in which the side
        streams are idle); pass the returned object as ``pre`` to the next canonical_step.  Results are bit-identical to computing
        the same things inside the step (tests/test_step_parity_gpu.py)."""
        if after is None:
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
        return self._input_only_work(clean_wav, noise_wav, lengths, after, with_features=True, utt_ids=utt_ids, with_metrics=with_metrics)

    def canonical_step(self, clean_wav, noise_wav, feats=None, lengths=None, pre=None, next_batch=None, utt_ids=None, early=None):
        """features -> G-step -> generate -> true metrics -> D-step on the same batch.  lengths [B] (optional): samples of each
        utterance inside the padded batch (every utterance needs >= 21 frames, i.e. 5120 samples, for D).
        pre: what prefetch() returned for THIS batch (its input-only work is then already in flight or done);
        next_batch: (clean, noise[, lengths[, ids]]) of the following step - its input-only work is enqueued behind this step's targets
        (large batches: it fills the D backward pass) or, ``early`` (default: batches of at most ``early_prefetch_max_batch``), at the
        start of this step on the second set of side streams and workspaces: a small batch leaves most of the GPU idle and its step
        is as long as its longest dependent chain - SIIB's clean-signal half with the eigen-decomposition, 4.2 of 5.8 ms at B = 32 -
        which then runs a step ahead.  The result is left in ``self.prefetched``; bit-identical either way."""
        # The metric kernels run on a side stream.  (1) Everything SIIB derives from the CLEAN signal alone - VAD, clean spectra,
        # the covariance and its eigen-decomposition (the KLT basis) - is enqueued first and runs beside features / G-step /
        # generate.  (2) Once the enhanced signal exists the remaining metric work follows on the side stream while the main
        # stream runs D's forward pass, which does not need the targets; the loss waits for them.
        main = torch.cuda.current_stream()
        if self._pending_d is not None:
            # the previous step's D update is still in flight (its gradient all-reduce runs on the collective library's stream): this batch's
            # features need neither D nor its gradients, so they are enqueued first and the collective finishes under them
            if feats is None and (pre is None or pre.get('feats') is None):
                feats = self.features(clean_wav, noise_wav, lengths)
            self._flush_d()
        L = 256 * (clean_wav.shape[1] // 256)              # length of the resynthesised signal (audio_util.py:76-110)
        start = torch.cuda.Event()
        start.record(main)
        B_, T_ = clean_wav.shape[0], 1 + clean_wav.shape[1] // 256
        if early is None:
            early = B_ <= self.early_prefetch_max_batch
        early = bool(early) and next_batch is not None and not torch.cuda.is_current_stream_capturing()
        if pre is not None:
            self._use_set(pre.get('set', self._cur_set))    # the workspaces that hold this batch's clean-signal halves
        self._pipeline_queues(early)
        if self._fside is None:
            self._fside = ops.side_stream(self.device)
        if self._side2 is None:
            self._side2 = ops.side_stream(self.device)
        # D's spectral-norm iteration and weight layouts depend on its parameters only: they run on a side stream ahead of each of D's two
        # forward passes (beside the generator's forward pass / beside generate) instead of at the head of those passes.  With a
        # prefetched batch the metric side stream already carries HASPI's clean half: the feature stream (idle then) takes it.
        p1 = self._fside if (pre is not None and not early) else self._side2
        with torch.cuda.stream(p1):
            p1.wait_event(start)                           # after the previous step's D update
            self.D.prepare(B_, T_, self.device)
        if pre is not None:
            assert pre['clean'] is clean_wav and pre['noise'] is noise_wav, "canonical_step: `pre` belongs to another batch"
        if pre is None or not pre['metrics']:
            # nothing prefetched, or the features only (late_prefetch = 'features'): the metrics' clean-signal halves start with the step
            mine_ = self._input_only_work(clean_wav, noise_wav, lengths, start, with_features=False, utt_ids=utt_ids)
            if pre is not None:
                mine_['feats'], mine_['feats_ready'] = pre['feats'], pre['feats_ready']
            pre = mine_
        early_pre = None
        if early:
            # the next batch's input-only work, now: features on the feature stream, SIIB's clean-signal half on the metric stream (whose
            # hardware queue it has to itself, _pipeline_queues), HASPI's on the second metric stream - into the OTHER workspace set
            mine = self._cur_set
            self._use_set(1 - mine)
            early_pre = self._input_only_work(next_batch[0], next_batch[1], next_batch[2] if len(next_batch) > 2 else None, start,
                                              with_features=True, utt_ids=next_batch[3] if len(next_batch) > 3 else None)
            self._use_set(mine)
        side = self._side
        lengths, mlens, x, split, hsplit = pre['lengths'], pre['mlens'], pre['x'], pre['split'], pre['hsplit']
        if feats is None and pre['feats'] is not None:
            feats = pre['feats']
            main.wait_event(pre['feats_ready'])
            for t in (feats['clean_band'], feats['noise_band'], feats['clean_spec']):
                t.record_stream(main)
        f = feats or self.features(clean_wav, noise_wav, lengths)
        frames = f.get('frames')
        lg = self.g_step(f['clean_band'], f['noise_band'], frames)
        gdone = torch.cuda.Event()
        gdone.record(main)                                 # the G-step's backward pass is the last reader of D's current weight layouts
        # second D.prepare: on the feature side stream when the metric side stream is busy with HASPI's clean part
        pstream = self._fside if (hsplit is not None and self._fside is not None) else self._side2
        with torch.cuda.stream(pstream):
            pstream.wait_event(gdone)
            self.D.prepare(B_, T_, self.device)
        enh = self.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=frames)
        assert enh.shape[1] == L
        self._last_enh = enh
        ready = torch.cuda.Event()
        ready.record(main)
        if self._side2 is None:
            self._side2 = ops.side_stream(self.device)
        side2 = self._side2
        cols = {}
        if early:
            # the metric stream is busy with the next batch's clean-signal chain: this batch's degraded-signal half (and ESTOI behind it)
            # takes the feature stream, behind the event that closed its own clean-signal half
            side = self._fside
        with torch.cuda.stream(side):
            side.wait_event(ready)
            if pre['clean_done'] is not None and side != pre['clean_stream']:   # (same stream: FIFO; a self-wait also upsets ROCm's stream capture)
                side.wait_event(pre['clean_done'])
            y = (enh + noise_wav[:, :L]).contiguous()
            y_ready = torch.cuda.Event()
            y_ready.record(side)
            if split is not None:
                cols['siib'] = split.degraded_part(y)[1]
            if 'estoi' in self.metrics and 'haspi' in self.metrics:
                cols['estoi'] = mt.batch_estoi(x, y, lengths=mlens)[1]    # HASPI's degraded-signal chain owns the other stream: ESTOI rides behind SIIB
        haspi_info = None
        with torch.cuda.stream(side2):                     # the cheaper metrics beside SIIB's degraded-signal part
            side2.wait_event(start)
            side2.wait_event(y_ready)
            for m in self.metrics:
                if m == 'haspi' and hsplit is not None:
                    cols[m] = hsplit.degraded_part(y, dither=pre['dither'])[1]
                    haspi_info = hsplit.info
                elif m == 'haspi':
                    _, cols[m], haspi_info = mt.batch_haspi(x, y, return_info=True, lengths=mlens, dither=self._dither(x, pre.get('utt_ids')))
                elif m != 'siib' and m not in cols:
                    cols[m] = getattr(mt, _METRIC_FN[m])(x, y, lengths=mlens)[1]
            others = torch.cuda.Event()
            others.record(side2)
            # status accounting behind the events the main stream waits for (off the critical path), each metric on its own stream's accumulator
            self._note_status('side2', haspi_info=haspi_info)
        with torch.cuda.stream(side):
            done = torch.cuda.Event()
            done.record(side)                               # SIIB's degraded part (and ESTOI behind it) is complete
            self._note_status('fside' if early else 'side', siib_info=split.info if split is not None else None)
        for t in (x, y):
            t.record_stream(side2)
            t.record_stream(side)
        din = self.d_inputs(enh, f['noise_band'], f['clean_band'], lengths)
        self.optimizer_d.zero_grad()
        score = self.D.forward_packed(din, frames)
        for t in (x, y):
            t.record_stream(main)
        enh.record_stream(side)
        clean_wav.record_stream(side)
        noise_wav.record_stream(side)
        self.prefetched = early_pre
        if next_batch is not None and early_pre is None:    # the next batch's input-only work fills the D backward pass
            self.prefetched = self.prefetch(next_batch[0], next_batch[1], next_batch[2] if len(next_batch) > 2 else None, after=done,
                                            utt_ids=next_batch[3] if len(next_batch) > 3 else None, with_metrics=self.late_prefetch == 'all')
        # Both metric streams join the MAIN stream (behind D's forward pass, which does not need the targets), and the [B, n_metrics]
        # target tensor is stacked there.  (Until round 4 the second metric stream joined the first one, which stacked: ROCm 7's stream
        # capture segfaults in hipStreamEndCapture on that side-stream-into-side-stream join - tools/graph_probe2.py v3.)
        main.wait_event(done)
        main.wait_event(others)
        for v in cols.values():
            v.record_stream(main)
        tgt = torch.stack([cols[m] for m in self.metrics], dim=1)
        ld = self._d_finish(score, tgt, defer=True)
        if torch.cuda.is_current_stream_capturing():
            self._join_side_streams(main)                  # a captured step must end with every forked stream joined back
        return lg, ld, tgt

    def _join_side_streams(self, main):
        for st_ in tuple(self._all_side_streams()) + tuple(self.D._wstream or ()) + ((self.G._wstream,) if self.G._wstream is not None else ()):
            if st_ is not None:
                main.wait_stream(st_)

    # ---------------------------------------------------------------- D epoch: 3 passes + replay (train_nele.py:342-426)
    @staticmethod
    def _padded_chunks(lst, batch, round_to=16, fill=False):
        """Shuffled sample list -> batches of at most ``batch`` items in list order.  The reference trains D at batch 1 on utterances
        of any length (train_nele.py:349-367); here the items of a batch are zero-padded along the frame axis to the batch's longest
        (rounded up to a multiple of ``round_to`` so that few distinct buffer shapes occur) and carry their own frame counts, which
        D's pooling honours (nele_gap_mlp_fwd_var).  -> list of (din [b,64,T,4], target [b,n], target_qua [b,2] | None, frames [b],
        items): the loss runs over the first ``items`` rows.  ``fill``: the short last batch of a pass that has full ones is filled up to
        the next multiple of an eighth of ``batch`` (at least 8) rows with all-zero items (outside the loss: zero gradient rows) - the
        replay list grows every epoch, and a batch size that has not occurred before costs a set of activation buffers and recorded
        passes (tens of milliseconds) for one step."""
        out = []
        for k in range(0, len(lst), batch):
            ch = lst[k:k + batch]
            n = len(ch)
            q = max(8, batch // 8)                                       # (row buckets: an eighth of a batch - a handful of shapes, little fill)
            rows = min(batch, (n + q - 1) // q * q) if (fill and n < batch and len(lst) > batch) else n
            Ts = [int(c[0].shape[1]) for c in ch]
            Tm = (max(Ts) + round_to - 1) // round_to * round_to
            if all(t == Ts[0] for t in Ts):
                Tm = Ts[0]
            if rows == n and Tm == Ts[0] and all(t == Ts[0] for t in Ts):
                din, frames = torch.stack([c[0] for c in ch]), None
            elif ch[0][0].is_cuda:
                # one gather launch per 64 items (zero padding and the device-side frame counts included) instead of a copy per item and a
                # host -> device copy of the frame list per batch
                din, frames = ops.d_gather([c[0] for c in ch], rows, Tm, want_frames=not all(t == Tm for t in Ts))
            else:
                din = ch[0][0].new_zeros((rows, 64, Tm, 4))
                for r, c in enumerate(ch):
                    din[r, :, :Ts[r]] = c[0]
                frames = None if all(t == Tm for t in Ts) else torch.tensor(Ts + [Tm] * (rows - n), dtype=torch.int32, device=din.device)
            hq = [len(c) > 2 and c[2] is not None for c in ch]
            if any(hq) != all(hq):
                raise ValueError("d_epoch: quality targets must be given for every sample of a pass or for none")

            def rows_of(ts):
                t = torch.stack(ts)
                return t if rows == n else torch.cat([t, t.new_zeros((rows - n,) + tuple(t.shape[1:]))])
            tq = rows_of([c[2] for c in ch]) if hq[0] else None
            out.append((din, rows_of([c[1] for c in ch]), tq, frames, n))
        return out

    def _d_pass(self, lst, batch):
        random.shuffle(lst)
        # D_Qua is stepped in this pass iff the samples carry quality targets: one decision per pass, the same on every rank (a rank
        # whose shard is shorter joins the D_Qua collectives with empty steps; mixed presence is an error).  The decision AND the error
        # are collective: every rank learns of a mixed pass in the same all-reduce and raises, so no rank enters the step loop (and its
        # gradient all-reduces) alone and waits for the collective timeout.
        hq = [len(c) > 2 and c[2] is not None for c in lst]
        mixed = int(any(hq) != all(hq))
        with_q = int(bool(hq) and all(hq))
        without_q = int(bool(hq) and not any(hq))
        chunks = self._padded_chunks(lst, batch, fill=True) if not mixed else []
        n_steps = len(chunks)
        if self.world > 1:
            # ranks hold different shards: every rank must join the same number of all-reduces
            n_steps, with_q, without_q, mixed = ndist.allreduce_max_ints([n_steps, with_q, without_q, mixed], self.device)
        if mixed:
            raise ValueError("d_epoch: quality targets must be given for every sample of a pass or for none")
        if with_q and without_q:
            raise ValueError("d_epoch: some ranks carry quality targets and others do not")
        has_qua = int(self.D_Qua is not None and bool(with_q))
        for k in range(n_steps):
            if k < len(chunks):
                din, tgt, tq, frames, items = chunks[k]
                self.d_step(din, tgt, tq if has_qua else None, weight=items if self.world > 1 else None, frames=frames,
                            has_qua=bool(has_qua), items=items)
            else:
                self.d_step(None, None, None, weight=0, has_qua=bool(has_qua))
        return n_steps

    def d_epoch(self, samples, batch=32):
        """samples: list of (din [64,T,4], target [n]) or (din, target, target_qua [2]) tensors of this epoch (generated +
        pre-enhanced 'DRC' examples), this rank's shard.  Three passes as train_nele.py:342-426."""
        cur = list(samples)
        self._d_pass(cur, batch)                                        # pass A (:349-367)
        n_hist = len(self.history)
        if self.world > 1:
            # the same replay positions on every rank (drawn on rank 0, broadcast): shards have equal history lengths up to the
            # remainder, so positions are drawn below the shortest one
            n_hist = -ndist.allreduce_max_int(-n_hist, self.device)
            idx = ndist.replay_indices(n_hist, 30, seed=self.step_d, device=self.device if ndist.backend_is_nccl() else 'cpu')
            replay = [self.history[i] for i in idx]
        else:
            random.shuffle(self.history)                                # :373-376
            replay = self.history[0:n_hist // 30]
        replay = [self._on_device(it) for it in replay]                 # (items of old epochs may live in host memory: _history_trim)
        self._d_pass(replay + cur, batch)                               # pass B: 1/30 of the history + current (:380-398)
        self.history = self.history + cur                               # :403
        self._history_trim()
        self._d_pass(cur, batch)                                        # pass C (:406-424)

    def _on_device(self, item):
        return item if item[0].device == self.device else tuple(None if t is None else t.to(self.device, non_blocking=True) for t in item)

    def _history_trim(self):
        """The reference's replay list holds FILE NAMES and re-reads them (train_nele.py:373-403); here it holds the D inputs themselves
        ([64, T, 4] float32, ~0.2 MB per item), which grows by an epoch's worth of samples per epoch.  Beyond ``history_hbm_bytes`` of
        device memory the items that follow in list order - the list is shuffled before every replay draw (d_epoch), so these are a random
        subset, the newest epoch's items among them - move to page-locked host memory: ONE slab per call (a page-locked allocation per item
        would cost more than the epoch), asynchronous copies on the current stream; a replayed item is uploaded again."""
        if self.device.type != 'cuda':
            return
        used, spill = 0, []
        for k, it in enumerate(self.history):
            if it[0].device.type != 'cuda':
                continue
            used += it[0].numel() * it[0].element_size()
            if used > self.history_hbm_bytes:
                spill.append(k)
        if not spill:
            return
        al = lambda n: (n + 15) // 16 * 16
        total = sum(al(t.numel() * t.element_size()) for k in spill for t in self.history[k] if t is not None)
        slab = torch.empty(max(16, total), dtype=torch.uint8, pin_memory=True)
        off = 0
        for k in spill:
            out = []
            for t in self.history[k]:
                if t is None:
                    out.append(None)
                    continue
                nb = t.numel() * t.element_size()
                h = slab[off:off + nb].view(t.dtype).view(t.shape)     # (views keep the slab alive)
                h.copy_(t, non_blocking=True)                           # ordered on the current stream; read again only through _on_device (same stream)
                out.append(h)
                off += al(nb)
            self.history[k] = tuple(out)

    # ---------------------------------------------------------------- one GAN epoch (train_nele.py:110-429)
    def d_mse(self, samples, batch=32):
        """Mean squared error of D's predictions over a list of D training items (din, target[, target_qua]) - evaluation mode, no
        gradient, no optimiser step, the spectral-norm iteration does not advance.  The health signal of the D half of the loop
        (train_nele.py:356-367 prints its loss every 1000 steps): before ``d_epoch`` it says how well D predicts the true metric
        scores of samples it has not seen, afterwards how well it fits them.  This rank's items only."""
        if not samples:
            return None
        self._flush_d()
        was = self.D.training
        self.D.eval()
        tot = torch.zeros((), dtype=torch.float64, device=self.device)
        cnt = 0
        try:
            with torch.no_grad():
                for din, tgt, _tq, frames, items in self._padded_chunks(list(samples), batch):
                    score = self.D.forward_packed(din, frames)[:items]
                    tot = tot + ((score.double() - tgt[:items].double()) ** 2).sum()
                    cnt += items * score.shape[1]
        finally:
            self.D.train(was)
        return float(tot) / max(1, cnt)

    def _quality_async(self, clean, deg, n_samples, rows=None, mapped=True):
        """PESQ / ViSQOL of the rows of a batch on a background thread (both are host programs: quality.Scorer): the two batches leave
        through pinned buffers behind everything enqueued so far, the caller's stream is not waited for.  ``n_samples`` [B] (host): the
        samples of row k the reference compares (the enhanced FILE's length - 256 * (L // 256) for a generated example, audio_util.py:216-218).
        ``rows``: score only these (the others stay 0).  -> Future of float32 [B, 2]: D_Qua's targets (``mapped``) or the raw scores."""
        import concurrent.futures as cf
        from . import dataio
        B = clean.shape[0]
        sel = list(range(B)) if rows is None else [int(r) for r in rows]
        ns = [int(min(int(n_samples[k]), clean.shape[1], deg.shape[1])) for k in range(B)]
        hc = dataio.pinned_get(tuple(clean.shape), torch.float32)
        hd = dataio.pinned_get(tuple(deg.shape), torch.float32)
        hc.copy_(clean.detach(), non_blocking=True)
        hd.copy_(deg.detach(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        scorer = self.quality_scorer

        def job():
            ev.synchronize()
            try:
                refs = [hc[k, :ns[k]].numpy().copy() for k in sel]
                degs = [hd[k, :ns[k]].numpy().copy() for k in sel]
            finally:
                dataio.pinned_put(hc)
                dataio.pinned_put(hd)
            raw = scorer.raw(refs, degs, fs)
            out = np.zeros((B, 2), dtype=np.float32)
            out[sel] = scorer.map(raw) if mapped else raw
            return out
        if self._quality_pool is None:
            self._quality_pool = cf.ThreadPoolExecutor(max_workers=2)
        return self._quality_pool.submit(job)

    def run_epoch(self, gan_epoch, train_batches, valid_batches=(), chkpt_path=None, sample_dir=None, log_path=None, d_batch=32,
                  check=True, d_eval=False):
        """One iteration of ``for gan_epoch in np.arange(1, GAN_epoch+1)`` (train_nele.py:110) in the reference's order.

        train_batches / valid_batches: sequences of dicts {'clean': wav [B,L], 'noise': wav [B,L], optional 'names': [B] wave names,
        optional 'drc': pre-enhanced wav [B,L] (the MultiEnh example of the same utterance, train_nele.py:333-340), optional
        'qua': quality targets [B,2] / 'drc_qua' (mapped PESQ / ViSQOL of the generated / pre-enhanced example; only used with D_Qua -
        without them a trainer that has a ``quality_scorer`` scores the examples itself, as train_nele.py:323-324, 336-337 do),
        optional 'lengths': [B] samples of each utterance inside the zero-padded batch (files of different lengths side by side, as
        the reference's batch-1 loop handles them one at a time) and 'drc_lengths' (of the pre-enhanced files; default 'lengths'),
        optional 'ids': [B] int64 utterance ids (haspi_dither='utterance': the dither rows follow the utterance, not the rank),
        optional 'keys' (dataio.FileBatches: the clean file's path + the folders it is paired with; default: 'names' - base names, which
        must then be unique over everything this trainer ever scores): one hashable per utterance - with enable_clean_cache() the clean-signal halves of SIIB / HASPI
        of an utterance are computed the first time it is scored and reused in every later epoch}.
        Under data parallelism every rank passes its own shard of batches; ranks may hold different batch counts and sizes (empty
        G-steps / D-steps join the collectives, gradients are item-weighted means, the validation means run over all ranks).
          1. G-steps over the training batches - from epoch 2 on (:122-156; epoch 1 fits D to the untrained generator first)
          2. validation: enhance, raw (unmapped) metrics, learning-curve line (:159-225)
          3. checkpoint (:272-277)
          4. generate the D training samples of the same training utterances (:279-316)
          5. true metric targets of the generated and of the pre-enhanced examples (:318-340)
          6. D (and D_Qua) training: three passes with 1/30 history replay (:342-426)
        ``d_eval``: also report D's mean squared error on the epoch's new samples before (``d_mse_fresh``) and after (``d_mse_fit``)
        its three training passes and the mean true targets of those samples (``target_mean``) - see tools/learn_curve.py.
        Returns a dict of what happened (losses, validation means, counts)."""
        out = {'gan_epoch': int(gan_epoch), 'g_steps': 0, 'g_loss': None, 'valid': None, 'samples': 0}
        feats = [None] * len(train_batches)
        dp = self.world > 1

        def fts(b):
            # with the clean-signal cache on, a batch's features (STFT / band energies of the clean file, IMCRA noise track of its noise
            # file: both files are fixed per utterance, train_nele.py:119, dataloader.py:30-42) are kept too, keyed by the batch's keys
            kw_ = ckw(b)
            fc = getattr(self, '_feat_cache', None)
            k_ = None
            if kw_ and kw_['keys'] is not None and fc is not None:
                k_ = (tuple(kw_['keys']), tuple(b['clean'].shape))
                hit = fc.get(k_)
                if hit is not None:
                    fc.move_to_end(k_)
                    return hit[0]
            f_ = self.features(b['clean'], b['noise'], b.get('lengths'))
            if k_ is not None:
                # least-recently-used, at most an eighth of the cache's budget: a loop that re-draws its batches every epoch (fit: the
                # reference shuffles its file list, train_nele.py:118) never sees a batch twice and must not crowd the per-utterance states out
                nb_ = sum(t.numel() * t.element_size() for t in f_.values() if torch.is_tensor(t))
                cap_ = self.clean_cache.budget // 8
                seen_ = self._feat_seen
                first_ = k_ not in seen_                                # admitted on its SECOND sighting: a composition that never comes back
                seen_[k_] = True                                        # (re-drawn batches) is not kept at all
                seen_.move_to_end(k_)
                while len(seen_) > 4096:
                    seen_.popitem(last=False)
                if nb_ <= cap_ and not first_:
                    while fc and self._feat_bytes + nb_ > cap_:
                        _, old_ = fc.popitem(last=False)
                        self._feat_bytes -= old_[1]
                    fc[k_] = (f_, nb_)
                    self._feat_bytes += nb_
            return f_

        def hkw(b):                                                     # host copies of the lengths, when the loader has them (test doubles take none)
            return ({'lengths_host': b['lengths_host'], 'drc_lengths_host': b.get('drc_lengths_host')}
                    if (b.get('lengths_host') is not None and (b.get('drc_lengths_host') is not None or b.get('drc_lengths') is None)) else {})

        def ckw(b):                                                     # utterance keys travel only when the clean-signal cache is on
            return {'keys': b.get('keys', b.get('names'))} if getattr(self, 'clean_cache', None) is not None else {}

        if gan_epoch >= 2:                                              # :122
            tot = None
            n_g = len(train_batches)
            if dp:                                                      # ranks may hold different batch counts: same number of all-reduces
                n_g = ndist.allreduce_max_int(n_g, self.device)
            for i in range(n_g):
                if i < len(train_batches):
                    b = train_batches[i]
                    f = feats[i] = fts(b)
                    lg = self.g_step(f['clean_band'], f['noise_band'], f.get('frames'), weight=b['clean'].shape[0] if dp else None)
                    tot = lg if tot is None else tot + lg
                else:
                    self.g_step(None, None, weight=0)
                out['g_steps'] += 1
            out['g_loss'] = tot / max(1, len(train_batches)) if tot is not None else None
        raw, vq = [], []
        scorer = getattr(self, 'quality_scorer', None)

        def nsamp(b, wav, resynth=True, other=None):
            # samples of every row the quality programs compare: the file the reference would have written (256 * (L // 256) samples of a
            # generated example), or min(clean, pre-enhanced) for a pre-enhanced one
            lh = b.get('lengths_host')
            if lh is None:
                lh = b['lengths'].tolist() if b.get('lengths') is not None else [b['clean'].shape[1]] * b['clean'].shape[0]
            ns_ = [int(v) for v in lh]
            if other is not None:
                ns_ = [min(a_, int(o_)) for a_, o_ in zip(ns_, other)]
            if resynth:
                ns_ = [256 * (v // 256) for v in ns_]
            return [min(v, wav.shape[1]) for v in ns_]

        for b in valid_batches:                                         # :159-222
            f = fts(b)
            enh = self.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=f.get('frames'))
            if scorer is not None:                                      # :216-222 (norm=False: the raw scores of the learning curve)
                vq.append(self._quality_async(b['clean'], enh, nsamp(b, enh), mapped=False))
            if sample_dir is not None and 'names' in b:                 # :190-198 (the reference keeps the first 20 for listening)
                self.write_samples(enh, b['names'], sample_dir + '/Test_epoch' + str(gan_epoch), gan_epoch,
                                   lengths=b.get('lengths_host', b.get('lengths')), wait=False)
            raw.append(self.true_metrics(b['clean'], enh, b['noise'], norm=False, lengths=b.get('lengths'), utt_ids=b.get('ids'), defer=True, **ckw(b)))
        raw = [p.result() for p in raw]                                 # (a batch's metrics ran under the next batch's generator)
        if raw or (dp and valid_batches is not None):
            n_m = len(self.metrics)
            acc = torch.zeros(2 * n_m + 1 + 3, dtype=torch.float64, device=self.device)
            if vq:                                                      # [.., sum PESQ, sum ViSQOL, utterances scored]
                q_ = np.concatenate([p.result() for p in vq], axis=0).astype(np.float64)
                acc[2 * n_m + 1:] = torch.tensor([q_[:, 0].sum(), q_[:, 1].sum(), float(q_.shape[0])], dtype=torch.float64)
            if raw:
                r = torch.cat(raw, dim=0).double()
                acc[:n_m] = r.sum(dim=0)
                acc[n_m] = r.shape[0]
                # the same scores through the reference's logistic maps (intel.py:52-55,102-106,116-118): what D is trained to predict
                # and G is trained to push to 1 - not part of the reference's log line, reported beside it (out['valid_mapped'])
                for i, m in enumerate(self.metrics):
                    k_, x0_ = _MAPS[m]
                    acc[n_m + 1 + i] = (1.0 / (1.0 + torch.exp(-k_ * (r[:, i] - x0_)))).sum()
            if dp:                                                      # the learning curve is the mean over ALL validation utterances
                import torch.distributed as tdist
                tdist.all_reduce(acc)
            if float(acc[n_m]) > 0:
                r = (acc[:n_m] / acc[n_m]).cpu().numpy()
                col = {m: float(r[i]) for i, m in enumerate(self.metrics)}
                out['valid'] = col
                rm = (acc[n_m + 1:2 * n_m + 1] / acc[n_m]).cpu().numpy()
                out['valid_mapped'] = {m: float(rm[i]) for i, m in enumerate(self.metrics)}
                if float(acc[2 * n_m + 3]) > 0:                         # Test_PESQ / Test_VISQOL (:217-222)
                    col['pesq'] = float(acc[2 * n_m + 1] / acc[2 * n_m + 3])
                    col['visqol'] = float(acc[2 * n_m + 2] / acc[2 * n_m + 3])
                line = self.validation_log_line(col.get('siib', 0.0), col.get('haspi', 0.0), col.get('estoi', 0.0), gan_epoch)
                if log_path is not None and ndist.rank() == 0:
                    with open(log_path, 'a') as fh:                     # :224-225
                        fh.write(line)
        if chkpt_path is not None and ndist.rank() == 0:
            self.save_checkpoint(chkpt_path)                            # :272-277
        samples, pending, resolved = [], [], 0
        out['sample_files'] = []

        def rq(q):                                                      # quality targets scored on the host meanwhile (_quality_async)
            return torch.from_numpy(q.result()).to(self.device) if hasattr(q, 'result') else q

        def resolve(item):
            pend, din, qua, frames, din_d, drc_qua, fh, drc_items, store_keys, part = item
            qua, drc_qua = rq(qua), rq(drc_qua)
            tgt = pend.result()
            if din_d is not None:
                tgt, tgt_d = tgt
            samples.extend(self._items(din, tgt, qua, frames, fh))
            if din_d is not None:
                items_d = self._items(din_d, tgt_d, drc_qua, frames, fh)
                samples.extend(items_d)
                if store_keys is not None:                              # first time these utterances' pre-enhanced examples are scored: keep them
                    for k_, it in zip(store_keys, items_d):
                        nb_ = it[0].numel() * 4 + 64
                        if k_ not in self._drc_cache and self.clean_cache.used + nb_ <= self.clean_cache.budget:
                            self._drc_cache[k_] = tuple(None if t is None else t.clone() for t in it)   # (own storage: not a view of the batch)
                            self.clean_cache.used += nb_
            elif drc_items is not None:
                if part is not None:                                    # the batch's new pre-enhanced examples, computed as a batch of their own
                    pend_sub, din_sub, miss, dkeys, frames_b, fh_b, qua_b = part
                    qua_b = drc_qua                                     # (resolved above when it was a future)
                    tgt_sub = pend_sub.result()
                    fr_sub = None if frames_b is None else ([fh_b[r_] for r_ in miss] if fh_b is not None else [int(v) for v in frames_b[miss].tolist()])
                    new_items = self._items(din_sub, tgt_sub, None if qua_b is None else qua_b[miss], None if fr_sub is None else frames_b, fr_sub)
                    for r_, it in zip(miss, new_items):
                        it = tuple(None if t is None else t.clone() for t in it)
                        drc_items[r_] = it
                        nb_ = it[0].numel() * 4 + 64
                        if self.clean_cache.used + nb_ <= self.clean_cache.budget:
                            self._drc_cache[dkeys[r_]] = it
                            self.clean_cache.used += nb_
                samples.extend(drc_items)
        for i, b in enumerate(train_batches):                           # :279-340
            f = feats[i] if feats[i] is not None else fts(b)
            lens, frames = b.get('lengths'), f.get('frames')
            enh = self.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=frames)
            if sample_dir is not None and 'names' in b:
                out['sample_files'] += self.write_samples(enh, b['names'], sample_dir + '/For_discriminator_training', gan_epoch,
                                                          lengths=b.get('lengths_host', lens), wait=False)
            dl = b.get('drc_lengths', lens)
            # The pre-enhanced ('DRC') example of an utterance (:318-340; audio_util.py:267-321) never changes: its files are fixed, so its D
            # input and its true targets are the same in every epoch the utterance is drawn.  With the clean-signal cache on they are kept
            # per utterance key after the first time (the reference recomputes read_batch_*_DRC every epoch); a batch whose utterances are
            # all known skips the DRC file's STFT and its three metric calls altogether.
            dkeys = ckw(b).get('keys') if b.get('drc') is not None else None
            dcache = getattr(self, '_drc_cache', None)
            drc_items, miss = None, None
            if dkeys is not None and dcache is not None:
                drc_items = [dcache.get(k_) for k_ in dkeys]
                miss = [r_ for r_, it in enumerate(drc_items) if it is None]
                if len(miss) == len(dkeys):
                    drc_items, miss = None, None                         # nothing known: the ordinary path for the whole batch
            pend_sub = din_sub = None
            if b.get('drc') is not None and drc_items is None:
                # generated + pre-enhanced example of the same utterances: one pass over the clean signal for both when they are compared over
                # the same samples (audio_util.py:134-137)
                pend = self.true_metrics_pair(b['clean'], enh, b['drc'], b['noise'], lengths=lens, drc_lengths=dl, utt_ids=b.get('ids'), defer=True,
                                              **ckw(b), **hkw(b))
            else:
                pend = self.true_metrics(b['clean'], enh, b['noise'], lengths=lens, utt_ids=b.get('ids'), defer=True, **ckw(b))
                if miss:
                    # some pre-enhanced examples of this batch are new (a loop that re-draws its batches every epoch): their targets and D
                    # inputs as a batch of their own - true_metrics on (clean, pre-enhanced + noise) over min(clean, pre-enhanced) samples is
                    # what true_metrics_pair computes for them (audio_util.py:267-321), and every kernel is batch-invariant
                    idx = torch.tensor(miss, dtype=torch.long).to(self.device, non_blocking=True)
                    sub = lambda t: None if t is None else t.index_select(0, idx)
                    Ld_ = min(b['clean'].shape[1], b['drc'].shape[1])
                    ml_d = None
                    if dl is not None or lens is not None:
                        full_ = lambda t, w: torch.full((w.shape[0],), w.shape[1], dtype=torch.int32, device=self.device) if t is None else au._i32(t, self.device)
                        ml_d = torch.clamp(torch.minimum(full_(dl, b['drc']), full_(lens, b['clean'])), max=Ld_)
                    pend_sub = self.true_metrics(sub(b['clean']), sub(b['drc']), sub(b['noise']), lengths=sub(ml_d), resynth=False,
                                                 utt_ids=sub(b.get('ids')), defer=True, keys=[('drc', dkeys[r_]) for r_ in miss])
                    din_sub = self.d_inputs(sub(b['drc']), sub(f['noise_band']), sub(f['clean_band']),
                                            sub(au._i32(dl, self.device)) if dl is not None else None, resynth=False)
            # the targets are not waited for here: this batch's metric kernels (three streams, SIIB's a latency chain through the
            # eigensolver) run under the next batch's generator and feature kernels; everything is resolved behind the loop
            din = self.d_inputs(enh, f['noise_band'], f['clean_band'], lens)
            din_d = None
            if b.get('drc') is not None and drc_items is None:
                din_d = self.d_inputs(b['drc'], f['noise_band'], f['clean_band'], au._i32(dl, self.device) if dl is not None else None, resynth=False)
            lh = b.get('lengths_host')
            fh_ = None if lh is None else [1 + int(v) // 256 for v in lh]
            qua, drc_qua = b.get('qua'), b.get('drc_qua')
            if scorer is not None and self.D_Qua is not None:          # :323-324, 336-337: PESQ / ViSQOL of the examples D_Qua is trained on
                if qua is None:
                    qua = self._quality_async(b['clean'], enh, nsamp(b, enh))
                if drc_qua is None and b.get('drc') is not None and (drc_items is None or miss):
                    dlh = b.get('drc_lengths_host')
                    if dlh is None:
                        dlh = dl.tolist() if torch.is_tensor(dl) else (dl if dl is not None else [b['drc'].shape[1]] * b['drc'].shape[0])
                    drc_qua = self._quality_async(b['clean'], b['drc'], nsamp(b, b['drc'], resynth=False, other=dlh),
                                                  rows=None if drc_items is None else miss)
            pending.append((pend, din, qua, frames, din_d, drc_qua, fh_,
                            drc_items, dkeys if (dcache is not None and drc_items is None) else None,
                            None if not miss else (pend_sub, din_sub, miss, dkeys, frames, fh_, drc_qua)))
            # bounded lag: the targets of batch i - target_lag are resolved now, so that at most target_lag batches keep their metric inputs
            # (x, y, lengths: 2 - 3 x B x L x 4 bytes each) alive and the main stream cannot run arbitrarily far ahead of the metric streams
            while len(pending) - resolved > getattr(self, 'target_lag', 3):
                resolve(pending[resolved])
                resolved += 1
        while resolved < len(pending):
            resolve(pending[resolved])
            resolved += 1
        out['samples'] = len(samples)
        d0 = self.step_d
        if d_eval:                                                      # D's error on this epoch's samples before it has trained on them
            out['d_mse_fresh'] = self.d_mse(samples, d_batch)
            out['target_mean'] = torch.stack([s_[1] for s_ in samples]).double().mean(dim=0).tolist() if samples else None
        self.d_epoch(samples, batch=d_batch)                            # :342-426
        out['d_steps'] = self.step_d - d0
        if d_eval:
            out['d_mse_fit'] = self.d_mse(samples, d_batch)
        self.flush_writes()                                             # the epoch's sample files are on disk when it returns
        if check:
            out['status'] = self.check_status()
        return out

    # ---------------------------------------------------------------- the script's outer loop (train_nele.py:28-120, 272-277, 426-428)
    def fit(self, train_clean_files, train_noise_path, valid_clean_files=(), valid_noise_path=None, train_enh_path=None, epochs=GAN_epoch,
            sampling=num_of_sampling, valid_samples=num_of_valid_sample, batch=32, output_path='./output', pt_dir='./chkpt', log_path='./log.txt',
            workers=8, first_epoch=1, clean_cache=False, on_epoch=None):
        """``for gan_epoch in np.arange(1, GAN_epoch+1)`` of the reference script over folders of wav files, in its order:
        every epoch the training list is shuffled and its first ``sampling`` files are drawn (train_nele.py:118-119), `run_epoch` runs on them
        (G-steps from epoch 2, validation on the first ``valid_samples`` validation files with the learning-curve line appended to
        ``log_path``, checkpoint ``pt_dir/chkpt_<epoch>.pt``, generated samples as ``<output_path>/For_discriminator_training/<name>@<epoch>.wav``,
        true targets of the generated and - ``train_enh_path``: the folder of pre-enhanced 'MultiEnh' examples, train_nele.py:52 - the
        pre-enhanced examples, three D passes with 1/30 replay).  Files are read in batches of ``batch`` through dataio.FileBatches (the
        reference: batch 1); ``clean_cache``: keep the clean-signal halves of the metrics across epochs (enable_clean_cache).
        ``on_epoch(result_dict)``: called after every epoch (return True to stop).  -> list of the epochs' result dicts.
        The shuffles use Python's global generator, seeded by the constructor like train_nele.py:28."""
        from . import dataio
        dataio.creatdir(pt_dir)                                         # :45-46
        dataio.creatdir(output_path)
        if clean_cache and self.clean_cache is None:
            self.enable_clean_cache()
        train_files = list(train_clean_files)
        random.shuffle(train_files)                                     # :57
        valid_files = list(valid_clean_files)
        random.shuffle(valid_files)                                     # :68
        valid_files = valid_files[:int(valid_samples)]
        if self.world > 1:                                              # data parallelism: every rank scores its contiguous shard of the validation
            lo_, hi_ = ndist.shard_range(len(valid_files))              # list (run_epoch all-reduces the means) ...
            valid_files = valid_files[lo_:hi_]
        vb = None
        if valid_files:
            vb = dataio.FileBatches(valid_files, valid_noise_path, batch=batch, workers=workers, ahead=2, keep=2, device=self.device)
        results = []
        try:
            for gan_epoch in range(int(first_epoch), int(epochs) + 1):
                random.shuffle(train_files)                             # :118
                sel = train_files[0:int(round(sampling))]
                if self.world > 1:                                      # ... and trains on its shard of the epoch's draw (the same shuffle on every
                    lo_, hi_ = ndist.shard_range(len(sel))              # rank: the constructor seeds the generator; gradients are item-weighted means)
                    sel = sel[lo_:hi_]
                fb = dataio.FileBatches(sel, train_noise_path, batch=batch, drc_path=train_enh_path, workers=workers, ahead=2, keep=2, device=self.device)
                try:
                    res = self.run_epoch(gan_epoch, fb, vb if vb is not None else (), chkpt_path=os.path.join(pt_dir, 'chkpt_%d.pt' % gan_epoch),
                                         sample_dir=output_path, log_path=log_path, d_batch=batch)
                finally:
                    fb.close()
                results.append(res)
                if on_epoch is not None and on_epoch(res):
                    break
        finally:
            if vb is not None:
                vb.close()
        return results

    @staticmethod
    def _items(din, tgt, qua, frames, frames_host=None):
        """A batch's D training items: (din [64, T_k, 4] cut to the utterance's own frames, target [n], quality target [2] | None).
        The items are VIEWS of the batch's rows (the gather kernel of _padded_chunks takes band rows of any stride); ``frames_host``:
        the frame counts on the host when the loader knows them (dataio.FileBatches: 'lengths_host') - reading them back from the device
        stalls the thread that enqueues the GPU work until everything before it has run."""
        if frames is None:
            fr = None
        elif frames_host is not None:
            fr = [int(v) for v in frames_host]
        else:
            fr = [int(v) for v in frames.tolist()]
        T = din.shape[2]
        return [(din[k] if (fr is None or fr[k] == T) else din[k, :, :fr[k]], tgt[k], qua[k] if qua is not None else None)
                for k in range(din.shape[0])]

    # ---------------------------------------------------------------- file hand-off (train_nele.py:303-340, 224-225)
    def write_samples(self, enh_wav, wave_names, directory, gan_epoch, lengths=None, wait=True):
        """Enhanced batch -> '<directory>/<name>@<epoch>.wav' PCM_16 files, as the reference stores its generated D samples
        (train_nele.py:309-313).  ``enh_wav`` is what ``generate`` returned (already PCM_16-quantised when ``self.pcm16``);
        lengths [B] (host list / array, or a tensor): samples of the ORIGINAL utterances of a padded batch (file k then holds
        256 * (lengths[k] // 256) samples).

        PCM_16 batches on the GPU leave as int16 sample values (nele_float_to_pcm16) through a pinned buffer and are written by the
        library's own threads, one call per batch (nele_wav_write_pcm16_batch); ``wait=False`` hands that to a background thread - the
        caller's stream is not synchronised - and ``flush_writes()`` waits for the files (run_epoch does before it returns)."""
        from . import dataio
        dataio.creatdir(directory)
        if lengths is None:
            lens = None
        elif torch.is_tensor(lengths):
            lens = [256 * (int(v) // 256) for v in lengths.tolist()]
        else:
            lens = [256 * (int(v) // 256) for v in lengths]
        out = [dataio.enhanced_name(directory, name, gan_epoch) for name in wave_names]
        if self.pcm16 and enh_wav.is_cuda and enh_wav.dim() == 2 and enh_wav.is_contiguous() and enh_wav.dtype == torch.float32:
            from . import _lib
            B, L = enh_wav.shape
            q = torch.empty((B, L), dtype=torch.int16, device=enh_wav.device)
            _lib.check(_lib.lib.nele_float_to_pcm16(enh_wav.data_ptr(), L, B, L, q.data_ptr(), L, 1, _lib.stream()), 'nele_float_to_pcm16')
            host = dataio.pinned_get((B, L), torch.int16)
            host.copy_(q, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ns = lens if lens is not None else [L] * B

            def job():
                ev.synchronize()
                try:
                    dataio.write_wav_batch_pcm16(out, host.numpy(), ns, fs, threads=8)
                finally:
                    dataio.pinned_put(host)
            if wait:
                job()
            else:
                if getattr(self, '_write_pool', None) is None:
                    import concurrent.futures as cf
                    self._write_pool, self._writes = cf.ThreadPoolExecutor(max_workers=1), []
                self._writes.append(self._write_pool.submit(job))
            return out
        host = enh_wav.detach().cpu().numpy()
        for k, (w, path) in enumerate(zip(host, out)):
            dataio.write_wav_pcm16(path, w if lens is None else w[:lens[k]], fs, quantised=self.pcm16)
        return out

    def flush_writes(self):
        """Wait for the sample files handed to the background writer (write_samples(wait=False)); re-raises a writer's error."""
        pend, self._writes = getattr(self, '_writes', None) or [], []
        for f in pend:
            f.result()

    def score_lines(self, targets, enhanced_names):
        """[B, n_metrics] targets -> 's_siib,s_haspi,s_estoi,s_pesq,s_visqol,path' items of the reference's D training list
        (train_nele.py:334-340: unused metrics are zero)."""
        from . import dataio
        t = targets.detach().double().cpu().numpy()
        col = {m: t[:, i] for i, m in enumerate(self.metrics)}
        zero = np.zeros(len(t))
        five = [col.get('siib', zero), col.get('haspi', zero), col.get('estoi', zero), zero, zero]
        return dataio.List_concat(dataio.List_concat_5scores(*[list(map(float, c)) for c in five]), enhanced_names)

    @staticmethod
    def validation_log_line(siib, haspi, estoi, gan_epoch):
        """The learning-curve line of train_nele.py:224-225 (PESQ / ViSQOL are reported as 0 there too)."""
        return 'SIIB is %.3f, HASPI is %.3f, ESTOI is %.3f, PESQ is %.3f, VISQOL is %.3f, EPOCH:%d \n' % (
            float(np.mean(siib)), float(np.mean(haspi)), float(np.mean(estoi)), 0, 0, gan_epoch)

    # ---------------------------------------------------------------- checkpoints (train_nele.py:272-277)
    def save_checkpoint(self, path):
        self._flush_d()
        sd = {'enhance-model': self.G.state_dict(), 'intel-model': self.D.state_dict()}
        if self.D_Qua is not None:
            sd['quality-model'] = self.D_Qua.state_dict()
        torch.save(sd, path)

    def flush(self):
        """Complete a deferred D update (overlap_allreduce: canonical_step leaves D's gradient all-reduce and Adam-D step pending until
        something touches D).  Call before reading ``self.D``'s weights or ``step_d`` directly after canonical_step."""
        self._flush_d()

    def load_checkpoint(self, path):
        self._flush_d()                 # a pending pre-load gradient must not be applied to the loaded weights
        ck = torch.load(path, map_location=self.device)
        self.G.load_state_dict(ck['enhance-model'])
        if 'intel-model' in ck:
            self.D.load_state_dict(ck['intel-model'])
        if self.D_Qua is not None and 'quality-model' in ck:
            self.D_Qua.load_state_dict(ck['quality-model'])


def main(argv=None):
    """``python -m nele_gan_amd.train_nele --data ./toy_dataset`` = the reference's usage step 3 (`python train_nele.py`, README) with the
    script's constants (train_nele.py:28-43, 49-70) as options instead of edits: <data>/Train/{Clean,Noise,MultiEnh}, <data>/Test/{Clean,Noise}
    as in ./toy_dataset; checkpoints in --chkpt, samples and the learning-curve log under --output.  Under torchrun every rank takes its
    shard of the epoch's draw (one process per GPU, RCCL)."""
    import argparse
    from . import dataio
    ap = argparse.ArgumentParser(prog='python -m nele_gan_amd.train_nele', description=main.__doc__)
    ap.add_argument('--data', required=True, help='corpus root with Train/Clean, Train/Noise, Train/MultiEnh (optional), Test/Clean, Test/Noise')
    ap.add_argument('--metrics', default=TargetMetric)
    ap.add_argument('--epochs', type=int, default=GAN_epoch)
    ap.add_argument('--sampling', type=int, default=num_of_sampling)
    ap.add_argument('--valid-samples', type=int, default=num_of_valid_sample)
    ap.add_argument('--batch', type=int, default=32, help='utterances per batch (the reference: 1)')
    ap.add_argument('--output', default='./output')
    ap.add_argument('--chkpt', default='./chkpt')
    ap.add_argument('--log', default='./log.txt')
    ap.add_argument('--precision', choices=('bf16', 'f32'), default='bf16', help='MFMA operand precision of G and D (f32 = the reference\'s arithmetic)')
    ap.add_argument('--valid-filter', default=None, help='keep validation files whose name contains this (train_nele.py:65: \'AIR_stairway\')')
    ap.add_argument('--resume', default=None, help='checkpoint to start from (train_nele.py:82-84)')
    ap.add_argument('--first-epoch', type=int, default=1)
    ap.add_argument('--no-cache', action='store_true', help='recompute the clean-signal halves of the metrics every epoch')
    ap.add_argument('--quality', action='store_true', help='train Discriminator_Quality too (train_nele.py:150-152, 362-365): needs --pesq and --visqol - external programs')
    from . import quality as _q
    _q.add_cli_arguments(ap)
    a = ap.parse_args(argv)
    if a.quality and not _q.backends_from_cli(a):
        ap.error('--quality needs both programs: --pesq MODULE:FUNCTION and --visqol PROGRAM --visqol-model FILE (neither is part of this build)')
    root = a.data.rstrip('/')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as tdist
        torch.cuda.set_device(local)
        tdist.init_process_group('nccl', device_id=torch.device('cuda', local))
    scorer = None
    if a.quality:
        from . import quality
        scorer = quality.Scorer()
    tr = GanTrainer(a.metrics, device='cuda:%d' % local, use_quality=a.quality, quality_scorer=scorer)
    for m_ in (tr.G, tr.D, tr.D_Qua):
        if m_ is not None:
            m_.precision = a.precision
    if a.resume:
        tr.load_checkpoint(a.resume)
    train = sorted(dataio.get_filepaths(root + '/Train/Clean/'))
    valid = sorted(dataio.get_filepaths(root + '/Test/Clean/'))
    if a.valid_filter:
        valid = [x for x in valid if a.valid_filter in x]
    enh = root + '/Train/MultiEnh/'
    if ndist.rank() == 0:
        print('%d training files, %d validation files, %s' % (len(train), len(valid), 'pre-enhanced examples: ' + enh if os.path.isdir(enh) else 'no pre-enhanced examples'))

    def report(res):
        if ndist.rank() == 0:
            v = res.get('valid') or {}
            print('epoch %d: g_loss %s, %d samples, %d D steps, valid %s' % (res['gan_epoch'], res['g_loss'] if res['g_loss'] is None else '%.4f' % float(res['g_loss']),
                                                                             res['samples'], res['d_steps'], ', '.join('%s %.3f' % kv for kv in v.items())), flush=True)
    try:
        tr.fit(train, root + '/Train/Noise/', valid, root + '/Test/Noise/', train_enh_path=enh if os.path.isdir(enh) else None, epochs=a.epochs,
               sampling=a.sampling, valid_samples=a.valid_samples, batch=a.batch, output_path=a.output, pt_dir=a.chkpt, log_path=a.log,
               first_epoch=a.first_epoch, clean_cache=not a.no_cache, on_epoch=report)
    finally:
        if world > 1:
            import torch.distributed as tdist
            tdist.destroy_process_group()
    return 0


if __name__ == '__main__':
    import sys
    sys.exit(main())
