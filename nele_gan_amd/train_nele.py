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


class GanTrainer:
    def __init__(self, target_metric=TargetMetric, device='cuda', lr_g=5e-4, lr_d=2.5e-4, use_quality=False, pcm16=True, seed=666):
        self.metrics = parse_metrics(target_metric)
        self.device = torch.device(device)
        torch.manual_seed(seed)                      # same initial weights on every rank
        random.seed(seed)                            # train_nele.py:28
        self.G = M.Generator_Conv1D_cLN().to(self.device)
        self.D = M.Discriminator(nout=len(self.metrics)).to(self.device)
        self.D_Qua = M.Discriminator_Quality().to(self.device) if use_quality else None
        self.optimizer_g = Adam(self.G, lr=lr_g)     # train_nele.py:89-91
        self.optimizer_d = Adam(self.D, lr=lr_d)
        self.optimizer_dqua = Adam(self.D_Qua, lr=lr_d) if use_quality else None
        self.MSELoss = nn.MSELoss()
        self.pcm16 = pcm16
        self.step_g = 0
        self.step_d = 0
        self.history = []                            # Previous_Discriminator_training_list (train_nele.py:373-403)
        self._side = None
        self._side2 = None
        self._fside = None
        self.world = ndist.world_size()
        for m in (self.G, self.D, self.D_Qua):
            if m is not None:
                ndist.broadcast_module_(m, 0)

    # ---------------------------------------------------------------- data-parallel glue
    def _allreduce_grads(self, module):
        if self.world > 1:
            ndist.allreduce_mean_(module.flat_parameters().grad)

    # ---------------------------------------------------------------- features (dataloader.py:30-42)
    def features(self, clean_wav, noise_wav):
        """wav [B,L] x2 -> dict(clean_band, noise_band [B,T,64], clean_spec [B,T,257] complex64).
        The noise branch (STFT -> IMCRA, a 1.1 ms scan that is serial over frames and fills 1/8 of the GPU) heads the step's critical
        path, so it is issued first; the clean STFT + band energies run beside it on their own stream."""
        main = torch.cuda.current_stream()
        if self._fside is None:
            self._fside = ops.side_stream(self.device)
        fs_ = self._fside
        ev0 = torch.cuda.Event()
        ev0.record(main)
        with torch.cuda.stream(fs_):
            fs_.wait_event(ev0)
            clean_spec, clean_band = au.stft_band(clean_wav, p_power)
            evc = torch.cuda.Event()
            evc.record(fs_)
        noise_spec, _ = au.stft_band(noise_wav, p_power, want_band=False)
        _, noise_band = au.imcra_band(noise_spec, p_power)
        main.wait_event(evc)
        clean_spec.record_stream(main)
        clean_band.record_stream(main)
        if clean_wav.is_cuda:
            clean_wav.record_stream(fs_)
        return {'clean_band': clean_band, 'noise_band': noise_band, 'clean_spec': clean_spec}

    # ---------------------------------------------------------------- G-step (train_nele.py:122-156)
    def g_step(self, clean_band, noise_band):
        B = clean_band.shape[0]
        self.D.weight_grad_enabled = False           # D / D_Qua gradients of this step are never applied (train_nele.py:153-155)
        if self.D_Qua is not None:
            self.D_Qua.weight_grad_enabled = False
        self.optimizer_g.zero_grad()
        mask = self.G(clean_band, noise_band)
        din, _ = M.energy_norm_pack(mask, clean_band, noise_band, p_power, inv_p)
        self._last_din = din                         # bench.py re-launches D's forward on it for the isolated roofline figure
        self.D.profile_prefix = 'gstep.'             # bench.py times these launches
        score = self.D.forward_packed(din)
        self.D.profile_prefix = ''
        loss = self.MSELoss(score, torch.ones_like(score))
        if self.D_Qua is not None:
            din_q = torch.zeros_like(din)
            din_q[..., 0] = din[..., 0]
            din_q[..., 1] = din[..., 2]
            score_q = self.D_Qua.forward_packed(din_q)
            loss = loss + weight_qua * self.MSELoss(score_q, torch.ones_like(score_q))
        loss.backward()
        self._allreduce_grads(self.G)
        self.optimizer_g.step()
        self.step_g += 1
        self.D.weight_grad_enabled = True
        if self.D_Qua is not None:
            self.D_Qua.weight_grad_enabled = True
        return loss.detach()

    # ---------------------------------------------------------------- sample generation (train_nele.py:279-316)
    @torch.no_grad()
    def generate(self, clean_band, noise_band, clean_spec, rms_target=0.0):
        self.G.eval()
        mask = self.G(clean_band, noise_band)
        alpha2 = M.normed_alpha2(mask, clean_band, inv_p)
        enh_wav = au.gain_istft(alpha2, clean_spec, rms_target=rms_target, pcm16=self.pcm16)
        self.G.train()
        return enh_wav

    # ---------------------------------------------------------------- true metric targets (train_nele.py:318-340)
    @torch.no_grad()
    def true_metrics(self, clean_wav, enh_wav, noise_wav, norm=True):
        L = min(clean_wav.shape[1], enh_wav.shape[1])          # audio_util.py:134-141
        x = clean_wav[:, :L].contiguous()
        y = (enh_wav[:, :L] + noise_wav[:, :L]).contiguous()
        cols = []
        for m in self.metrics:
            raw, mapped = getattr(mt, _METRIC_FN[m])(x, y)
            cols.append(mapped if norm else raw)
        return torch.stack(cols, dim=1)

    # ---------------------------------------------------------------- D-step (train_nele.py:349-367)
    def d_inputs(self, enh_wav, noise_band, clean_band):
        """dataloader.py:54-84: features of the enhanced wav, stacked (enhanced, noise, clean)."""
        _, enh_band = au.stft_band(enh_wav, p_power, want_spec=False)
        return ops.d_pack(enh_band, noise_band, clean_band)

    def d_step(self, din, target):
        self.optimizer_d.zero_grad()
        score = self.D.forward_packed(din)
        return self._d_finish(score, target)

    def _d_finish(self, score, target):
        loss = self.MSELoss(score, target)
        loss.backward()
        self._allreduce_grads(self.D)
        self.optimizer_d.step()
        self.step_d += 1
        return loss.detach()

    # ---------------------------------------------------------------- one canonical step
    def canonical_step(self, clean_wav, noise_wav, feats=None):
        """features -> G-step -> generate -> true metrics -> D-step on the same batch."""
        # The metric kernels run on a side stream.  (1) Everything SIIB derives from the CLEAN signal alone - VAD, clean spectra,
        # the covariance and its eigen-decomposition (the KLT basis) - is enqueued first and runs beside features / G-step /
        # generate.  (2) Once the enhanced signal exists the remaining metric work follows on the side stream while the main
        # stream runs D's forward pass, which does not need the targets; the loss waits for them.
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = ops.side_stream(self.device)
        side = self._side
        L = 256 * (clean_wav.shape[1] // 256)              # length of the resynthesised signal (audio_util.py:76-110)
        start = torch.cuda.Event()
        start.record(main)
        split = None
        with torch.cuda.stream(side):
            side.wait_event(start)
            x = clean_wav[:, :L].contiguous()
            if 'siib' in self.metrics:
                split = mt.SiibSplit(x)
                split.clean_part()
        if self._side2 is None:
            self._side2 = ops.side_stream(self.device)
        B_, T_ = clean_wav.shape[0], 1 + clean_wav.shape[1] // 256
        # D's spectral-norm iteration and weight layouts depend on its parameters only: they run on a side stream ahead of each of D's two
        # forward passes (beside the generator's forward pass / beside generate) instead of at the head of those passes
        with torch.cuda.stream(self._side2):
            self._side2.wait_event(start)                  # after the previous step's D update
            self.D.prepare(B_, T_, self.device)
        f = feats or self.features(clean_wav, noise_wav)
        lg = self.g_step(f['clean_band'], f['noise_band'])
        gdone = torch.cuda.Event()
        gdone.record(main)                                 # the G-step's backward pass is the last reader of D's current weight layouts
        with torch.cuda.stream(self._side2):
            self._side2.wait_event(gdone)
            self.D.prepare(B_, T_, self.device)
        enh = self.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
        assert enh.shape[1] == L
        ready = torch.cuda.Event()
        ready.record(main)
        if self._side2 is None:
            self._side2 = ops.side_stream(self.device)
        side2 = self._side2
        cols = {}
        with torch.cuda.stream(side):
            side.wait_event(ready)
            y = (enh + noise_wav[:, :L]).contiguous()
            y_ready = torch.cuda.Event()
            y_ready.record(side)
            if split is not None:
                cols['siib'] = split.degraded_part(y)[1]
        with torch.cuda.stream(side2):                     # the cheaper metrics beside SIIB's degraded-signal part
            side2.wait_event(start)
            side2.wait_event(y_ready)
            for m in self.metrics:
                if m != 'siib':
                    cols[m] = getattr(mt, _METRIC_FN[m])(x, y)[1]
            others = torch.cuda.Event()
            others.record(side2)
        with torch.cuda.stream(side):
            side.wait_event(others)
            for v in cols.values():
                v.record_stream(side)
            tgt = torch.stack([cols[m] for m in self.metrics], dim=1)
            done = torch.cuda.Event()
            done.record(side)
        for t in (x, y):
            t.record_stream(side2)
        din = self.d_inputs(enh, f['noise_band'], f['clean_band'])
        self.optimizer_d.zero_grad()
        score = self.D.forward_packed(din)
        for t in (tgt, x, y):
            t.record_stream(main)
        enh.record_stream(side)
        clean_wav.record_stream(side)
        noise_wav.record_stream(side)
        main.wait_event(done)
        ld = self._d_finish(score, tgt)
        return lg, ld, tgt

    # ---------------------------------------------------------------- D epoch: 3 passes + replay (train_nele.py:342-426)
    def d_epoch(self, samples, batch=32):
        """samples: list of (din [64,T,4], target [n]) tensors of this epoch (generated + pre-enhanced 'DRC' examples)."""
        def run(lst):
            random.shuffle(lst)
            for i in range(0, len(lst), batch):
                chunk = lst[i:i + batch]
                self.d_step(torch.stack([c[0] for c in chunk]), torch.stack([c[1] for c in chunk]))
        cur = list(samples)
        run(cur)                                                        # pass A
        random.shuffle(self.history)
        run(self.history[0:len(self.history) // 30] + cur)              # pass B: replay 1/30 of the history
        self.history = self.history + cur
        run(cur)                                                        # pass C

    # ---------------------------------------------------------------- file hand-off (train_nele.py:303-340, 224-225)
    def write_samples(self, enh_wav, wave_names, directory, gan_epoch):
        """Enhanced batch -> '<directory>/<name>@<epoch>.wav' PCM_16 files, as the reference stores its generated D samples
        (train_nele.py:309-313).  ``enh_wav`` is what ``generate`` returned (already PCM_16-quantised when ``self.pcm16``)."""
        from . import dataio
        dataio.creatdir(directory)
        host = enh_wav.detach().cpu().numpy()
        out = []
        for w, name in zip(host, wave_names):
            path = dataio.enhanced_name(directory, name, gan_epoch)
            dataio.write_wav_pcm16(path, w, fs, quantised=self.pcm16)
            out.append(path)
        return out

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
        sd = {'enhance-model': self.G.state_dict(), 'intel-model': self.D.state_dict()}
        if self.D_Qua is not None:
            sd['quality-model'] = self.D_Qua.state_dict()
        torch.save(sd, path)

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location=self.device)
        self.G.load_state_dict(ck['enhance-model'])
        if 'intel-model' in ck:
            self.D.load_state_dict(ck['intel-model'])
        if self.D_Qua is not None and 'quality-model' in ck:
            self.D_Qua.load_state_dict(ck['quality-model'])
