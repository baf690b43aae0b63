"""Oracle: the canonical GAN_epoch step on the CPU: features -> G-step -> generate ->
true metrics -> D-step, batch 1 semantics per utterance as in the reference, restated with the
oracle modules (numpy / scipy / torch-CPU float32).  TEST INFRASTRUCTURE ONLY: used by the tests
as the checker and by bench.py's ``cpu_baseline`` leg as the timed CPU port."""
import time

import numpy as np
import torch

from . import estoi as o_estoi
from . import features as F
from . import intel as o_intel
from . import nets
from . import siib as o_siib

P_POWER, INV_P = 1 / 6, 6


def pcm16_roundtrip(x):
    """sf.write(..., 'PCM_16') + librosa.load: libsndfile scales by 0x7FFF and rounds to nearest even,
    reading divides by 0x8000 (PARITY UNPINNED: libsndfile not present)."""
    q = np.clip(np.rint(np.asarray(x, dtype=np.float32) * np.float32(32767.0)), -32768, 32767)
    return (q / 32768.0).astype(np.float32)


def metric_targets(clean, enh, noise, metrics, norm=True):
    L = min(len(clean), len(enh))
    x = clean[:L]
    y = enh[:L] + noise[:L]
    out = []
    for m in metrics:
        if m == 'siib':
            out.append(o_siib.siib_wrapper(x, y, norm=norm))
        elif m == 'estoi':
            out.append(o_estoi.estoi_wrapper(x, y, norm=norm))
        elif m == 'haspi':
            from . import haspi as o_haspi
            out.append(o_haspi.haspi_wrapper(x, y, norm=norm))
        else:
            raise ValueError(m)
    return np.asarray(out, dtype=np.float32)


def _one_metric(clean, enh, noise, metric, norm=True):
    return float(metric_targets(clean, enh, noise, (metric,), norm)[0])


def _one_feature(c, v):
    b, m, p = F.sp_and_phase_speech(c, P_POWER)
    return b, m, p, F.sp_and_phase_noise(v, P_POWER)[0]


def _noop(i):
    return i


class CpuStep:
    """State (weights, Adam moments) + one canonical step over a batch."""

    def __init__(self, g_state, d_state, metrics=('siib', 'estoi'), lr_g=5e-4, lr_d=2.5e-4, dq_state=None, weight_qua=0.5):
        """dq_state: state_dict of Discriminator_Quality - the G-step then carries the 0.5 * MSE(D_Qua) term (train_nele.py:150-152)
        and d_step trains D_Qua on [enhanced, clean] against quality targets with its own Adam (train_nele.py:362-365)."""
        self.metrics = tuple(metrics)
        self.weight_qua = weight_qua
        self.dq = None
        if dq_state is not None:
            self.dq = {k: v.detach().clone().float() for k, v in dq_state.items()}
            for k, v in self.dq.items():
                if not (k.endswith('_u') or k.endswith('_v')):
                    v.requires_grad_(True)
            self.opt_dq = torch.optim.Adam([v for v in self.dq.values() if v.requires_grad], lr=lr_d)
        self.g = {k: v.detach().clone().float().requires_grad_(True) for k, v in g_state.items()}
        self.d = {k: v.detach().clone().float() for k, v in (d_state or {}).items()}      # d_state=None: inference only (enhance)
        for k, v in self.d.items():
            if not (k.endswith('_u') or k.endswith('_v')):
                v.requires_grad_(True)
        self.opt_g = torch.optim.Adam(list(self.g.values()), lr=lr_g)
        self.opt_d = torch.optim.Adam([v for v in self.d.values() if v.requires_grad], lr=lr_d) if self.d else None
        self.times = {}

    def _t(self, key, t0):
        self.times[key] = self.times.get(key, 0.0) + time.perf_counter() - t0

    def features(self, clean, noise, workers=1):
        """workers > 1: one process per utterance, as the reference's DataLoader(num_workers=8) (dataloader.py:91)."""
        t0 = time.perf_counter()
        if workers > 1:
            from joblib import Parallel, delayed
            res = Parallel(n_jobs=workers)(delayed(_one_feature)(c, v) for c, v in zip(clean, noise))
        else:
            res = [_one_feature(c, v) for c, v in zip(clean, noise)]
        cb, cm, cp, nb = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res], [r[3] for r in res]
        self._t('features', t0)
        return np.stack(cb), cm, cp, np.stack(nb)

    def _d_forward(self, x, train=True):
        s, nb = nets.discriminator_forward(self.d, x, train=train)
        for k, v in nb.items():
            self.d[k] = v.detach()
        return s

    def _dq_forward(self, x2):
        s, nb = nets.discriminator_forward(self.dq, x2, train=True)
        for k, v in nb.items():
            self.dq[k] = v.detach()
        return s

    def g_step(self, cb, nb):
        t0 = time.perf_counter()
        cbt, nbt = torch.from_numpy(cb), torch.from_numpy(nb)
        mask = nets.generator_forward(self.g, cbt, nbt)
        enh, _ = nets.energy_norm(mask, cbt, P_POWER, INV_P)
        x = nets.d_inputs(enh, nbt, cbt)
        score = self._d_forward(x)
        loss = torch.nn.functional.mse_loss(score, torch.ones_like(score))
        if self.dq is not None:                                    # train_nele.py:146-152: d_inputs_qua = cat(enh, ref)
            score_q = self._dq_forward(x[:, [0, 2]])
            loss = loss + self.weight_qua * torch.nn.functional.mse_loss(score_q, torch.ones_like(score_q))
            self.opt_dq.zero_grad()
        self.opt_g.zero_grad()
        self.opt_d.zero_grad()
        loss.backward()
        self.opt_g.step()
        self._t('g_step', t0)
        return float(loss.detach())

    def generate(self, cb, nb, cm, cp, pcm16=True):
        t0 = time.perf_counter()
        with torch.no_grad():
            cbt = torch.from_numpy(cb)
            mask = nets.generator_forward(self.g, cbt, torch.from_numpy(nb))
            cpow = torch.pow(cbt, INV_P)
            beta2 = cpow.sum(dim=(1, 2), keepdim=True) / (mask * cpow).sum(dim=(1, 2), keepdim=True)
            alpha2 = (mask * beta2).numpy()
        out = []
        for b in range(cb.shape[0]):
            w = F.sp_to_wav(alpha2[b], cm[b], cp[b])
            out.append(pcm16_roundtrip(w) if pcm16 else w)
        self._t('generate', t0)
        return out

    def enhance(self, clean, noise, rms_target=0.030, pcm16=True):
        """inference.py:79-115 per utterance: features -> G (eval) -> mask * beta_2 -> SP_to_wav -> enh / rms(enh) * 0.03 -> PCM_16.
        -> list of float32 arrays of 256 * (len // 256) samples."""
        out = []
        with torch.no_grad():
            for c, v in zip(clean, noise):
                cb, cm, cp, nb = _one_feature(c, v)
                cbt = torch.from_numpy(cb[None])
                mask = nets.generator_forward(self.g, cbt, torch.from_numpy(nb[None]))
                cpow = torch.pow(cbt, INV_P)
                beta2 = cpow.sum() / (mask * cpow).sum()
                w = F.sp_to_wav((mask * beta2)[0].numpy(), cm, cp)
                w = (w / np.sqrt(np.mean(w ** 2)) * rms_target).astype(np.float32)          # inference.py:109, audio_util.py:463-464
                out.append(pcm16_roundtrip(w) if pcm16 else w)
        return out

    def targets(self, clean, enh, noise, workers=1):
        """workers > 1: the reference's joblib fan-out, one job per (utterance, metric) (audio_util.py:146, 174, 202)."""
        t0 = time.perf_counter()
        if workers > 1:
            from joblib import Parallel, delayed
            jobs = [(i, m) for m in self.metrics for i in range(len(clean))]
            vals = Parallel(n_jobs=workers)(delayed(_one_metric)(clean[i], enh[i], noise[i], m) for i, m in jobs)
            t = np.zeros((len(clean), len(self.metrics)), dtype=np.float32)
            for (i, m), x in zip(jobs, vals):
                t[i, self.metrics.index(m)] = x
        else:
            t = np.stack([metric_targets(c, e, v, self.metrics) for c, e, v in zip(clean, enh, noise)])
        self._t('metrics', t0)
        return t

    def epoch_slice(self, clean, noise, feature_workers=8, metric_workers=32):
        """The reference's work for these utterances in its own order and batching (train_nele.py:119-156, 279-367): features by the
        loader workers, one G-step per utterance (batch 1), generate, metric fan-out over processes, one D-step per utterance."""
        cb, cm, cp, nb = self.features(clean, noise, workers=feature_workers)
        for i in range(len(clean)):
            self.g_step(cb[i:i + 1], nb[i:i + 1])
        enh = self.generate(cb, nb, cm, cp)
        tgt = self.targets(clean, enh, noise, workers=metric_workers)
        for i in range(len(clean)):
            self.d_step(enh[i:i + 1], nb[i:i + 1], cb[i:i + 1], tgt[i:i + 1])
        return tgt

    def d_step(self, enh, nb, cb, tgt, tgt_qua=None):
        """train_nele.py:349-367; with ``tgt_qua`` [B,2] (and a D_Qua state) also the D_Qua step -> (loss, loss_qua)."""
        t0 = time.perf_counter()
        eb = np.stack([F.sp_and_phase_speech(e, P_POWER)[0] for e in enh])
        x = nets.d_inputs(torch.from_numpy(eb), torch.from_numpy(nb), torch.from_numpy(cb))
        score = self._d_forward(x)
        score_q = self._dq_forward(x[:, [0, 2]]) if tgt_qua is not None else None      # dataloader.py:83
        loss = torch.nn.functional.mse_loss(score, torch.from_numpy(tgt))
        self.opt_d.zero_grad()
        loss.backward()
        self.opt_d.step()
        if score_q is not None:
            loss_q = torch.nn.functional.mse_loss(score_q, torch.from_numpy(np.asarray(tgt_qua, dtype=np.float32)))
            self.opt_dq.zero_grad()
            loss_q.backward()
            self.opt_dq.step()
            self._t('d_step', t0)
            return float(loss.detach()), float(loss_q.detach())
        self._t('d_step', t0)
        return float(loss.detach())

    def canonical_step(self, clean, noise):
        cb, cm, cp, nb = self.features(clean, noise)
        lg = self.g_step(cb, nb)
        enh = self.generate(cb, nb, cm, cp)
        tgt = self.targets(clean, enh, noise)
        ld = self.d_step(enh, nb, cb, tgt)
        return lg, ld, tgt, enh
