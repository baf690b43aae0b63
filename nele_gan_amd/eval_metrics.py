"""Validation conditions of the reference ``eval_metrics.py`` on the GPU, batched over utterances:
anechoic (``NO_rev``, eval_metrics.py:118-122) and reverberant (``MIRD_610`` / ``AIR_stairway21``, :124-165) listening
conditions, then the raw (un-mapped) SIIB / HASPI / ESTOI of (direct-path clean, reverberated enhanced + noise).

    lfilter_fir(h, x)        scipy.signal.lfilter(h, [1], x)           -> nele_fir_filter
    norm_clip(a, add, rms)   a / rms(a) * rms (+ add), audio_util.clip   -> nele_norm_clip
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import metrics as mt
from ._lib import c_int, c_void_p, call, declare, ptr, stream

_P = c_void_p
declare('nele_fir_filter', [_P, c_int, c_int, _P, c_int, _P, _P])
_lib._SIGS['nele_fir_filter'] = _lib.lib.nele_fir_filter.argtypes
declare('nele_norm_clip', [_P, _P, _P, c_int, c_int, ctypes.c_double, _P, _P, _P, _P])
_lib._SIGS['nele_norm_clip'] = _lib.lib.nele_norm_clip.argtypes

fs = 16000


def _dev32(x):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.cuda().float()
    return (t.unsqueeze(0) if t.dim() == 1 else t).contiguous()


def lfilter_fir(h, x):
    """scipy.signal.lfilter(h, [1], x) along the last axis: x [B, L] float32 (device or host), h [Lh] -> [B, L] float64."""
    x = _dev32(x)
    h = torch.as_tensor(np.asarray(h, dtype=np.float64)).cuda().contiguous()
    y = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    call('nele_fir_filter', ptr(x), x.shape[0], x.shape[1], ptr(h), h.numel(), ptr(y), stream())
    return y


def norm_clip(a, add=None, target_rms=0.0, return_steps=False, keep64=False):
    """(a [+ add]) scaled to ``target_rms`` (eval_metrics.py:104, 138, 141; 0: no scaling) and passed through clip()
    (audio_util.py:67-74).  a: [B, N] float64 or float32 device tensor -> float32 [B, N] (float64 with ``keep64``)."""
    a = a.contiguous()
    B, N = a.shape
    out = torch.empty((B, N), dtype=torch.float64 if keep64 else torch.float32, device=a.device)
    steps = torch.zeros(B, dtype=torch.int32, device=a.device)
    is64 = a.dtype == torch.float64
    if not is64:
        a = a.float()
    add = None if add is None else _dev32(add)
    call('nele_norm_clip', ptr(a) if is64 else None, None if is64 else ptr(a), ptr(add), B, N, float(target_rms), None if keep64 else ptr(out),
         ptr(out) if keep64 else None, ptr(steps), stream())
    return (out, steps) if return_steps else out


def listening_condition(clean, enh, noise, rir=None, tau=32, enh_rms=0.03):
    """-> (clean_a [B, L'], mixed [B, L']) float32 device tensors: what eval_metrics.py hands to the three metric wrappers.
    ``enh_rms`` > 0 applies the ``enh / rms(enh) * 0.03`` of :104 first.  rir None = 'NO_rev'."""
    clean, enh, noise = _dev32(clean), _dev32(enh), _dev32(noise)
    n = min(enh.shape[1], noise.shape[1])                                       # :111-114
    clean, enh, noise = clean[:, :n].contiguous(), enh[:, :n].contiguous(), noise[:, :n].contiguous()
    if enh_rms > 0:
        e64 = enh.double()
        enh = (e64 / torch.sqrt(torch.mean(e64 * e64, dim=1, keepdim=True)) * enh_rms).float()
    if rir is None:
        return clean, norm_clip(enh, add=noise)                                 # :118-122
    rir = np.asarray(rir, dtype=np.float32)
    b = int(np.argmax(rir))                                                     # :127-131
    h_direct = np.hstack([rir[:b + tau], np.zeros(len(rir) - (b + tau))])
    direct = norm_clip(lfilter_fir(h_direct, clean), target_rms=0.03)           # :132-135
    reverb = norm_clip(lfilter_fir(rir, enh), target_rms=0.03, keep64=True)     # :137-139
    clean_a = direct[:, b:].contiguous()
    mixed = norm_clip(reverb[:, b:].contiguous(), add=noise[:, b:].contiguous())  # :141-144
    return clean_a, mixed


def evaluate(clean, enh, noise, rir=None, metrics=('siib', 'haspi', 'estoi'), enh_rms=0.03):
    """Raw scores per utterance (eval_metrics.py:167-169) and their means (the printed summary line, :186)."""
    clean_a, mixed = listening_condition(clean, enh, noise, rir, enh_rms=enh_rms)
    fn = {'siib': mt.batch_siib, 'haspi': mt.batch_haspi, 'estoi': mt.batch_estoi}
    out = {m: fn[m](clean_a, mixed)[0].double().cpu().numpy() for m in metrics}
    out['summary'] = ', '.join('%s is %.3f' % (m.upper(), float(np.mean(out[m]))) for m in metrics)
    return out
