"""Oracle: listening conditions of the reference eval_metrics.py:100-169 (numpy / scipy, one utterance).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference script itself cannot run (hard-coded paths, absent
dependencies); lfilter / clip / rms are its own calls and are restated line by line."""
import numpy as np
from scipy.signal import lfilter


def rms(x):
    return np.sqrt(np.mean(x ** 2))                       # audio_util.py:463-464


def clip(x):
    """audio_util.py:67-74."""
    if np.max(x) >= 1 or np.min(x) < -1:
        small = 0.05
        while np.max(x) >= 1 or np.min(x) < -1:
            x = x / (1.0 + small)
            small = small + 0.05
    return x


def listening_condition(clean, enh, noise, rir=None, tau=32, enh_rms=0.03):
    clean, enh, noise = (np.asarray(a, dtype=np.float32) for a in (clean, enh, noise))
    if enh_rms > 0:
        enh = (enh.astype(np.float64) / rms(enh.astype(np.float64)) * enh_rms).astype(np.float32)
    n = min(len(enh), len(noise))
    enh, noise, clean = enh[:n], noise[:n], clean[:n]
    if rir is None:
        return clean, clip(enh.astype(np.float64) + noise)
    rir = np.asarray(rir, dtype=np.float32)
    b = int(np.argmax(rir))
    N = b + tau
    h_direct = np.hstack([rir[:N], np.zeros(len(rir) - N)])
    direct = lfilter(h_direct, [1], clean)
    direct = clip(direct / rms(direct) * 0.03)
    clean_a = direct[b:]
    reverb = lfilter(rir, [1], enh)
    reverb = clip(reverb / rms(reverb) * 0.03)
    mixed = clip(reverb[b:] + noise[b:])
    return clean_a, mixed
