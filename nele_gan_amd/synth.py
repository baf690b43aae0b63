"""Seeded synthetic 16 kHz utterances (SURVEY.md section 8d recipe).

Clean: white Gaussian noise with a 1/f spectral tilt, multiplied by a 4 Hz raised-cosine
syllabic envelope with >= 15 % silent gaps, scaled to RMS 0.03 (the level the reference's data
is pre-normalised to, inference.py:109).  Noise: stationary pink noise at an SNR drawn from
{-11,-9,-7,-5,-3,-1} dB (the toy set's file-name convention).  Host-side numpy only; used by
bench.py, smoke() and the tests to make identical inputs for the HIP path and the oracle.
"""
import numpy as np

FS = 16000
SNRS_DB = (-11, -9, -7, -5, -3, -1)


def _shape_1_over_f(w, power):
    n = w.shape[0]
    spec = np.fft.rfft(w)
    f = np.fft.rfftfreq(n, 1.0 / FS)
    tilt = np.ones_like(f)
    tilt[1:] = (f[1:] / 100.0) ** (-power)
    tilt[f < 60.0] = 0.0
    return np.fft.irfft(spec * tilt, n)


def clean_utterance(i, length):
    rng = np.random.default_rng(1234 + i)
    w = rng.standard_normal(length)
    x = _shape_1_over_f(w, 0.5)
    t = np.arange(length) / FS
    rate = 4.0 + 0.5 * rng.standard_normal()
    phase = rng.uniform(0, 2 * np.pi)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * rate * t + phase)
    env = np.clip((env - 0.2) / 0.8, 0.0, 1.0) ** 2          # ~30 % of each syllable cycle silent
    # slower phrase-level modulation so that frames differ in level
    env *= 0.6 + 0.4 * np.cos(2 * np.pi * 0.7 * t + rng.uniform(0, 2 * np.pi)) ** 2
    x = x * np.maximum(env, 3e-3)                            # gaps are -50 dB, not digital silence
    x = x / np.sqrt(np.mean(x ** 2)) * 0.03
    return x.astype(np.float32)


def noise_utterance(i, length, clean=None, tilt=0.5):
    """tilt: spectral slope of the noise, 1/f^tilt in amplitude (0.5 = the bench recipe: the same long-term spectrum as the clean signal;
    larger = low-pass noise such as car or babble noise, where moving speech energy to higher bands pays - tools/learn_curve.py)"""
    rng = np.random.default_rng(987654 + i)
    w = rng.standard_normal(length)
    v = _shape_1_over_f(w, tilt)
    snr = SNRS_DB[int(rng.integers(0, len(SNRS_DB)))]
    ref_rms = 0.03 if clean is None else float(np.sqrt(np.mean(clean.astype(np.float64) ** 2)))
    v = v / np.sqrt(np.mean(v ** 2)) * ref_rms * 10 ** (-snr / 20.0)
    return v.astype(np.float32)


def batch(n, length, start=0, noise_tilt=0.5):
    """-> (clean [n, length] float32, noise [n, length] float32)."""
    c = np.stack([clean_utterance(start + k, length) for k in range(n)])
    v = np.stack([noise_utterance(start + k, length, c[k], tilt=noise_tilt) for k in range(n)])
    return c, v
