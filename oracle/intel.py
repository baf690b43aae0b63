"""Oracle: metric wrappers of reference ``intel.py`` (framing / VAD / SIIB replication rule /
logistic maps) in numpy.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Pinned against tests/golden/intel.npz (made by importing the reference's intel.py)."""
import numpy as np

EPS = np.finfo(np.float64).eps


def hann_periodic(n):
    # scipy.signal.get_window('hann', n) (fftbins=True); intel.py:34 asks for 'hanning' (same window)
    from scipy.signal import get_window
    return get_window('hann', n)


def framing(x, window_length=400, window_shift=200):
    """intel.py:16-35: rows start at 0, shift, ... < L - window_length (the frame starting exactly at
    L - window_length is NOT produced: as_strided yields L - window_length rows)."""
    x = np.asarray(x)
    slen = x.shape[-1]
    if slen < window_length + 1:
        x = np.pad(x, (0, window_length + 1 - slen), mode='constant')
    n = x.shape[-1] - window_length
    starts = np.arange(0, n, window_shift)
    idx = starts[:, None] + np.arange(window_length)[None, :]
    return x[idx] * hann_periodic(window_length)[None, :]


def n_frames(L, window_length=400, window_shift=200):
    L = max(L, window_length + 1)
    return -(-(L - window_length) // window_shift)


def get_vad(x, window_length=400, window_shift=200, delta_db=40):
    """intel.py:37-50."""
    fr = framing(x, window_length, window_shift)
    x_dB = 10 * np.log10((fr ** 2).mean(axis=1) + EPS)
    ind = int(round(len(x_dB) * 0.999) - 1)          # banker's rounding, as the reference
    max_x = np.partition(x_dB, ind)[ind]
    return x_dB > (max_x - delta_db)


def stft(x, window_length=400, window_shift=200):
    """intel.py:52-54."""
    return np.fft.fft(framing(x, window_length, window_shift), n=window_length, axis=-1)[:, :window_length // 2 + 1]


def siib_replication(x, fs=16000):
    """intel.py:84-97: M copies so that the active duration reaches 25 s when it is below 20 s."""
    R = 1 / 200 * fs
    nact = int(get_vad(x).sum())
    if nact / R < 20:
        return int(np.floor(25 / (nact / R))), nact
    return 1, nact


def mapping_SIIB_harvard(x):   # intel.py:102-106
    return 1 / (1 + np.exp(-0.06 * (x - 32)))


def mapping_HASPI_harvard(x):  # intel.py:116-120
    return 1 / (1 + np.exp(-0.95 * (x - 2.8)))


def mapping_ESTOI_harvard(x):  # intel.py:136-140
    return 1 / (1 + np.exp(-8.0 * (x - 0.25)))
