"""Oracle: ESTOI (Jensen & Taal 2016) as computed by ``pystoi.stoi(x, y, fs, extended=True)``,
the call at reference intel.py:126,133.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: pystoi (github mpariente/pystoi, no version pin in the reference, README.md:14) is
not vendored or installed; this restates its published algorithm and constants:
  resample to 10 kHz with the Octave-compatible polyphase filter (scipy.signal.resample_poly,
  Kaiser-windowed sinc, 60 dB, half-length 290 at 5/8), remove frames more than 40 dB below the
  loudest clean frame (256-sample Hann(258)[1:-1] frames, hop 128, overlap-add), 512-point STFT,
  15 third-octave bands from 150 Hz, 30-frame segments, row- then column- mean/variance
  normalisation, mean of the element-wise product.  Frame loops run while i + 256 <= len
  (pystoi >= 0.3).  pystoi adds eps-scaled Gaussian noise inside the normalisation (1e-16 relative);
  omitted here so that the score is deterministic.
"""
import numpy as np
from scipy.signal import resample_poly

FS = 10000
N_FRAME = 256
NFFT = 512
NUMBAND = 15
MINFREQ = 150
N = 30
DYN_RANGE = 40
EPS = np.finfo(np.float64).eps


def thirdoct(fs=FS, nfft=NFFT, num_bands=NUMBAND, min_freq=MINFREQ):
    f = np.linspace(0, fs, nfft + 1)[:nfft // 2 + 1]
    k = np.arange(num_bands, dtype=np.float64)
    freq_low = min_freq * np.power(2., (2 * k - 1) / 6)
    freq_high = min_freq * np.power(2., (2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)))
    edges = []
    for i in range(num_bands):
        lo = int(np.argmin(np.square(f - freq_low[i])))
        hi = int(np.argmin(np.square(f - freq_high[i])))
        obm[i, lo:hi] = 1
        edges.append((lo, hi))
    return obm, edges


def resample_window_oct(p, q):
    g = np.gcd(p, q)
    p, q = p // g, q // g
    log10_rejection = -3.0
    stopband_cutoff_f = 1. / (2 * max(p, q))
    roll_off_width = stopband_cutoff_f / 10
    rejection_dB = -20 * log10_rejection
    L = int(np.ceil((rejection_dB - 8) / (28.714 * roll_off_width)))
    t = np.arange(-L, L + 1)
    ideal = 2 * p * stopband_cutoff_f * np.sinc(2 * stopband_cutoff_f * t)
    beta = 0.1102 * (rejection_dB - 8.7)
    return np.kaiser(2 * L + 1, beta) * ideal


def resample_16k_to_10k(x):
    h = resample_window_oct(FS, 16000)
    return resample_poly(np.asarray(x, dtype=np.float64), FS, 16000, window=h / np.sum(h))


def hann_sym(n):
    return np.hanning(n + 2)[1:-1]


def frame_starts(n, framelen=N_FRAME, hop=N_FRAME // 2):
    return np.arange(0, n - framelen + 1, hop)


def remove_silent_frames(x, y, dyn_range=DYN_RANGE, framelen=N_FRAME, hop=N_FRAME // 2):
    w = hann_sym(framelen)
    st = frame_starts(len(x), framelen, hop)
    idx = st[:, None] + np.arange(framelen)[None, :]
    xf = w * x[idx]
    yf = w * y[idx]
    en = 20 * np.log10(np.linalg.norm(xf, axis=1) + EPS)
    mask = (np.max(en) - dyn_range - en) < 0
    xf, yf = xf[mask], yf[mask]
    n_sil = (len(xf) - 1) * hop + framelen
    xs = np.zeros(n_sil)
    ys = np.zeros(n_sil)
    for i in range(xf.shape[0]):
        xs[i * hop:i * hop + framelen] += xf[i]
        ys[i * hop:i * hop + framelen] += yf[i]
    return xs, ys, mask


def stft(x, win_size=N_FRAME, fft_size=NFFT, overlap=2):
    hop = win_size // overlap
    w = hann_sym(win_size)
    st = frame_starts(len(x), win_size, hop)
    idx = st[:, None] + np.arange(win_size)[None, :]
    return np.fft.rfft(w * x[idx], n=fft_size, axis=-1)


def row_col_normalize(x):
    x = x - np.mean(x, axis=-1, keepdims=True)
    x = x / np.sqrt(np.sum(np.square(x), axis=-1, keepdims=True))
    x = x - np.mean(x, axis=1, keepdims=True)
    x = x / np.sqrt(np.sum(np.square(x), axis=1, keepdims=True))
    return x


def estoi(x, y, fs=16000):
    """x clean, y degraded (same length, 16 kHz) -> raw ESTOI."""
    assert fs == 16000
    x = np.asarray(x)
    y = np.asarray(y)
    if x.shape != y.shape:
        raise ValueError('x and y should have the same length')
    x = resample_16k_to_10k(x)
    y = resample_16k_to_10k(y)
    x, y, _ = remove_silent_frames(x, y)
    X = stft(x).T
    Y = stft(y).T
    if X.shape[-1] < N:
        return 1e-5                     # pystoi: "Not enough STFT frames" warning path
    obm, _ = thirdoct()
    xt = np.sqrt(obm @ np.square(np.abs(X)))
    yt = np.sqrt(obm @ np.square(np.abs(Y)))
    J = xt.shape[1] - N + 1
    xs = np.stack([xt[:, m:m + N] for m in range(J)])
    ys = np.stack([yt[:, m:m + N] for m in range(J)])
    xn = row_col_normalize(xs)
    yn = row_col_normalize(ys)
    return float(np.sum(xn * yn / N) / J)


def estoi_wrapper(x, y, fs=16000, norm=True):
    """intel.py:122-134 (+ mapping :136-140 when norm)."""
    from .intel import mapping_ESTOI_harvard
    L = min(len(x), len(y))
    s = estoi(x[:L], y[:L], fs)
    return float(mapping_ESTOI_harvard(s)) if norm else s
